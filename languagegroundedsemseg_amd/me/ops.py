"""`MinkowskiEngine.MinkowskiOps` -- models/res16unet.py:6 does `import MinkowskiEngine.MinkowskiOps as me`
and calls `me.cat(out, skip)`."""
from .core import cat  # noqa: F401
