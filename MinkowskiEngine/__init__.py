"""Drop-in alias: `import MinkowskiEngine as ME` resolves to the MI355X-native engine.
Put the repository root on PYTHONPATH and the reference's models/, lib/losses and
trainers import unchanged."""
from languagegroundedsemseg_amd.me import *  # noqa: F401,F403
from languagegroundedsemseg_amd.me import (MinkowskiConvolutionFunction, MinkowskiConvolutionTransposeFunction,  # noqa: F401
                                           __version__, get_backend, invalidate_packed_weights, set_backend, utils)
from . import MinkowskiOps  # noqa: F401
