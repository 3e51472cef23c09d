from languagegroundedsemseg_amd.me.ops import *  # noqa: F401,F403
from languagegroundedsemseg_amd.me.ops import cat  # noqa: F401
