from languagegroundedsemseg_amd.me.utils import *  # noqa: F401,F403
from languagegroundedsemseg_amd.me.utils import SparseCollation, batched_coordinates, sparse_collate, sparse_quantize  # noqa: F401
