"""languagegroundedsemseg_amd -- MI355X-native sparse-voxel engine behind the MinkowskiEngine
operator surface used by RozDavid/LanguageGroundedSemseg's Res16UNet + CLIP-alignment hot path.

Layout:
  csrc/      hand-written HIP kernels + the C-ABI (include/lgs_engine.h)
  engine.py  ctypes binding (fails loudly when the library is missing)
  me/        host-side mirror of the MinkowskiEngine Python API (also importable as `MinkowskiEngine`)
  models.py  the Res16UNet family reproduced on that surface (state-dict compatible)
  losses.py  contrastive CLIP loss on the MFMA contraction
"""
from . import tuning as _tuning

# GPU_MAX_HW_QUEUES is a process-wide HIP runtime setting: the APPLICATION makes that call (tuning.configure_hw_queues(), as
# bench.py and tests/conftest.py do -- INTEGRATION.md section 1), importing this package does not, unless asked to with
# LGS_SET_HW_QUEUES=1.
if _tuning.host("SET_HW_QUEUES"):
    _tuning.configure_hw_queues()

__version__ = "0.1.0"
