"""languagegroundedsemseg_amd -- MI355X-native sparse-voxel engine behind the MinkowskiEngine
operator surface used by RozDavid/LanguageGroundedSemseg's Res16UNet + CLIP-alignment hot path.

Layout:
  csrc/      hand-written HIP kernels + the C-ABI (include/lgs_engine.h)
  engine.py  ctypes binding (fails loudly when the library is missing)
  me/        host-side mirror of the MinkowskiEngine Python API (also importable as `MinkowskiEngine`)
  models.py  the Res16UNet family reproduced on that surface (state-dict compatible)
  losses.py  contrastive CLIP loss on the MFMA contraction
"""
import os as _os

# The engine drives FOUR HIP streams per device (compute, weight gradients, coordinate / kernel maps, input staging) next to
# whatever torch and RCCL create.  The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4):
# with a fifth stream two of them share a queue, i.e. run IN ORDER -- measured: the input-staging / map stream landed behind the
# compute stream, so every `SparseTensor(...)` (its insert returns a count to the host) blocked the host until the GPU had
# finished the previous training step (one 145 k-voxel scene per step: 11.5 vs 10.4 ms).  The variable is read when the HIP
# runtime initialises, i.e. at the process's first device call: it has to be set before that (importing this package first
# is enough; an explicit setting by the user wins).  One process per GPU is assumed, as everywhere in this build: several processes
# time-sharing a GPU should keep the default (2 x 8 queues oversubscribe the device: 136 vs 520 ms per step in the two-rank dry run).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

__version__ = "0.1.0"
