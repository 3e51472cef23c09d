"""`MinkowskiEngine`-compatible namespace.  Imported as
`languagegroundedsemseg_amd.me`, and re-exported verbatim by the top-level `MinkowskiEngine/` package
so the reference's `import MinkowskiEngine as ME` resolves to this engine."""
import collections
import collections.abc

# The reference targets Python 3.8 and uses collections.Sequence / collections.Iterable
# (models/modules/common.py:81,96; lib/voxelizer.py:53), removed in 3.10.  "Loads unchanged" needs the aliases.
for _n in ("Sequence", "Iterable", "Mapping", "MutableMapping", "Callable"):
    if not hasattr(collections, _n):
        setattr(collections, _n, getattr(collections.abc, _n))

from .core import (CoordinateManager, CoordinateMapKey, SparseTensor, SparseTensorQuantizationMode, cat,  # noqa: E402
                   get_backend, set_backend)
from .kernel import (KernelGenerator, RegionType, convert_region_type, convert_to_int_list,  # noqa: E402
                     convert_to_int_tensor, get_kernel_volume)
from .modules import *  # noqa: E402,F401,F403
from .modules import MinkowskiConvolutionFunction, MinkowskiConvolutionTransposeFunction  # noqa: E402,F401
from . import utils  # noqa: E402,F401
from . import ops as MinkowskiOps  # noqa: E402,F401



def invalidate_packed_weights():
    """Extension (not in MinkowskiEngine): call after writing conv weights behind torch's version counters -- `p.data.copy_()`,
    `dist.broadcast(p.data)`, raw-pointer updates.  reset_parameters(), load_state_dict(), BucketedDDP's initial broadcast
    and FlatSGD.step() already do."""
    be = get_backend()
    if hasattr(be, "invalidate_packed_weights"):
        be.invalidate_packed_weights()


__version__ = "0.5.4"  # API level mirrored
