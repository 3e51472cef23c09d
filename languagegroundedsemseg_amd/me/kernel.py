"""RegionType / KernelGenerator -- the kernel-shape descriptors of the ME surface
(/root/reference/models/modules/common.py:55-64,192-193,219-226).  On D=3 the model family only
uses HYPER_CUBE regions with kernel sizes {1,2,3}, strides {1,2}, dilation 1."""
import collections.abc
from enum import Enum

import torch


class RegionType(Enum):
    HYPER_CUBE = 0
    HYPER_CROSS = 1
    CUSTOM = 2


def convert_to_int_list(arg, dimension):
    if isinstance(arg, torch.Tensor):
        arg = arg.tolist()
    if isinstance(arg, collections.abc.Sequence):
        assert len(arg) == dimension, "argument length %d != dimension %d" % (len(arg), dimension)
        return [int(a) for a in arg]
    return [int(arg)] * dimension


def convert_to_int_tensor(arg, dimension):
    return torch.IntTensor(convert_to_int_list(arg, dimension))


def convert_region_type(region_type, tensor_stride=None, kernel_size=None, up_stride=None, dilation=None,
                        region_offset=None, axis_types=None, dimension=None, center=True):
    """Kept for import compatibility (models/conditional_random_fields.py:5-6)."""
    return region_type, region_offset, 0


def get_kernel_volume(region_type, kernel_size, region_offset=None, axis_types=None, dimension=3):
    ks = convert_to_int_list(kernel_size, dimension)
    if region_type == RegionType.HYPER_CUBE:
        v = 1
        for k in ks:
            v *= k
        return v
    if region_type == RegionType.HYPER_CROSS:
        return sum(k - 1 for k in ks) + 1
    return int(region_offset.shape[0]) if region_offset is not None else 0


class KernelGenerator:
    def __init__(self, kernel_size=-1, stride=1, dilation=1, is_transpose=False, region_type=RegionType.HYPER_CUBE,
                 region_offsets=None, expand_coordinates=False, axis_types=None, dimension=-1):
        assert dimension > 0
        self.dimension = dimension
        self.kernel_size = convert_to_int_list(kernel_size, dimension)
        self.kernel_stride = convert_to_int_list(stride, dimension)
        self.kernel_dilation = convert_to_int_list(dilation, dimension)
        self.region_type = region_type if region_type is not None else RegionType.HYPER_CUBE
        self.region_offsets = region_offsets
        self.axis_types = axis_types
        self.is_transpose = is_transpose
        self.expand_coordinates = expand_coordinates
        self.kernel_volume = get_kernel_volume(self.region_type, self.kernel_size, region_offsets, axis_types, dimension)
