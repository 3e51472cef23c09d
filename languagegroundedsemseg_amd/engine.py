"""ctypes binding of the C-ABI in include/lgs_engine.h (liblgs_engine.so).

This is the only place the product path touches native code, and it FAILS LOUDLY: if the
library is missing or a call returns non-zero a RuntimeError is raised -- there is no CPU or
PyTorch fallback anywhere in this package.
"""
import ctypes
import os

from . import build as _build

LGS_F32, LGS_BF16 = 0, 1
ABI_VERSION = 13     # LGS_ABI_VERSION of include/lgs_engine.h


class PackDesc(ctypes.Structure):
    """lgs_pack_desc (include/lgs_engine.h): layout of one packed weight image"""
    _fields_ = [("weight", ctypes.c_void_p), ("packed", ctypes.c_void_p), ("bytes", ctypes.c_int64), ("total", ctypes.c_int64),
                ("K", ctypes.c_int), ("cin_w", ctypes.c_int), ("cout_w", ctypes.c_int), ("transposed", ctypes.c_int),
                ("mirror", ctypes.c_int), ("g_real", ctypes.c_int), ("o_real", ctypes.c_int), ("ncp", ctypes.c_int),
                ("nbp", ctypes.c_int), ("dtype", ctypes.c_int)]

class BnParams(ctypes.Structure):
    """lgs_bn_params"""
    _fields_ = [("gamma", ctypes.c_void_p), ("beta", ctypes.c_void_p), ("running_mean", ctypes.c_void_p), ("running_var", ctypes.c_void_p),
                ("num_batches_tracked", ctypes.c_void_p), ("eps", ctypes.c_float), ("momentum", ctypes.c_float)]


class BlockFwd(ctypes.Structure):
    """lgs_block_fwd (include/lgs_engine.h)"""
    _fields_ = [("km3", ctypes.c_void_p), ("km1", ctypes.c_void_p),
                ("dtype", ctypes.c_int), ("relu_final", ctypes.c_int), ("cin", ctypes.c_int), ("planes", ctypes.c_int),
                ("n", ctypes.c_int64), ("x", ctypes.c_void_p),
                ("w1", ctypes.c_void_p), ("w2", ctypes.c_void_p), ("wd", ctypes.c_void_p),
                ("pk1", ctypes.c_void_p), ("pk2", ctypes.c_void_p), ("pkd", ctypes.c_void_p),
                ("pm1", ctypes.c_int), ("pm2", ctypes.c_int), ("pmd", ctypes.c_int),
                ("n1", BnParams), ("n2", BnParams), ("nd", BnParams),
                ("o1", ctypes.c_void_p), ("y1", ctypes.c_void_p), ("o2", ctypes.c_void_p), ("od", ctypes.c_void_p),
                ("res", ctypes.c_void_p), ("y2", ctypes.c_void_p),
                ("st1", ctypes.c_void_p), ("st2", ctypes.c_void_p), ("std_", ctypes.c_void_p),
                ("conv_ws", ctypes.c_void_p), ("bn_ws", ctypes.c_void_p)]


class BlockBwd(ctypes.Structure):
    """lgs_block_bwd (include/lgs_engine.h)"""
    _fields_ = [("km3", ctypes.c_void_p), ("km1", ctypes.c_void_p),
                ("dtype", ctypes.c_int), ("relu_final", ctypes.c_int), ("cin", ctypes.c_int), ("planes", ctypes.c_int),
                ("want_gin", ctypes.c_int), ("x_row_stride", ctypes.c_int),
                ("n", ctypes.c_int64), ("dy_row_stride", ctypes.c_int64), ("dy", ctypes.c_void_p),
                ("x", ctypes.c_void_p), ("o1", ctypes.c_void_p), ("y1", ctypes.c_void_p), ("o2", ctypes.c_void_p),
                ("y2", ctypes.c_void_p), ("od", ctypes.c_void_p),
                ("st1", ctypes.c_void_p), ("st2", ctypes.c_void_p), ("std_", ctypes.c_void_p),
                ("w1", ctypes.c_void_p), ("w2", ctypes.c_void_p), ("wd", ctypes.c_void_p),
                ("pk1", ctypes.c_void_p), ("pk2", ctypes.c_void_p), ("pkd", ctypes.c_void_p),
                ("pm1", ctypes.c_int), ("pm2", ctypes.c_int), ("pmd", ctypes.c_int),
                ("gamma1", ctypes.c_void_p), ("beta1", ctypes.c_void_p), ("gamma2", ctypes.c_void_p), ("beta2", ctypes.c_void_p),
                ("gammad", ctypes.c_void_p), ("betad", ctypes.c_void_p),
                ("dx2", ctypes.c_void_p), ("dres", ctypes.c_void_p), ("dy1", ctypes.c_void_p), ("dx1", ctypes.c_void_p),
                ("dxd", ctypes.c_void_p), ("gind", ctypes.c_void_p),
                ("gw1", ctypes.c_void_p), ("gw2", ctypes.c_void_p), ("gwd", ctypes.c_void_p),
                ("dgamma1", ctypes.c_void_p), ("dbeta1", ctypes.c_void_p), ("dgamma2", ctypes.c_void_p), ("dbeta2", ctypes.c_void_p),
                ("dgammad", ctypes.c_void_p), ("dbetad", ctypes.c_void_p),
                ("conv_ws", ctypes.c_void_p), ("bn_ws", ctypes.c_void_p),
                ("wgrad_stream", ctypes.c_void_p), ("wgrad_ws", ctypes.c_void_p), ("fork_event", ctypes.c_void_p),
                ("ev_w1", ctypes.c_void_p), ("ev_w2", ctypes.c_void_p), ("ev_wd", ctypes.c_void_p)]


# struct-module formats of lgs_block_fwd / lgs_block_bwd (native alignment); checked against the ctypes layouts at import
BLOCK_FWD_FMT = "@PPiiiiqP" + "PPP" + "PPP" + "iii" + "PPPPPff" * 3 + "PPPPPP" + "PPP" + "PP"
BLOCK_BWD_FMT = "@PPiiiiiiqqP" + "PPPPPP" + "PPP" + "PPP" + "PPP" + "iii" + "PPPPPP" + "PPPPPP" + "PPP" + "PPPPPP" + "PP" + "PPPPPP"
import struct as _struct
assert _struct.calcsize(BLOCK_FWD_FMT) == ctypes.sizeof(BlockFwd), (_struct.calcsize(BLOCK_FWD_FMT), ctypes.sizeof(BlockFwd))
assert _struct.calcsize(BLOCK_BWD_FMT) == ctypes.sizeof(BlockBwd), (_struct.calcsize(BLOCK_BWD_FMT), ctypes.sizeof(BlockBwd))
BLOCK_FWD_PACK = _struct.Struct(BLOCK_FWD_FMT)
BLOCK_BWD_PACK = _struct.Struct(BLOCK_BWD_FMT)

_lib = None

# every symbol include/lgs_engine.h declares; tests check the built library exports all of them
EXPORTS = [
    "lgs_abi_version", "lgs_last_error",
    "lgs_tuning_set", "lgs_tuning_get", "lgs_tuning_describe", "lgs_debug_dispatch_counts",
    "lgs_manager_create", "lgs_manager_destroy", "lgs_manager_insert", "lgs_manager_stride2", "lgs_manager_check",
    "lgs_manager_parent_of", "lgs_manager_map_size", "lgs_manager_get_coords", "lgs_manager_kernel_map",
    "lgs_kmap_export",
    "lgs_conv_workspace_bytes", "lgs_conv_bn_partial_rows", "lgs_conv_forward", "lgs_conv_dgrad", "lgs_conv_wgrad",
    "lgs_conv_wgrad_supports_stride", "lgs_conv_dgrad_can_accumulate", "lgs_conv_dgrad_accumulate",
    "lgs_conv_pack_desc", "lgs_pack_weights_batch",
    "lgs_bn_workspace_bytes", "lgs_bn_forward", "lgs_bn_backward",
    "lgs_block_workspace_bytes", "lgs_block_forward", "lgs_block_backward",
    "lgs_bn_stats", "lgs_bn_sync_combine", "lgs_bn_apply", "lgs_bn_backward_reduce", "lgs_bn_backward_apply",
    "lgs_clip_similarity", "lgs_clip_workspace_bytes",
    "lgs_clip_loss_workspace_bytes", "lgs_clip_loss_forward", "lgs_clip_loss_backward",
    "lgs_clip_anchor_grad_workspace_bytes", "lgs_clip_loss_backward_anchors",
    "lgs_ce_forward_backward", "lgs_ce_forward_backward_rows", "lgs_split_stats",
    "lgs_ce_count_valid",
    "lgs_comm_unique_id", "lgs_comm_create", "lgs_comm_create_ipc", "lgs_comm_ipc_open", "lgs_comm_destroy", "lgs_comm_world", "lgs_bn_sync_workspace_bytes",
    "lgs_bn_forward_sync", "lgs_bn_backward_sync",
    "lgs_voxelize", "lgs_label_vote", "lgs_cluster_workspace_bytes", "lgs_cluster", "lgs_sgd_step",
]


def lib_path():
    return _build.LIB_PATH


def lib():
    """Load (once) the engine. Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            "liblgs_engine.so not found at %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or languagegroundedsemseg_amd.build.build()). There is no fallback path." % path)
    L = ctypes.CDLL(path)
    vp, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
    pi, pi64, pvp = ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_void_p)
    cf = ctypes.c_float
    L.lgs_abi_version.restype = ci
    L.lgs_abi_version.argtypes = []
    L.lgs_last_error.restype = ctypes.c_char_p
    L.lgs_last_error.argtypes = []
    sig = {
        "lgs_block_forward": [vp, vp],        # (const lgs_block_fwd *, stream): the host packs the struct with struct.pack_into
        "lgs_block_backward": [vp, vp],
        "lgs_tuning_set": [ctypes.c_char_p, i64],
        "lgs_tuning_get": [ctypes.c_char_p, pi64],
        "lgs_manager_create": [ci, pvp],
        "lgs_manager_destroy": [vp],
        "lgs_manager_insert": [vp, vp, i64, vp, vp, vp, pi, pi64],
        "lgs_manager_stride2": [vp, ci, vp, pi, pi64],
        "lgs_manager_check": [vp, pi],
        "lgs_manager_parent_of": [vp, ci, pi],
        "lgs_manager_map_size": [vp, ci, pi64, pi],
        "lgs_manager_get_coords": [vp, ci, vp, vp],
        "lgs_manager_kernel_map": [vp, ci, ci, ci, vp, pvp],
        "lgs_kmap_export": [vp, vp, vp, vp, vp, pi64],
        "lgs_conv_forward": [vp, ci, vp, ci, vp, ci, vp, vp, ci, vp, vp, vp, vp, ci, ci, vp],
        "lgs_conv_pack_desc": [vp, ci, ci, ci, ci, ci, ctypes.POINTER(PackDesc)],
        "lgs_pack_weights_batch": [vp, ci, i64, vp],
        "lgs_conv_bn_partial_rows": [vp, ci, ci, ci],
        "lgs_conv_dgrad": [vp, ci, vp, ci, vp, ci, vp, ci, vp, vp, ci, vp],
        "lgs_sgd_step": [vp, vp, vp, vp, i64, cf, cf, cf, cf, ci, vp],
        "lgs_cluster": [vp, vp, vp, i64, cf, ci, vp, ctypes.POINTER(ctypes.c_int32), vp, vp],
        "lgs_voxelize": [vp, i64, ctypes.POINTER(ctypes.c_double), ci, vp, vp],
        "lgs_label_vote": [vp, i64, vp, vp, i64, i64, vp, vp],
        "lgs_conv_wgrad": [vp, ci, vp, ci, vp, ci, vp, ci, vp, ci, vp],
        "lgs_conv_wgrad_supports_stride": [vp, ci, ci, ci, ci, ci],
        "lgs_conv_dgrad_can_accumulate": [vp, ci, ci, ci, ci],
        "lgs_conv_dgrad_accumulate": [vp, ci, vp, ci, vp, ci, vp, ci, vp, vp, ci, vp],
        "lgs_bn_forward": [vp, i64, ci, vp, vp, cf, cf, vp, vp, vp, vp, ci, vp, vp, ci, vp, vp, ci, vp, i64, vp],
        "lgs_bn_backward": [vp, vp, vp, i64, i64, ci, vp, vp, vp, ci, vp, vp, vp, vp, ci, vp, i64, vp],
        "lgs_bn_stats": [vp, i64, ci, vp, ci, vp, vp, ci, vp, vp],
        "lgs_bn_apply": [vp, i64, ci, vp, vp, vp, vp, ci, vp, ci, i64, vp],
        "lgs_bn_backward_reduce": [vp, vp, vp, i64, ci, vp, vp, vp, ci, vp, vp, vp, ci, vp, i64, i64, vp],
        "lgs_bn_sync_combine": [vp, ci, ci, cf, cf, vp, vp, vp, vp, vp, vp],
        "lgs_bn_backward_apply": [vp, vp, vp, i64, ci, vp, vp, vp, vp, cf, vp, ci, vp, vp, ci, i64, i64, vp],
        "lgs_clip_similarity": [vp, i64, ci, vp, ci, vp, vp, ci, vp, vp],
        "lgs_ce_forward_backward": [vp, i64, ci, vp, i64, vp, vp, vp, ci, vp],
        "lgs_ce_forward_backward_rows": [vp, i64, ci, vp, i64, vp, vp, vp, vp, ci, vp],
        "lgs_split_stats": [vp, vp, i64, vp, ci, i64, vp, ci, vp],
        "lgs_ce_count_valid": [vp, i64, ci, i64, vp, vp],
        "lgs_comm_unique_id": [vp],
        "lgs_comm_create": [vp, ci, ci, ci, ctypes.POINTER(vp)],
        "lgs_comm_create_ipc": [ci, ci, ci, ctypes.POINTER(vp), vp],
        "lgs_comm_ipc_open": [vp, vp],
        "lgs_comm_destroy": [vp],
        "lgs_comm_world": [vp],
        "lgs_bn_forward_sync": [vp, vp, i64, ci, vp, vp, cf, cf, vp, vp, vp, vp, ci, vp, vp, vp, ci, vp, i64, vp],
        "lgs_bn_backward_sync": [vp, vp, vp, vp, i64, ci, vp, vp, vp, vp, ci, vp, vp, vp, vp, ci, vp, i64, i64, vp],
        "lgs_clip_loss_forward": [vp, i64, ci, vp, ci, vp, vp, ci, i64, vp, vp, vp, vp, vp, vp, ci, vp, vp],
        "lgs_clip_loss_backward": [vp, i64, ci, vp, ci, vp, vp, ci, i64, vp, vp, vp, vp, vp, vp, ci, vp],
        "lgs_clip_loss_backward_anchors": [vp, i64, ci, ci, vp, vp, ci, i64, vp, vp, vp, vp, ci, vp, vp],
    }
    for name, args in sig.items():
        f = getattr(L, name)
        f.restype = ci
        f.argtypes = args
    L.lgs_tuning_describe.restype = i64
    L.lgs_tuning_describe.argtypes = [ctypes.c_char_p, i64]
    L.lgs_debug_dispatch_counts.restype = i64
    L.lgs_debug_dispatch_counts.argtypes = [ctypes.c_char_p, i64, ci]
    L.lgs_block_workspace_bytes.restype = i64
    L.lgs_block_workspace_bytes.argtypes = [vp, vp, ci, ci, ci]
    L.lgs_cluster_workspace_bytes.restype = i64
    L.lgs_cluster_workspace_bytes.argtypes = [i64]
    L.lgs_conv_workspace_bytes.restype = i64
    L.lgs_conv_workspace_bytes.argtypes = [vp, ci, ci, ci, ci]
    L.lgs_bn_workspace_bytes.restype = i64
    L.lgs_bn_workspace_bytes.argtypes = [i64, ci]
    L.lgs_bn_sync_workspace_bytes.restype = i64
    L.lgs_bn_sync_workspace_bytes.argtypes = [i64, ci, ci]
    L.lgs_clip_workspace_bytes.restype = i64
    L.lgs_clip_workspace_bytes.argtypes = [ci, ci, ci]
    L.lgs_clip_loss_workspace_bytes.restype = i64
    L.lgs_clip_loss_workspace_bytes.argtypes = [ci, ci, ci]
    L.lgs_clip_anchor_grad_workspace_bytes.restype = i64
    L.lgs_clip_anchor_grad_workspace_bytes.argtypes = [i64, ci, ci, ci]
    if L.lgs_abi_version() != ABI_VERSION:
        raise RuntimeError("liblgs_engine.so ABI version mismatch")
    _lib = L
    return L


def check(rc):
    if rc != 0:
        msg = lib().lgs_last_error()
        raise RuntimeError("lgs_engine: " + (msg.decode(errors="replace") if msg else "error %d" % rc))


# ---- tuning table / dispatch counters (include/lgs_engine.h, csrc/lgs_tuning.hip)
TUNING_EPOCH = 0      # bumped by every tuning_set(): host-side caches of knob-dependent answers (pack descriptors) key on it


def tuning_set(name, value):
    global TUNING_EPOCH
    check(lib().lgs_tuning_set(name.encode(), int(value)))
    TUNING_EPOCH += 1


def tuning_get(name):
    v = ctypes.c_int64()
    check(lib().lgs_tuning_get(name.encode(), ctypes.byref(v)))
    return int(v.value)


class tuning:
    """with engine.tuning(WW_MIN_ROWS=0): ...   -- set knobs for a block (tests), restore afterwards"""

    def __init__(self, **knobs):
        self.knobs = knobs

    def __enter__(self):
        self.prev = {k: tuning_get(k) for k in self.knobs}
        for k, v in self.knobs.items():
            tuning_set(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.prev.items():
            tuning_set(k, v)


def tuning_table():
    """-> [(name, default, value, doc)]"""
    L = lib()
    n = L.lgs_tuning_describe(None, 0)
    buf = ctypes.create_string_buffer(int(n))
    L.lgs_tuning_describe(buf, n)
    rows = []
    for line in buf.value.decode().splitlines():
        name, d, v, doc = line.split("\t", 3)
        rows.append((name, int(d), int(v), doc))
    return rows


def dispatch_counts(reset=False):
    """-> {launch site: launches since the last reset}"""
    L = lib()
    cap = 1 << 19
    buf = ctypes.create_string_buffer(cap)
    L.lgs_debug_dispatch_counts(buf, cap, 1 if reset else 0)
    out = {}
    for line in buf.value.decode().splitlines():
        c, name = line.split("\t", 1)
        out[name] = int(c)
    return out
