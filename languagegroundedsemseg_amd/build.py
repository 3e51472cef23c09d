"""Builds liblgs_engine.so (the C-ABI HIP engine) in-tree for gfx950.

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with the
gpurun snapshot, so nothing is JIT-compiled at run time.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "_lib")
LIB_PATH = os.path.join(LIB_DIR, "liblgs_engine.so")
SOURCES = ["lgs_manager.hip", "lgs_conv.hip", "lgs_conv_wide.hip", "lgs_pointwise.hip", "lgs_wgrad.hip", "lgs_wgrad_wide.hip", "lgs_norm.hip", "lgs_loss.hip", "lgs_voxel.hip", "lgs_cluster.hip", "lgs_tuning.hip", "lgs_block.hip", "lgs_comm.hip"]
HEADERS = ["lgs_common.h", os.path.join("..", "..", "include", "lgs_engine.h")]


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    """Compile every HIP source of the engine for gfx950 -> _lib/liblgs_engine.so."""
    if not force and not _stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(LIB_DIR, src.replace(".hip", ".o"))
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + os.environ.get("LGS_EXTRA_CFLAGS", "").split() + \
              ["-c", os.path.join(CSRC, src), "-o", obj]      # LGS_EXTRA_CFLAGS: experiment builds only (e.g. -DLGS_CONV_DBG)
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode(errors="replace")))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs + ["-ldl"]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if out.returncode != 0:
        raise RuntimeError("hipcc link failed:\n%s" % out.stdout.decode(errors="replace"))
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
