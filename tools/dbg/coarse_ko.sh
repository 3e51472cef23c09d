# experiment build: first `patch -p1 < tools/dbg/conv_gather_instrumentation.patch`, then rebuild with LGS_EXTRA_CFLAGS=-DLGS_CONV_DBG (round 5 moved the
# knock-out bits / shader-clock trace of k_conv_gather out of the product source into that patch)
cd $GRAFT_REPO_ROOT
for d in 0 1 2 4 6 7 3; do echo "== CONV_DBG=$d (1 no MFMA, 2 no gathers, 4 no weight loads)"; LGS_CONV_DBG=$d python tools/microbench.py coarse 2>&1 | grep -E "^L[34]"; done
