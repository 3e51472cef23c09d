# experiment build: first `patch -p1 < tools/dbg/conv_gather_instrumentation.patch`, then rebuild with LGS_EXTRA_CFLAGS=-DLGS_CONV_DBG (round 5 moved the
# knock-out bits / shader-clock trace of k_conv_gather out of the product source into that patch)
cd $GRAFT_REPO_ROOT
for d in 0 7; do echo "== CONV_DBG=$d"; LGS_CONV_TRACE=1 LGS_CONV_DBG=$d python tools/microbench.py coarse 2>&1 | grep -E "trace" | awk 'NR%13==1' | head -12; done
