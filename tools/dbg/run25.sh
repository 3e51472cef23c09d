cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
for d in 0 16 20 28 1; do echo "== LGS_WIDE_DBG=$d"; LGS_WIDE_DBG=$d timeout 600 python tools/microbench.py wide 2>&1 | grep "L0 3^3 512->512" | cut -c1-120; done
