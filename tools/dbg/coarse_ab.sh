cd $GRAFT_REPO_ROOT
for cfg in 0 3 7; do
  echo "== SMALL_CFG=$cfg"; LGS_SMALL_CFG=$cfg python tools/microbench.py coarse 2>&1 | grep "^L[234]"
done
