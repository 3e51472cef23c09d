cd $GRAFT_REPO_ROOT
for cfg in 0 12 13 9 10 11; do
  echo "== SMALL_CFG=$cfg"; LGS_SMALL_CFG=$cfg python tools/microbench.py coarse 2>&1 | grep "^L[34]"
done
echo "== CONV_SPLIT=0"; LGS_CONV_SPLIT=0 python tools/microbench.py coarse 2>&1 | grep "^L[34]"
