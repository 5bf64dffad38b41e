#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
sed -i 's/^    check(/    pass  # check(/' profiles/scripts/alignconv_mfma_check.py
for t in base acm_zeroa acm_nob acm_nomfma acm_zab; do
  echo "== $t"
  if [ $t = base ]; then timeout 100 python3 profiles/scripts/alignconv_mfma_check.py 2>&1 | grep bench
  else RSDET_LIB_PATH=$R/scratch/lib_$t.so timeout 100 python3 profiles/scripts/alignconv_mfma_check.py 2>&1 | grep bench; fi
done
