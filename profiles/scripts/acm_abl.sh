#!/bin/bash
# the implicit-GEMM AlignConv under several wave splits (scratch/lib_<tag>.so built by ab_build.sh -DACM_CW= -DACM_PW=)
R=$GRAFT_REPO_ROOT
cd $R
for t in "$@"; do
  echo "== $t"
  if [ $t = base ]; then timeout 100 python3 profiles/scripts/alignconv_mfma_check.py 2>&1 | grep "^bench"
  else RSDET_LIB_PATH=$R/scratch/lib_$t.so timeout 100 python3 profiles/scripts/alignconv_mfma_check.py 2>&1 | grep "^bench\|rror"; fi
done
