#!/bin/bash
# HBM traffic of the hand-written kernels from the PMC counters, one counter per pass (MI355X_MICROARCH.md: --pmc only
# together with --kernel-trace).  Run on the GPU box:  gpurun -- 'bash profiles/scripts/pmc_passes.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for c in WRITE_SIZE FETCH_SIZE; do
  rm -rf $O/pmc_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o p -- python3 $R/profiles/scripts/pmc_kernels.py > /dev/null 2>&1
done
python3 $R/profiles/scripts/pmc_summary.py $O/pmc_WRITE_SIZE $O/pmc_FETCH_SIZE
