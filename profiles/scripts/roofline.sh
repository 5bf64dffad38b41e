#!/bin/bash
# One kernel-trace pass + the two PMC passes (WRITE_SIZE, FETCH_SIZE: separate passes, --kernel-trace only beside them,
# MI355X_MICROARCH.md) over profiles/scripts/pmc_kernels.py, then profiles/scripts/roofline.py -> <tag>_roofline.json.
# On the GPU box:  gpurun -- 'bash profiles/scripts/roofline.sh r02'   (outputs under gpurun_out/roofline/)
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/roofline
rm -rf $O; mkdir -p $O
export RSDET_ROOFLINE_DIR=$O
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o p -- python3 $R/profiles/scripts/pmc_kernels.py > $O/trace.log 2>&1
for c in WRITE_SIZE FETCH_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o p -- python3 $R/profiles/scripts/pmc_kernels.py > $O/pmc_$c.log 2>&1
done
python3 $R/profiles/scripts/roofline.py $O $O/${TAG}_roofline.json
