#!/bin/bash
# builds scratch/lib_<tag>.so with extra -D flags: ./ab_build.sh tag -DFOO
set -e
cd /root/repo/rs_detection_amd/csrc
tag=$1; shift
mkdir -p /tmp/ab_$tag
for f in box_iou_rotated iou_fast anchor_target losses nms_rotated assign box_coder arf deform_conv alignconv_mfma conv3x3_mfma conv3x3_wrw_mfma rroi_align bn_act poly_iou feature_refine convex_sort dwconv layout optim canvas van_ops; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function "$@" -c $f.hip -o /tmp/ab_$tag/$f.o &
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/scratch/lib_$tag.so /tmp/ab_$tag/*.o
