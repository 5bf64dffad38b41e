#!/usr/bin/env python3
"""Build oracle/_ref/libjdet_ref_hip.so: the reference's CUDA-ONLY ops compiled AS DEVICE CODE for gfx950.

TEST INFRASTRUCTURE ONLY (nothing under rs_detection_amd/ or bench.py's timed region loads it).  Runs only where
/root/reference exists (the build container: hipcc cross-compiles without a GPU); the .so travels to the GPU box with
the snapshot (oracle/_ref/ is git-ignored, not gpurun-ignored).

The reference keeps four ops as CUDA text only (no CPU source): deformable im2col / col2im / col2im_coord
(python/jdet/ops/dcn_v1.py:6-306 ``HEADER``), ROIAlignRotated_v1 (ops/roi_align_rotated_v1.py:7-298 ``CUDA_HEADER``),
FeatureRefine (ops/fr.py:5-232 ``HEADER``) and polygon NMS (ops/nms_poly.py:4-185 ``HEADER``).  Rounds 1-4 pinned
them through a HOST macro shim (tests/golden/make_golden.py: ``__global__`` -> nothing, one host thread walks the
index range, host cos/sin, no FMA).  That text is plain CUDA C++ with nothing NVIDIA-specific in it, so ``hipcc``
compiles it UNMODIFIED as device code: this recipe

 1. reads the header strings with ``ast`` (no import, no exec of reference code);
 2. drops ``#include <executor.h>`` (Jittor's runtime header; nothing of it is used by the kernels) -- the ONLY edit;
 3. puts each header in its own namespace and appends thin ``extern "C"`` launchers that restate the launch statement
    of the op's ``jt.code`` source (grid = GET_BLOCKS(n), block = the header's own constant, the ``cudaMemsetAsync`` of
    the output where the reference has one) -- file:line cited at each launcher;
 4. compiles with ``hipcc --offload-arch=gfx950 -O2`` and the compiler's DEFAULT floating-point contraction, which is
    what nvcc's default (-fmad=true) is for the reference: device sinf/cosf, FMA where the compiler forms one.

The generated translation unit lives in a temporary directory and is deleted; only the .so lands in oracle/_ref/.
Launchers take DEVICE pointers, run on the null stream (Jittor's default stream) and return hipGetLastError()."""
import ast
import os
import re
import shutil
import subprocess
import sys
import tempfile

REF_OPS = "/root/reference/python/jdet/ops"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_ref")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def module_string(path, name):
    tree = ast.parse(open(path).read())
    for st in tree.body:
        if isinstance(st, ast.Assign) and getattr(st.targets[0], "id", None) == name:
            return ast.literal_eval(st.value)
    raise KeyError(name)


def header(fname, var):
    return re.sub(r"#include\s*<\s*executor\.h\s*>", "", module_string(os.path.join(REF_OPS, fname), var))


PRELUDE = r'''
#include <hip/hip_runtime.h>
#include <algorithm>
#include <vector>
#include <iostream>     // (system headers first: the reference headers re-include them INSIDE our namespaces)
#include <stdio.h>
#include <math.h>
#include <float.h>
#include <cstring>
#include <cstdio>
#include <cmath>
#include <cfloat>
#include <climits>
'''

# launch of dcn_v1.py:314-338 (im2col), :382-409 (col2im), :346-372 (col2im_coord)
LAUNCH_DCN = r'''
extern "C" int ref_hip_dcn_im2col(const float* in0_p, const float* in1_p, int channels, int height, int width,
    int ksize_h, int ksize_w, int pad_h, int pad_w, int stride_h, int stride_w, int dilation_h, int dilation_w,
    int parallel_imgs, int deformable_group, float* out0_p) {
  using namespace ref_dcn;
  int height_col = (height + 2 * pad_h - (dilation_h * (ksize_h - 1) + 1)) / stride_h + 1;
  int width_col = (width + 2 * pad_w - (dilation_w * (ksize_w - 1) + 1)) / stride_w + 1;
  int num_kernels = channels * height_col * width_col * parallel_imgs;
  int channel_per_deformable_group = channels / deformable_group;
  hipMemsetAsync(out0_p, 0, sizeof(float) * (size_t)channels * ksize_h * ksize_w * parallel_imgs * height_col * width_col);
  deformable_im2col_gpu_kernel<<<GET_BLOCKS(num_kernels), CUDA_NUM_THREADS>>>(
      num_kernels, in0_p, in1_p, height, width, ksize_h, ksize_w, pad_h, pad_w, stride_h, stride_w, dilation_h,
      dilation_w, channel_per_deformable_group, parallel_imgs, channels, deformable_group, height_col, width_col, out0_p);
  return (int)hipGetLastError();
}
extern "C" int ref_hip_dcn_col2im(const float* in0_p, const float* in1_p, int channels, int height, int width,
    int ksize_h, int ksize_w, int pad_h, int pad_w, int stride_h, int stride_w, int dilation_h, int dilation_w,
    int parallel_imgs, int deformable_group, float* out0_p) {
  using namespace ref_dcn;
  int height_col = (height + 2 * pad_h - (dilation_h * (ksize_h - 1) + 1)) / stride_h + 1;
  int width_col = (width + 2 * pad_w - (dilation_w * (ksize_w - 1) + 1)) / stride_w + 1;
  int num_kernels = channels * ksize_h * ksize_w * height_col * width_col * parallel_imgs;
  int channel_per_deformable_group = channels / deformable_group;
  const int whole_size = parallel_imgs + channels + height + width;          // `sum(grad_im_shape)`, :395 (unused by the kernel's arithmetic)
  hipMemsetAsync(out0_p, 0, sizeof(float) * (size_t)parallel_imgs * channels * height * width);
  deformable_col2im_gpu_kernel<<<GET_BLOCKS(num_kernels), CUDA_NUM_THREADS>>>(
      num_kernels, in0_p, in1_p, channels, height, width, ksize_h, ksize_w, pad_h, pad_w, stride_h, stride_w,
      dilation_h, dilation_w, channel_per_deformable_group, parallel_imgs, deformable_group, height_col, width_col,
      out0_p, whole_size);
  return (int)hipGetLastError();
}
extern "C" int ref_hip_dcn_col2im_coord(const float* in0_p, const float* in1_p, const float* in2_p, int channels,
    int height, int width, int ksize_h, int ksize_w, int pad_h, int pad_w, int stride_h, int stride_w, int dilation_h,
    int dilation_w, int parallel_imgs, int deformable_group, float* out0_p) {
  using namespace ref_dcn;
  int height_col = (height + 2 * pad_h - (dilation_h * (ksize_h - 1) + 1)) / stride_h + 1;
  int width_col = (width + 2 * pad_w - (dilation_w * (ksize_w - 1) + 1)) / stride_w + 1;
  int num_kernels = height_col * width_col * 2 * ksize_h * ksize_w * deformable_group * parallel_imgs;
  int channel_per_deformable_group = channels * ksize_h * ksize_w / deformable_group;
  hipMemsetAsync(out0_p, 0, sizeof(float) * (size_t)num_kernels);
  deformable_col2im_coord_gpu_kernel<<<GET_BLOCKS(num_kernels), CUDA_NUM_THREADS>>>(
      num_kernels, in0_p, in1_p, in2_p, channels, height, width, ksize_h, ksize_w, pad_h, pad_w, stride_h, stride_w,
      dilation_h, dilation_w, channel_per_deformable_group, parallel_imgs, 2 * ksize_h * ksize_w * deformable_group,
      deformable_group, height_col, width_col, out0_p);
  return (int)hipGetLastError();
}
'''

# launch of roi_align_rotated_v1.py:308-325 (forward), :329-350 (backward)
LAUNCH_RROI = r'''
extern "C" int ref_hip_rroi_forward(const float* input_p, const float* rois_p, int num_rois, int channels, int height,
    int width, int pooled_height, int pooled_width, float spatial_scale, float sampling_ratio, float* output_p) {
  using namespace ref_rroi;
  auto output_size = num_rois * pooled_height * pooled_width * channels;
  ROIAlignRotatedForward<<<GET_BLOCKS(output_size), THREADS_PER_BLOCK>>>(
      output_size, input_p, rois_p, spatial_scale, sampling_ratio, channels, height, width, pooled_height, pooled_width,
      output_p);
  return (int)hipGetLastError();
}
extern "C" int ref_hip_rroi_backward(const float* grad_p, const float* rois_p, int num_rois, int batch, int channels,
    int height, int width, int pooled_height, int pooled_width, float spatial_scale, float sampling_ratio,
    float* grad_input_p) {
  using namespace ref_rroi;
  auto output_size = num_rois * pooled_height * pooled_width * channels;
  hipMemsetAsync(grad_input_p, 0, sizeof(float) * (size_t)batch * channels * height * width);
  ROIAlignBackward<<<GET_BLOCKS(output_size), THREADS_PER_BLOCK>>>(
      output_size, grad_p, rois_p, spatial_scale, sampling_ratio, channels, height, width, pooled_height, pooled_width,
      grad_input_p);
  return (int)hipGetLastError();
}
'''

# launch of fr.py:234-240 (forward), :244-252 (backward: bottom_grad = zeros_like first)
LAUNCH_FR = r'''
extern "C" int ref_hip_fr_forward(const float* in0_p, const float* in1_p, int n, int channels, int height, int width,
    int points, float spatial_scale, float* out0_p) {
  using namespace ref_fr;
  const int output_size = n * channels * height * width;
  feature_refine_forward_kernel<<<GET_BLOCKS(output_size), THREADS_PER_BLOCK>>>(
      output_size, points, in0_p, in1_p, spatial_scale, channels, height, width, out0_p);
  return (int)hipGetLastError();
}
extern "C" int ref_hip_fr_backward(const float* in0_p, const float* in1_p, int n, int channels, int height, int width,
    int points, float spatial_scale, float* out0_p) {
  using namespace ref_fr;
  const int output_size = n * channels * height * width;
  hipMemsetAsync(out0_p, 0, sizeof(float) * (size_t)output_size);
  feature_refine_backward_kernel<<<GET_BLOCKS(output_size), THREADS_PER_BLOCK>>>(
      output_size, points, in0_p, in1_p, spatial_scale, channels, height, width, out0_p);
  return (int)hipGetLastError();
}
'''

# nms_poly.py:197-229: mask kernel, device sync, host sweep over the 64-bit mask rows (keep is a HOST array here)
LAUNCH_POLY = r'''
extern "C" int ref_hip_poly_nms(const float* boxes_sorted_p, int boxes_num, float nms_overlap_thresh,
                                unsigned char* keep_host) {
  using namespace ref_poly;
  const int col_blocks = THCCeilDiv(boxes_num, threadsPerBlock);
  size_t matrices_size = (size_t)boxes_num * col_blocks * sizeof(unsigned long long);
  unsigned long long* mask_p = nullptr;
  if (hipMalloc(&mask_p, matrices_size) != hipSuccess) return -1;
  dim3 blocks(THCCeilDiv(boxes_num, threadsPerBlock), THCCeilDiv(boxes_num, threadsPerBlock));
  dim3 threads(threadsPerBlock);
  poly_nms_kernel<<<blocks, threads, 0>>>(boxes_num, nms_overlap_thresh, boxes_sorted_p, mask_p);
  if (hipDeviceSynchronize() != hipSuccess) { hipFree(mask_p); return -2; }
  std::vector<unsigned long long> mask_h((size_t)boxes_num * col_blocks);
  hipMemcpy(mask_h.data(), mask_p, matrices_size, hipMemcpyDeviceToHost);
  std::vector<unsigned long long> remv(col_blocks);
  memset(&remv[0], 0, sizeof(unsigned long long) * col_blocks);
  memset(keep_host, 0, boxes_num);
  for (int i = 0; i < boxes_num; i++) {
    int nblock = i / threadsPerBlock;
    int inblock = i % threadsPerBlock;
    if (!(remv[nblock] & (1ULL << inblock))) {
      keep_host[i] = 1;
      unsigned long long* p = mask_h.data() + (size_t)i * col_blocks;
      for (int j = nblock; j < col_blocks; j++) remv[j] |= p[j];
    }
  }
  hipFree(mask_p);
  return (int)hipGetLastError();
}
// the pair function of the mask kernel on its own (IoU of two 8-float polygons), one lane per pair
namespace ref_poly { __global__ void pair_iou_kernel(const float* p, const float* q, int n, float* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = devPolyIoU(p + 8 * i, q + 8 * i);
} }
extern "C" int ref_hip_poly_iou_pairs(const float* p, const float* q, int n, float* out) {
  ref_poly::pair_iou_kernel<<<(n + 63) / 64, 64>>>(p, q, n, out);
  return (int)hipGetLastError();
}
'''


def main():
    if not os.path.isdir(REF_OPS):
        print("build_ref_hip: %s not present -- using prebuilt oracle/_ref if any" % REF_OPS)
        return 0
    tu = PRELUDE
    tu += "namespace ref_dcn {\nusing std::min; using std::max;\n" + header("dcn_v1.py", "HEADER") + "\n}\n" + LAUNCH_DCN
    tu += "namespace ref_rroi {\nusing std::min; using std::max;\n" + header("roi_align_rotated_v1.py", "CUDA_HEADER") + "\n}\n" + LAUNCH_RROI
    tu += "namespace ref_fr {\nusing std::min; using std::max;\n" + header("fr.py", "HEADER") + "\n}\n" + LAUNCH_FR
    tu += "namespace ref_poly {\n" + header("nms_poly.py", "HEADER") + "\n}\n" + LAUNCH_POLY
    # the same polygon text once more with floating-point contraction OFF (nvcc -fmad=false / a CPU build): its signed
    # triangle sums cancel to ~1e-3 of the IoU, so the reference's own results move by that much with the compiler's
    # contraction choice; rsdet_poly_* follows the uncontracted arithmetic and must equal THIS build bit for bit
    tu += ("#pragma clang fp contract(off)\nnamespace ref_poly_nofma {\n" + header("nms_poly.py", "HEADER") + "\n}\n"
           + LAUNCH_POLY.replace("ref_poly", "ref_poly_nofma").replace("ref_hip_poly", "ref_hip_nofma_poly")
           + "#pragma clang fp contract(fast)\n")
    os.makedirs(OUT_DIR, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="jdet_ref_hip_")
    try:
        src = os.path.join(tmp, "jdet_ref_hip_tu.hip")
        with open(src, "w") as f:
            f.write(tu)
        out = os.path.join(OUT_DIR, "libjdet_ref_hip.so")
        cmd = [HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", "-o", out, src]
        subprocess.check_call(cmd)
        print("build_ref_hip: built", out)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
