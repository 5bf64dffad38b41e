#!/usr/bin/env python3
"""Generate tests/golden/*.npz -- golden input/output vectors for the hot path.

Run ONCE in the build container (needs /root/reference); the .npz files are committed,
this script is committed, no reference source is.  Provenance of every fixture:

  iou_v0.npz, iou_v1.npz, nms5.npz, nms6.npz, arf.npz
      outputs of oracle/_ref/libjdet_ref.so = the reference's OWN embedded CPU sources
      (python/jdet/ops/box_iou_rotated.py:312-326,487-500; box_iou_rotated_v1.py:317-331,492-505;
      nms_rotated.py:314-328,414-449; orn.py:132-257) compiled by oracle/build_ref.py with
      g++ -O2 -std=c++14.  ARF sizes keep O*I*nEntry <= 65535 (the CPU kernel's uint16 index
      wraps beyond that).
  dcn.npz, rroi.npz
      the reference has CUDA text only for these (dcn_v1.py:132-306, roi_align_rotated_v1.py:71-298).
      That text is compiled here AS HOST C++ behind a macro shim (__device__/__global__ -> nothing,
      blockIdx/threadIdx = 0, blockDim/gridDim = 1 so CUDA_KERNEL_LOOP walks the whole range on one
      thread, atomicAdd -> "+=") and executed single-threaded.  The arithmetic is the reference's
      own; what is NOT the reference toolchain: nvcc's FMA contraction and device cosf/sinf
      (host cos/sin in double here).  DESIGN.md records this as "pinned through a host shim".
  assign.npz, coder.npz
      Jittor tensor code cannot be imported; these hold seeded inputs and the outputs of the
      oracle's restatement (parity unpinned for Jittor's argmax tie-break; first index chosen).

Seeds, compiler and flags are stored inside each file (key "provenance").
"""
import ast
import ctypes
import os
import re
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF_OPS = "/root/reference/python/jdet/ops"

import oracle  # noqa: E402
from conftest import dota_boxes, degenerate_boxes, s2anet_anchors  # noqa: E402

F = ctypes.POINTER(ctypes.c_float)


def fp(a):
    return a.ctypes.data_as(F)


def module_string(path, name):
    tree = ast.parse(open(path).read())
    for st in tree.body:
        if isinstance(st, ast.Assign) and getattr(st.targets[0], "id", None) == name:
            return ast.literal_eval(st.value)
    raise KeyError(name)


SHIM = r'''
#include <cmath>
#include <cstdio>
#include <climits>
#include <cfloat>
#include <algorithm>
#include <math.h>
#define __device__
#define __global__
#define __host__
#define __forceinline__ inline
struct _idx3 { int x, y, z; };
static _idx3 blockIdx = {0, 0, 0}, threadIdx = {0, 0, 0}, blockDim = {1, 1, 1}, gridDim = {1, 1, 1};
template <class T> static inline void atomicAdd(T* p, T v) { *p += v; }
using std::min; using std::max;
'''

WRAP_DCN = r'''
extern "C" void ref_dcn_im2col(const float* im, const float* off, int C, int H, int W, int kh, int kw, int ph,
    int pw, int sh, int sw, int dh, int dw, int B, int dg, int Ho, int Wo, float* col) {
  int n = C * Ho * Wo * B;
  deformable_im2col_gpu_kernel<float>(n, im, off, H, W, kh, kw, ph, pw, sh, sw, dh, dw, C / dg, B, C, dg, Ho, Wo, col);
}
extern "C" void ref_dcn_col2im(const float* col, const float* off, int C, int H, int W, int kh, int kw, int ph,
    int pw, int sh, int sw, int dh, int dw, int B, int dg, int Ho, int Wo, float* gim) {
  int n = C * kh * kw * Ho * Wo * B;
  deformable_col2im_gpu_kernel<float>(n, col, off, C, H, W, kh, kw, ph, pw, sh, sw, dh, dw, C / dg, B, dg, Ho, Wo, gim, 0);
}
extern "C" void ref_dcn_col2im_coord(const float* col, const float* im, const float* off, int C, int H, int W,
    int kh, int kw, int ph, int pw, int sh, int sw, int dh, int dw, int B, int dg, int Ho, int Wo, float* goff) {
  int n = Ho * Wo * 2 * kh * kw * dg * B;
  deformable_col2im_coord_gpu_kernel<float>(n, col, im, off, C, H, W, kh, kw, ph, pw, sh, sw, dh, dw,
      C * kh * kw / dg, B, 2 * kh * kw * dg, dg, Ho, Wo, goff);
}
'''

WRAP_RROI = r'''
extern "C" void ref_rroi_fwd(const float* feat, const float* rois, int R, int C, int H, int W, int PH, int PW,
    float scale, int sr, float* out) {
  ROIAlignRotatedForward<float>(R * PH * PW * C, feat, rois, scale, sr, C, H, W, PH, PW, out);
}
extern "C" void ref_rroi_bwd(const float* gout, const float* rois, int R, int C, int H, int W, int PH, int PW,
    float scale, int sr, float* gfeat) {
  ROIAlignBackward<float>(R * PH * PW * C, gout, rois, scale, sr, C, H, W, PH, PW, gfeat);
}
'''


def build_shim_one(tmp, name, text):
    """Compile one shimmed translation unit (reference CUDA text as host C++) and load it."""
    cpp = os.path.join(tmp, name + ".cpp")
    open(cpp, "w").write(text)
    so = os.path.join(tmp, name + ".so")
    subprocess.check_call(["g++", "-O2", "-std=c++14", "-fPIC", "-shared", "-w", "-ffp-contract=off", "-o", so, cpp])
    return ctypes.CDLL(so)


def build_shim_lib(tmp):
    dcn = module_string(os.path.join(REF_OPS, "dcn_v1.py"), "HEADER")
    dcn = re.sub(r"#include\s*<\s*executor\.h\s*>", "", dcn)
    rroi = module_string(os.path.join(REF_OPS, "roi_align_rotated_v1.py"), "CUDA_HEADER")
    src = SHIM + "namespace dcn {" + dcn + "}\nusing namespace dcn;\n" + WRAP_DCN
    src2 = SHIM + "namespace rroi {" + rroi + "}\nusing namespace rroi;\n" + WRAP_RROI
    libs = []
    for name, text in (("dcn", src), ("rroi", src2)):
        cpp = os.path.join(tmp, name + ".cpp")
        open(cpp, "w").write(text)
        so = os.path.join(tmp, name + ".so")
        subprocess.check_call(["g++", "-O2", "-std=c++14", "-fPIC", "-shared", "-w", "-ffp-contract=off", "-o", so, cpp])
        libs.append(ctypes.CDLL(so))
    return libs


def prov(extra=""):
    gxx = subprocess.check_output(["g++", "--version"]).decode().splitlines()[0]
    return np.array("generated by tests/golden/make_golden.py from /root/reference (zcablii/RS_detection); "
                    "%s; flags -O2 -std=c++14; %s" % (gxx, extra))


def main():
    ref = oracle.ref()
    assert ref.available, "run oracle/build_ref.py first"
    c = oracle.c()
    rng = np.random.default_rng(20240601)

    # ---- rotated IoU ---------------------------------------------------------------
    for v in (0, 1):
        b1 = np.concatenate([dota_boxes(rng, 40, 300), degenerate_boxes()])
        b2 = np.concatenate([dota_boxes(rng, 300, 300), degenerate_boxes()])
        gts = dota_boxes(rng, 6)
        anchors = s2anet_anchors()[::7]  # subsampled S2ANet grid
        np.savez_compressed(os.path.join(HERE, "iou_v%d.npz" % v), boxes1=b1, boxes2=b2,
                            ious=ref.box_iou_rotated(b1, b2, v), gts=gts, anchors=anchors,
                            ious_anchor=ref.box_iou_rotated(gts, anchors, v),
                            known_in=np.array([[0, 0, 1, 1, 0], [.5, .5, 1, 2, 0]], np.float32),
                            known_out=np.array([[1, .2], [.2, 1]], np.float32),
                            provenance=prov("seed 20240601; reference CPU source, version %d" % v))
    # ---- NMS -------------------------------------------------------------------------
    for bl in (5, 6):
        centres = dota_boxes(rng, 40, 500)
        idx = rng.integers(0, 40, 600)
        d = centres[idx].copy()
        d[:, :2] += rng.normal(0, 3, (600, 2)).astype(np.float32)
        d[:, 4] += rng.normal(0, 0.1, 600).astype(np.float32)
        if bl == 6:
            d = np.concatenate([d, rng.integers(0, 15, (600, 1)).astype(np.float32)], 1)
        scores = rng.uniform(0.05, 1, 600).astype(np.float32)
        order = np.argsort(-scores, kind="stable").astype(np.int32)
        out = {"dets": d, "scores": scores, "order": order}
        for thr in (0.1, 0.3, 0.8):
            out["keep_%g" % thr] = ref.nms_rotated(d, order, thr)
        kd = np.array([[0, 0, 1, 1, 0], [0, 0, .5, .5, .3], [0, 0, .9, .9, 0]], np.float32)
        if bl == 6:
            kd = np.concatenate([kd, np.ones((3, 1), np.float32)], 1)
        ko = np.array([2, 1, 0], np.int32)
        out.update(known_dets=kd, known_order=ko, known_keep=ref.nms_rotated(kd, ko, 0.3))
        np.savez_compressed(os.path.join(HERE, "nms%d.npz" % bl), provenance=prov("seed 20240601; CPU greedy, >="), **out)
    # ---- ARF ---------------------------------------------------------------------------
    from rs_detection_amd.ops.orn import arf_indices
    out = {}
    for tag, (O, I, nOri, nRot, k) in {"a": (8, 256, 1, 8, 3), "b": (6, 5, 4, 8, 3), "c": (4, 9, 8, 4, 1)}.items():
        idx = arf_indices(nOri, nRot, (k, k)).numpy()
        w = rng.standard_normal((O, I, nOri, k, k)).astype(np.float32)
        fwd = ref.arf_forward(w, idx)
        go = rng.standard_normal(fwd.shape).astype(np.float32)
        out.update({tag + "_w": w, tag + "_idx": idx, tag + "_fwd": fwd, tag + "_go": go,
                    tag + "_bwd": ref.arf_backward(idx, go)})
    np.savez_compressed(os.path.join(HERE, "arf.npz"), provenance=prov("seed 20240601; reference CPU ARF"), **out)

    # ---- DCN / RROI via the host shim ----------------------------------------------------
    tmp = tempfile.mkdtemp(prefix="jdet_shim_")
    try:
        ldcn, lrroi = build_shim_lib(tmp)
        out = {}
        geoms = {"a": dict(B=2, C=4, H=6, W=5, kh=3, kw=3, ph=1, pw=1, sh=1, sw=1, dh=1, dw=1, dg=1),
                 "b": dict(B=1, C=6, H=9, W=11, kh=3, kw=3, ph=1, pw=1, sh=2, sw=2, dh=1, dw=1, dg=2),
                 "c": dict(B=2, C=8, H=12, W=12, kh=3, kw=3, ph=2, pw=2, sh=1, sw=1, dh=2, dw=2, dg=1)}
        for tag, g in geoms.items():
            gl = [g[k] for k in ("kh", "kw", "ph", "pw", "sh", "sw", "dh", "dw")]
            Ho, Wo = oracle._COracle.out_hw(g["H"], g["W"], *gl)
            im = rng.standard_normal((g["B"], g["C"], g["H"], g["W"])).astype(np.float32)
            off = (rng.standard_normal((g["B"], g["dg"] * 18, Ho, Wo)) * 2.5).astype(np.float32)
            col = np.zeros((g["C"] * 9, g["B"], Ho, Wo), np.float32)
            ldcn.ref_dcn_im2col(fp(im), fp(off), g["C"], g["H"], g["W"], *gl, g["B"], g["dg"], Ho, Wo, fp(col))
            gcol = rng.standard_normal(col.shape).astype(np.float32)
            gim = np.zeros_like(im)
            ldcn.ref_dcn_col2im(fp(gcol), fp(off), g["C"], g["H"], g["W"], *gl, g["B"], g["dg"], Ho, Wo, fp(gim))
            goff = np.zeros_like(off)
            ldcn.ref_dcn_col2im_coord(fp(gcol), fp(im), fp(off), g["C"], g["H"], g["W"], *gl, g["B"], g["dg"], Ho,
                                      Wo, fp(goff))
            out.update({tag + "_geom": np.array([g[k] for k in ("B", "C", "H", "W", "kh", "kw", "ph", "pw", "sh",
                                                                 "sw", "dh", "dw", "dg")], np.int32),
                        tag + "_im": im, tag + "_off": off, tag + "_col": col, tag + "_gcol": gcol,
                        tag + "_gim": gim, tag + "_goff": goff})
        np.savez_compressed(os.path.join(HERE, "dcn.npz"),
                            provenance=prov("seed 20240601; reference CUDA text on host via macro shim"), **out)
        out = {}
        for tag, (N, C, H, W, R, scale, sr) in {"a": (2, 3, 16, 20, 7, 0.25, 2), "b": (1, 5, 32, 32, 9, 0.125, 0),
                                                "c": (2, 4, 24, 24, 6, 1 / 16., 2)}.items():
            feat = rng.standard_normal((N, C, H, W)).astype(np.float32)
            b = dota_boxes(rng, R, W / scale, 8, 120, 60)
            rois = np.concatenate([rng.integers(0, N, (R, 1)).astype(np.float32), b], 1)
            rois[0, 1:3] = [-30, -30]
            rois[1, 3:5] = [0.5, 0.5]
            o = np.zeros((R, C, 7, 7), np.float32)
            lrroi.ref_rroi_fwd(fp(feat), fp(rois), R, C, H, W, 7, 7, ctypes.c_float(scale), sr, fp(o))
            go = rng.standard_normal(o.shape).astype(np.float32)
            gf = np.zeros_like(feat)
            lrroi.ref_rroi_bwd(fp(go), fp(rois), R, C, H, W, 7, 7, ctypes.c_float(scale), sr, fp(gf))
            out.update({tag + "_feat": feat, tag + "_rois": rois, tag + "_cfg": np.array([scale, sr], np.float64),
                        tag + "_out": o, tag + "_go": go, tag + "_gfeat": gf})
        np.savez_compressed(os.path.join(HERE, "rroi.npz"),
                            provenance=prov("seed 20240601; reference CUDA text on host via macro shim"), **out)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)

    # ---- assigner / coder (restatement outputs; parity unpinned vs Jittor) ------------------
    gts, anchors = dota_boxes(rng, 30), s2anet_anchors()[::3]
    ov = ref.box_iou_rotated(gts, anchors, 0)
    labels = rng.integers(1, 16, 30).astype(np.int32)
    gi, mo, lb = c.assign_wrt_overlaps(ov, 0.5, 0.4, 0.0, True, True, labels, 0)
    np.savez_compressed(os.path.join(HERE, "assign.npz"), gts=gts, anchors=anchors, overlaps=ov, gt_labels=labels,
                        gt_inds=gi, max_overlaps=mo, labels=lb,
                        provenance=prov("overlaps from the reference CPU IoU; assignment = oracle restatement of "
                                        "models/boxes/assigner.py:125-168, first-index argmax"))
    prop, gt = dota_boxes(rng, 500), dota_boxes(rng, 500)
    deltas = (rng.standard_normal((500, 5)) * 0.3).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "coder.npz"), proposals=prop, gt=gt, deltas=deltas,
                        encoded=oracle.np_bbox2delta_rotated(prop, gt),
                        decoded=oracle.np_delta2bbox_rotated(prop, deltas),
                        decoded_clip1e6=oracle.np_delta2bbox_rotated(prop, deltas, wh_ratio_clip=1e-6),
                        provenance=prov("NumPy transcription of models/boxes/box_ops.py:176-289, seed 20240601"))
    make_rie(ref)
    make_rroi_v0()
    make_fr()
    make_convex(ref)
    make_poly_nms()
    print("golden fixtures written to", HERE)


def make_rie(ref):
    """rie.npz: rotation-invariant encoding from the reference's own CPU source (ops/orn.py:290-363 compiled by
    oracle/build_ref.py; the backward through the header kernel, see build_ref.py).  Own RNG: adding this fixture
    did not change the others."""
    rng = np.random.default_rng(20240602)
    out = {}
    for tag, (n, nf, nori) in {"a": (7, 32, 8), "b": (3, 5, 4), "c": (16, 64, 8)}.items():
        f = rng.standard_normal((n, nf * nori, 1, 1)).astype(np.float32)
        f[0, :nori, 0, 0] = 1.0                     # all equal: the first orientation wins (strict '>')
        if nori >= 3:
            f[1, :3, 0, 0] = [2.0, 5.0, 5.0]        # tie of the maximum: the first of them
        d, al = ref.rie_forward(f, nori)
        go = rng.standard_normal(f.shape).astype(np.float32)
        out.update({tag + "_f": f, tag + "_nori": np.int32(nori), tag + "_dir": d, tag + "_aligned": al,
                    tag + "_go": go, tag + "_gi": ref.rie_backward(d, go, nori)})
    np.savez_compressed(os.path.join(HERE, "rie.npz"), provenance=prov("seed 20240602; reference CPU RIE"), **out)


def make_rroi_v0():
    """rroi_v0.npz: ROIAlignRotated (ops/roi_align_rotated.py CUDA_HEADER :7-254) through the host shim, fixed and
    adaptive sampling grids, RoIs outside the map and smaller than one pixel.  Own RNG."""
    rng = np.random.default_rng(20240603)
    tmp = tempfile.mkdtemp(prefix="jdet_shim_")
    try:
        text = module_string(os.path.join(REF_OPS, "roi_align_rotated.py"), "CUDA_HEADER")
        lib = build_shim_one(tmp, "rroi0", SHIM + "namespace rroi {" + text + "}\nusing namespace rroi;\n" + WRAP_RROI)
        out = {}
        for tag, (N, C, H, W, R, scale, sr) in {"a": (2, 3, 16, 20, 7, 0.25, 2), "b": (1, 5, 32, 32, 9, 0.125, 0),
                                                "c": (2, 4, 24, 24, 6, 1 / 16., 3)}.items():
            feat = rng.standard_normal((N, C, H, W)).astype(np.float32)
            b = dota_boxes(rng, R, W / scale, 8, 120, 60)
            rois = np.concatenate([rng.integers(0, N, (R, 1)).astype(np.float32), b], 1)
            rois[0, 1:3] = [-30, -30]
            rois[1, 3:5] = [0.5, 0.5]
            o = np.zeros((R, C, 7, 7), np.float32)
            lib.ref_rroi_fwd(fp(feat), fp(rois), R, C, H, W, 7, 7, ctypes.c_float(scale), sr, fp(o))
            go = rng.standard_normal(o.shape).astype(np.float32)
            gf = np.zeros_like(feat)
            lib.ref_rroi_bwd(fp(go), fp(rois), R, C, H, W, 7, 7, ctypes.c_float(scale), sr, fp(gf))
            out.update({tag + "_feat": feat, tag + "_rois": rois, tag + "_cfg": np.array([scale, sr], np.float64),
                        tag + "_out": o, tag + "_go": go, tag + "_gfeat": gf})
        np.savez_compressed(os.path.join(HERE, "rroi_v0.npz"),
                            provenance=prov("seed 20240603; reference CUDA text on host via macro shim"), **out)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


WRAP_FR = r'''
extern "C" void ref_fr_fwd(const float* feat, const float* boxes, int N, int C, int H, int W, float scale, int points,
    float* out) {
  feature_refine_forward_kernel<float>(N * C * H * W, points, feat, boxes, scale, C, H, W, out);
}
extern "C" void ref_fr_bwd(const float* top, const float* boxes, int N, int C, int H, int W, float scale, int points,
    float* gin) {
  feature_refine_backward_kernel<float>(N * C * H * W, points, top, boxes, scale, C, H, W, gin);
}
'''


def make_fr():
    """fr.npz: FeatureRefine (ops/fr.py HEADER :5-232) through the host shim; boxes drawn as the reference's own
    test does (fr.py:349-377: centres on the stride grid + noise, log-normal sizes, angle in (-pi/2, 0]), so many
    sample points fall outside the map.  points = 1 and 5.  Own RNG."""
    rng = np.random.default_rng(20240604)
    tmp = tempfile.mkdtemp(prefix="jdet_shim_")
    try:
        text = module_string(os.path.join(REF_OPS, "fr.py"), "HEADER")
        lib = build_shim_one(tmp, "fr", SHIM + "namespace fr {" + text + "}\nusing namespace fr;\n" + WRAP_FR)
        out = {}
        for tag, (N, C, H, W, stride, points) in {"a": (2, 4, 16, 16, 8.0, 1), "b": (2, 3, 12, 20, 8.0, 5),
                                                  "c": (1, 6, 9, 7, 16.0, 5)}.items():
            feat = rng.standard_normal((N, C, H, W)).astype(np.float32)
            base = 4.0 * stride
            yc, xc = np.meshgrid(stride * np.arange(H), stride * np.arange(W), indexing="ij")
            xc = xc[None] + base * rng.standard_normal((N, H, W))
            yc = yc[None] + base * rng.standard_normal((N, H, W))
            w = base * np.exp(rng.standard_normal((N, H, W)))
            h = base * np.exp(rng.standard_normal((N, H, W)))
            a = -np.pi / 2 * rng.random((N, H, W))
            # entry 0 is consumed as the ROW coordinate (fr.py:131): put yc first so that most points land inside
            boxes = np.stack([yc, xc, w, h, a], -1).astype(np.float32)
            o = np.zeros_like(feat)
            lib.ref_fr_fwd(fp(feat), fp(boxes), N, C, H, W, ctypes.c_float(1.0 / stride), points, fp(o))
            go = rng.standard_normal(o.shape).astype(np.float32)
            gi = np.zeros_like(feat)
            lib.ref_fr_bwd(fp(go), fp(boxes), N, C, H, W, ctypes.c_float(1.0 / stride), points, fp(gi))
            out.update({tag + "_feat": feat, tag + "_boxes": boxes, tag + "_cfg": np.array([1.0 / stride, points]),
                        tag + "_out": o, tag + "_go": go, tag + "_gin": gi})
        np.savez_compressed(os.path.join(HERE, "fr.npz"),
                            provenance=prov("seed 20240604; reference CUDA text on host via macro shim"), **out)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def convex_cases(rng):
    """Point sets for convex_sort: (i) random clouds with random masks, (ii) the poly_iou_loss shape (24 points:
    16 edge intersections + 2 x 4 vertices, most of them masked off), (iii) integer lattice points (collinear runs,
    exact duplicates, equal cosine keys), (iv) all points masked off / a single valid point, (v) npts = 100 (the
    global-workspace variant of the kernel)."""
    cases = {}
    pts = (rng.standard_normal((300, 24, 2)) * 20).astype(np.float32)
    cases["a"] = (pts, rng.random((300, 24)) > 0.3, True)
    pts = (rng.standard_normal((200, 8, 2)) * 50).astype(np.float32)
    cases["b"] = (pts, np.ones((200, 8), bool), False)
    pts = rng.integers(-3, 4, (400, 24, 2)).astype(np.float32)
    cases["c"] = (pts, rng.random((400, 24)) > 0.2, True)
    pts = (rng.standard_normal((70, 12, 2))).astype(np.float32)
    m = rng.random((70, 12)) > 0.5
    m[:10] = False
    m[10:20] = False
    m[10:20, 3] = True
    cases["d"] = (pts, m, True)
    pts = (rng.standard_normal((130, 100, 2)) * 5).astype(np.float32)
    cases["e"] = (pts, rng.random((130, 100)) > 0.1, True)
    return cases


def make_convex(ref):
    """convex.npz: the reference's own CPU Graham-scan loop (ops/convex_sort.py:93-154, compiled by
    oracle/build_ref.py) on start / order arrays prepared by oracle.np_convex_sort_prepare (the Jittor tensor code
    :67-84 restated in NumPy; its tie rules are unpinned).  Own RNG."""
    rng = np.random.default_rng(20240605)
    out = {}
    for tag, (pts, m, circ) in convex_cases(rng).items():
        x, y, mf, start, order = oracle.np_convex_sort_prepare(pts, m)
        out.update({tag + "_pts": pts, tag + "_masks": m, tag + "_circular": np.bool_(circ), tag + "_start": start,
                    tag + "_order": order, tag + "_index": ref.convex_sort_scan(x, y, mf, start, order, circ)})
    np.savez_compressed(os.path.join(HERE, "convex.npz"),
                        provenance=prov("seed 20240605; reference CPU convex_sort loop"), **out)


SHIM_POLY = r'''
#include <vector>
#include <iostream>
struct float2 { float x, y; };
static inline float2 make_float2(float x, float y) { float2 r; r.x = x; r.y = y; return r; }
#define __shared__ static
static inline void __syncthreads() {}
'''

WRAP_POLY = r'''
extern "C" void ref_poly_iou(const float* p1, int n1, const float* p2, int n2, float* out) {
  for (int i = 0; i < n1; ++i)
    for (int j = 0; j < n2; ++j) out[(size_t)i * n2 + j] = devPolyIoU(p1 + (size_t)i * 8, p2 + (size_t)j * 8);
}
'''


def poly_nms_cases(rng):
    """Quadrilateral detections for the fp32 in-model polygon NMS: (a) clustered rotated rectangles at image scale
    (the Gliding-Vertex use), (b) the same with the per-class coordinate offset of multiclass_poly_nms (large
    coordinates: this is where the float cancellation noise of the reference shows), (c) general convex / skewed
    quadrilaterals with both orientations, identical and degenerate (zero-area) ones."""
    from conftest import dota_boxes
    import oracle as _o
    cases = {}
    centres = dota_boxes(rng, 30, 800)
    b = centres[rng.integers(0, 30, 400)].copy()
    b[:, :2] += rng.normal(0, 4, (400, 2)).astype(np.float32)
    b[:, 4] += rng.normal(0, 0.1, 400).astype(np.float32)
    polys = _o.np_rotated_box_to_poly(b).astype(np.float32)
    sc = rng.uniform(0.05, 1, 400).astype(np.float32)
    cases["a"] = np.concatenate([polys, sc[:, None]], 1)
    lab = rng.integers(0, 15, 400).astype(np.float32)
    off = lab * (polys.max() - polys.min() + 1)
    cases["b"] = np.concatenate([polys + off[:, None], sc[:, None]], 1).astype(np.float32)
    q = (rng.uniform(0, 60, (150, 1, 2)) + rng.uniform(-15, 15, (150, 4, 2))
         + np.array([[0, 0], [30, 0], [30, 30], [0, 30]])).astype(np.float32)
    q[::2] = q[::2, ::-1]          # clockwise ones
    q[5] = q[4]                    # identical pair
    q[7] = 12.0                    # all four vertices equal: zero area
    q[8] = 40.0
    cases["c"] = np.concatenate([q.reshape(150, 8), rng.uniform(0.05, 1, (150, 1)).astype(np.float32)], 1)
    return cases


def make_poly_nms():
    """poly_nms.npz: devPolyIoU of ops/nms_poly.py HEADER (:4-184) through the host shim (dense IoU matrices), and the
    keep lists of the greedy sweep (:195-207 restated here in NumPy on those matrices, `> thr`).  Own RNG."""
    rng = np.random.default_rng(20240606)
    tmp = tempfile.mkdtemp(prefix="jdet_shim_")
    try:
        text = module_string(os.path.join(REF_OPS, "nms_poly.py"), "HEADER")
        text = re.sub(r"#include\s*<\s*executor\.h\s*>", "", text)
        lib = build_shim_one(tmp, "polynms", SHIM + SHIM_POLY + "namespace pn {" + text + "}\nusing namespace pn;\n"
                             + WRAP_POLY)
        out = {}
        for tag, dets in poly_nms_cases(rng).items():
            order = np.argsort(-dets[:, 8], kind="stable")
            d = np.ascontiguousarray(dets[order])
            n = len(d)
            p = np.ascontiguousarray(d[:, :8])
            iou = np.zeros((n, n), np.float32)
            lib.ref_poly_iou(fp(p), n, fp(p), n, fp(iou))
            out[tag + "_dets"], out[tag + "_iou_sorted"] = dets, iou[:96, :96].copy()  # corner: keeps the file small
            for thr in (0.1, 0.5):
                removed = np.zeros(n, bool)
                keep = []
                for i in range(n):
                    if removed[i]:
                        continue
                    keep.append(i)
                    removed[i + 1:] |= iou[i, i + 1:] > np.float32(thr)
                out["%s_keep_%g" % (tag, thr)] = order[np.array(keep, np.int64)]
        np.savez_compressed(os.path.join(HERE, "poly_nms.npz"),
                            provenance=prov("seed 20240606; reference CUDA text (devPolyIoU) on host via macro shim"),
                            **out)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "poly_nms":
        make_poly_nms()
    elif len(sys.argv) > 1 and sys.argv[1] == "convex":
        import oracle as _o
        make_convex(_o.ref())
    elif len(sys.argv) > 1 and sys.argv[1] == "fr":
        make_fr()
    elif len(sys.argv) > 1 and sys.argv[1] == "rie":
        import oracle as _o
        make_rie(_o.ref())
    elif len(sys.argv) > 1 and sys.argv[1] == "rroi_v0":
        make_rroi_v0()
    else:
        main()
