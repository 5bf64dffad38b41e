// van_block.hip -- a whole VAN Block forward / backward as ONE C-ABI call each: the launch sequence of ops/van_block.py
// (13 launches forward, 25 backward: csrc/van_gemm.hip, csrc/dwconv.hip) issued from C++ into caller-provided arenas.
//
// The block of /root/reference/python/jdet/models/backbones/van.py:216-261 (Block.execute) is 38 times in a VAN-B3 step;
// issued launch by launch from Python its node cost ~0.55 ms of host time (ctypes calls, ~60 tensor allocations), 20 ms per
// Oriented R-CNN step whose GPU time is ~60 ms -- the host paced the step.  Here the host side of a block is two calls and
// three allocations (saved activations, scratch, gradients); every buffer is a slice of those arenas at offsets this file
// owns (rsdet_van_block_*_floats tell the caller how much to allocate).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <initializer_list>

#include "rsdet_api_internal.h"

namespace {

struct Dims {
  int N, C, H, W, R, P, ns, ln;
  size_t ncp, nrp;
};
static Dims dims(const rsdet_van_block* b) {
  Dims d;
  d.N = b->N, d.C = b->C, d.H = b->H, d.W = b->W, d.R = b->R, d.P = b->H * b->W;
  const int sl = rsdet_van_chan_slices(d.P);
  d.ns = d.N * sl, d.ln = d.P / sl;
  d.ncp = (size_t)d.N * d.C * d.P, d.nrp = (size_t)d.N * d.R * d.P;
  return d;
}
static inline size_t up4(size_t n) { return (n + 3) & ~(size_t)3; }   // every slice 16-byte aligned

// bump allocator over a float arena
struct Arena {
  float* base;
  size_t off;
  float* take(size_t n) {
    float* p = base ? base + off : nullptr;
    off += up4(n);
    return p;
  }
};

struct Saved {   // what the backward reads again
  float *t1, *u, *a0, *a1, *a2, *gt, *x1, *h, *d3, *h3, *st1, *st2, *w1t, *w2t, *w3t, *w4t, *w5t;
};
static size_t saved_layout(const Dims& d, float* base, Saved* s) {
  Arena a{base, 0};
  s->t1 = a.take(d.ncp), s->u = a.take(d.ncp), s->a0 = a.take(d.ncp), s->a1 = a.take(d.ncp), s->a2 = a.take(d.ncp);
  s->gt = a.take(d.ncp), s->x1 = a.take(d.ncp);
  s->h = a.take(d.nrp), s->d3 = a.take(d.nrp), s->h3 = a.take(d.nrp);
  s->st1 = a.take(4 * (size_t)d.C), s->st2 = a.take(4 * (size_t)d.C);
  s->w1t = a.take((size_t)d.C * d.C), s->w2t = a.take((size_t)d.C * d.C), s->w3t = a.take((size_t)d.C * d.C);
  s->w4t = a.take((size_t)d.C * d.R), s->w5t = a.take((size_t)d.R * d.C);
  return a.off;
}

struct Grads {   // flat parameter gradients, in the order of the node's inputs (ops/van_block.py)
  float *g1, *be1, *wp1, *bp1, *wd5, *bd5, *wd7, *bd7, *wc1, *bc1, *wp2, *bp2, *ls1, *g2, *be2, *wf1, *bf1, *wd3, *bd3, *wf2,
      *bf2, *ls2;
};
static size_t grads_layout(const Dims& d, float* base, Grads* g) {
  Arena a{base, 0};
  const size_t C = d.C, R = d.R;
  g->g1 = a.take(C), g->be1 = a.take(C), g->wp1 = a.take(C * C), g->bp1 = a.take(C), g->wd5 = a.take(25 * C), g->bd5 = a.take(C);
  g->wd7 = a.take(49 * C), g->bd7 = a.take(C), g->wc1 = a.take(C * C), g->bc1 = a.take(C), g->wp2 = a.take(C * C);
  g->bp2 = a.take(C), g->ls1 = a.take(C), g->g2 = a.take(C), g->be2 = a.take(C), g->wf1 = a.take(R * C), g->bf1 = a.take(R);
  g->wd3 = a.take(9 * R), g->bd3 = a.take(R), g->wf2 = a.take(C * R), g->bf2 = a.take(C), g->ls2 = a.take(C);
  return a.off;
}

static bool supported(const rsdet_van_block* b) {
  if (!b || b->N < 1 || b->C < 1 || b->R < 1 || b->H < 1 || b->W < 1) return false;
  const int P = b->H * b->W;
  return rsdet_van_gemm_f32_supported(b->C, b->C, P, b->N) && rsdet_van_gemm_f32_supported(b->R, b->C, P, b->N) &&
         rsdet_van_gemm_f32_supported(b->C, b->R, P, b->N) && rsdet_van_wgrad_f32_supported(b->C, b->C, P, b->N) &&
         rsdet_van_wgrad_f32_supported(b->C, b->R, P, b->N) && (long long)b->N * b->R <= 65535 && (P & 3) == 0;
}

static size_t max_part(const Dims& d) {
  const size_t a = (size_t)rsdet_van_wgrad_f32_splits(d.C, d.C, d.P, d.N) * d.C * d.C;
  const size_t b = (size_t)rsdet_van_wgrad_f32_splits(d.C, d.R, d.P, d.N) * d.C * d.R;
  return a > b ? a : b;
}
static size_t dw_ws_floats(const Dims& d, int ch) {
  size_t m = rsdet_dwconv2d_backward_data_ws_size(d.N, ch, d.H, d.W);
  for (int k : {3, 5, 7}) {
    const size_t w = rsdet_dwconv2d_backward_weight_ws_size(d.N, ch, d.H, d.W, k);
    if (w > m) m = w;
  }
  return (m + 3) / 4;
}

// The channel sums of a depthwise backward-data result (the bias gradient of the 1x1 convolution before it): with few
// tiles per plane the per-tile table goes to the fold as it is (gs_ns = slots), else the depthwise call folds it first.
// partials of the three row folds (fc2, proj_2, conv1) and of the three depthwise weight gradients (3x3 on R, 7x7 and 5x5 on C)
static size_t rows_part_floats(const Dims& d) {
  return up4((size_t)rsdet_van_wgrad_f32_splits(d.C, d.R, d.P, d.N) * d.C * d.R) +
         2 * up4((size_t)rsdet_van_wgrad_f32_splits(d.C, d.C, d.P, d.N) * d.C * d.C);
}
static size_t dw_part_one(const Dims& d, int ch, int k) {
  return up4((rsdet_dwconv2d_backward_weight_ws_size(d.N, ch, d.H, d.W, k) + 3) / 4);
}
static size_t dw_part_floats(const Dims& d) { return dw_part_one(d, d.R, 3) + dw_part_one(d, d.C, 7) + dw_part_one(d, d.C, 5); }
static int dw_slots_per_channel(const Dims& d, int ch) {
  return (int)(rsdet_dwconv2d_backward_data_ws_size(d.N, ch, d.H, d.W) / 4 / (size_t)ch);
}
static bool gs_direct(const Dims& d, int ch) { return dw_slots_per_channel(d, ch) <= 16; }
static size_t gs_floats(const Dims& d, int ch) { return gs_direct(d, ch) ? (size_t)ch * dw_slots_per_channel(d, ch) : (size_t)ch; }

// ---- a side stream for the backward's off-chain work (weight gradients, their folds, the depthwise weight gradients): the
// chain that produces grad_x is gemm -> depthwise -> gemm ...; the weight gradients hang off it and nothing downstream in
// the block waits for them.  Issued on a second stream they overlap the chain: a memory-bound depthwise weight gradient
// beside an MFMA-bound GEMM costs almost nothing, and every kernel's ~5 us of launch / prologue / epilogue hides behind the
// other stream's main phase.  One stream and seven events per process (one process per GPU), created on first use.
struct SideStream {
  hipStream_t s = nullptr;
  hipEvent_t ev[7] = {};
  bool ok = false;
};
static SideStream* side_stream() {
  static SideStream sd = [] {
    SideStream x;
    x.ok = hipStreamCreateWithFlags(&x.s, hipStreamNonBlocking) == hipSuccess;
    for (auto& e : x.ev) x.ok = x.ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    return x;
  }();
  return &sd;
}
static int g_side_on = 0;   // measured (round 6, Oriented R-CNN VAN-B3 step, same box): 60.0 ms on one stream, 61.5 ms with the side
                           // stream -- seven cross-stream edges per block cost more than the overlap returns; kept, off
// `to` waits for everything enqueued on `from` so far
static int edge(hipStream_t from, hipStream_t to, hipEvent_t ev) {
  if (from == to) return RSDET_OK;
  if (hipEventRecord(ev, from) != hipSuccess || hipStreamWaitEvent(to, ev, 0) != hipSuccess) return RSDET_ELAUNCH;
  return RSDET_OK;
}

#define VB_CHECK(expr)       \
  do {                       \
    const int rc_ = (expr);  \
    if (rc_) return rc_;     \
  } while (0)

}  // namespace

extern "C" int rsdet_van_block_supported(const rsdet_van_block* b) { return supported(b) ? 1 : 0; }
// 1: the backward's weight gradients run on a side stream beside the grad_x chain; 0 (default): everything on `stream`
extern "C" int rsdet_van_block_side_stream(int on) {
  const int prev = g_side_on;
  if (on == 0 || on == 1) g_side_on = on;
  return prev;
}
extern "C" size_t rsdet_van_block_saved_floats(const rsdet_van_block* b) {
  if (!supported(b)) return 0;
  Saved s;
  return saved_layout(dims(b), nullptr, &s);
}
extern "C" size_t rsdet_van_block_grad_floats(const rsdet_van_block* b) {
  if (!supported(b)) return 0;
  Grads g;
  return grads_layout(dims(b), nullptr, &g);
}
extern "C" size_t rsdet_van_block_forward_scratch_floats(const rsdet_van_block* b) {
  if (!supported(b)) return 0;
  const Dims d = dims(b);
  return up4((size_t)d.C * d.ns * 2) + up4((size_t)d.R * d.C) + up4(d.R) + up4(2 * (size_t)d.C);
}
extern "C" size_t rsdet_van_block_backward_scratch_floats(const rsdet_van_block* b) {
  if (!supported(b)) return 0;
  const Dims d = dims(b);
  // tab (C ns 2) x 3, partials, two hidden-width maps, five block-width maps, vectors, depthwise workspaces
  // tab (C ns 2) x 3, partials x 2 (chain / side stream), two hidden-width maps, six block-width maps, vectors, depthwise
  // workspaces x 2
  // ... and since the three row folds and the three depthwise finishing passes are issued together behind the backward:
  // their partials each in a buffer of their own
  return 3 * up4((size_t)d.C * d.ns * 2) + up4(max_part(d)) + rows_part_floats(d) + 2 * up4(d.nrp) + 6 * up4(d.ncp) +
         up4(gs_floats(d, d.R)) + up4(gs_floats(d, d.C)) + 2 * up4(6 * (size_t)d.C) +
         up4(dw_ws_floats(d, d.R > d.C ? d.R : d.C)) + dw_part_floats(d);
}

extern "C" int rsdet_van_block_forward_f32(const rsdet_van_block* b, const float* x, float* out, float* saved, float* scratch,
                                           void* stream) {
  if (!supported(b) || !x || !out || !saved || !scratch) return RSDET_EINVAL;
  const Dims d = dims(b);
  const int N = d.N, C = d.C, H = d.H, W = d.W, R = d.R, P = d.P;
  Saved s;
  saved_layout(d, saved, &s);
  Arena a{scratch, 0};
  float* tab = a.take((size_t)C * d.ns * 2);
  float* wf = a.take((size_t)R * C);      // the BatchNorm-folded weight (C x C, then R x C)
  float* bf = a.take(R);
  float* e = a.take(2 * (size_t)C);
  auto bn_fold = [&](const float* inp, const float* gamma, const float* beta, const float* w, const float* bias, int O,
                     float* stats, float* rm, float* rv, void* nbt, float eps, float mom, const float* ls, const float* b2,
                     int shortcut) -> int {
    VB_CHECK(rsdet_van_chan_reduce_f32(inp, nullptr, N, C, P, 1, tab, stream));
    rsdet_van_bn_prep f{tab, gamma, beta, w, bias, wf, bf, stats, stats + C, stats + 2 * C, stats + 3 * C, rm, rv, nbt,
                        ls, b2, e, e + C, shortcut, O, C, d.ns, d.ln, eps, mom};
    return rsdet_van_bn_prep_f32(&f, stream);
  };
  // ---- attention half
  VB_CHECK(bn_fold(x, b->g1, b->be1, b->wp1, b->bp1, C, s.st1, b->rm1, b->rv1, b->nbt1, b->eps1, b->mom1, b->ls1, b->bp2, 1));
  VB_CHECK(rsdet_van_gemm_f32(wf, x, C, C, P, N, 2, bf, nullptr, nullptr, nullptr, nullptr, nullptr, s.t1, s.u, stream));
  VB_CHECK(rsdet_dwconv2d_forward_f32(s.u, nullptr, b->wd5, b->bd5, N, C, H, W, 5, 1, s.a0, stream));
  VB_CHECK(rsdet_dwconv2d_forward_f32(s.a0, nullptr, b->wd7, b->bd7, N, C, H, W, 7, 3, s.a1, stream));
  VB_CHECK(rsdet_van_gemm_f32(b->wc1, s.a1, C, C, P, N, 3, b->bc1, nullptr, nullptr, nullptr, s.u, nullptr, s.a2, s.gt, stream));
  VB_CHECK(rsdet_van_gemm_f32(b->wp2, s.gt, C, C, P, N, 4, e, b->ls1, e + C, nullptr, x, nullptr, s.x1, nullptr, stream));
  // ---- MLP half
  VB_CHECK(bn_fold(s.x1, b->g2, b->be2, b->wf1, b->bf1, R, s.st2, b->rm2, b->rv2, b->nbt2, b->eps2, b->mom2, b->ls2, b->bf2, 0));
  VB_CHECK(rsdet_van_gemm_f32(wf, s.x1, R, C, P, N, 1, bf, nullptr, nullptr, nullptr, nullptr, nullptr, s.h, nullptr, stream));
  VB_CHECK(rsdet_dwconv2d_forward_act_f32(s.h, b->wd3, b->bd3, N, R, H, W, 3, 1, 1, s.d3, s.h3, stream));
  VB_CHECK(rsdet_van_gemm_f32(b->wf2, s.h3, C, R, P, N, 4, nullptr, b->ls2, e + C, nullptr, s.x1, nullptr, out, nullptr, stream));
  // ---- the backward-data operands: five transposes, one launch
  const float* src[5] = {b->wp1, b->wc1, b->wp2, b->wf1, b->wf2};
  const float* rs[5] = {nullptr, nullptr, b->ls1, nullptr, b->ls2};
  float* dst[5] = {s.w1t, s.w2t, s.w3t, s.w4t, s.w5t};
  const int O[5] = {C, C, C, R, C}, K[5] = {C, C, C, C, R};
  return rsdet_van_transposes_f32(5, src, rs, dst, O, K, stream);
}

extern "C" int rsdet_van_block_backward_f32(const rsdet_van_block* b, const float* x, const float* grad_out,
                                            const float* saved, float* scratch, float* grad_x, float* grads, void* stream) {
  if (!supported(b) || !x || !grad_out || !saved || !scratch || !grads) return RSDET_EINVAL;
  const Dims d = dims(b);
  const int N = d.N, C = d.C, H = d.H, W = d.W, R = d.R, P = d.P, ns = d.ns;
  Saved s;
  saved_layout(d, const_cast<float*>(saved), &s);
  Grads g;
  grads_layout(d, grads, &g);
  Arena a{scratch, 0};
  float* tabg = a.take((size_t)C * ns * 2);
  float* tabr = a.take((size_t)C * ns * 2);
  float* tab2 = a.take((size_t)C * ns * 2);
  const int S_cc = rsdet_van_wgrad_f32_splits(C, C, P, N), S_cr = rsdet_van_wgrad_f32_splits(C, R, P, N);
  float* part = a.take(max_part(d));
  float* part_r0 = a.take((size_t)S_cr * C * R);
  float* part_r1 = a.take((size_t)S_cc * C * C);
  float* part_r2 = a.take((size_t)S_cc * C * C);
  float* gh2 = a.take(d.nrp);
  float* gh = a.take(d.nrp);
  float* G = a.take(d.ncp);
  float* ga2 = a.take(d.ncp);
  float* gug = a.take(d.ncp);
  float* ga1 = a.take(d.ncp);
  float* ga0 = a.take(d.ncp);
  float* gt1 = a.take(d.ncp);
  float* gsh = a.take(gs_floats(d, R));
  float* gs1 = a.take(gs_floats(d, C));
  const bool dir_r = gs_direct(d, R), dir_c = gs_direct(d, C);
  const int ns_r = dir_r ? dw_slots_per_channel(d, R) : 1, ns_c = dir_c ? dw_slots_per_channel(d, C) : 1;
  float* vec2 = a.take(6 * (size_t)C);   // (grad_gamma, grad_beta live in `grads`) v0..v3 of norm2 / norm1
  float* vec1 = a.take(6 * (size_t)C);
  const size_t dw_fl = dw_ws_floats(d, R > C ? R : C);
  float* dws = a.take(dw_fl);
  const size_t dw_bytes = dw_fl * 4;
  float* dwp3 = a.take(dw_part_one(d, R, 3) ? dw_part_one(d, R, 3) : 4);
  float* dwp7 = a.take(dw_part_one(d, C, 7) ? dw_part_one(d, C, 7) : 4);
  float* dwp5 = a.take(dw_part_one(d, C, 5) ? dw_part_one(d, C, 5) : 4);
  const float cnt = (float)N * (float)P;
  hipStream_t M = (hipStream_t)stream;
  SideStream* sd = g_side_on ? side_stream() : nullptr;
  hipStream_t S = (sd && sd->ok) ? sd->s : M;       // (no side stream: the same order on one stream)
  hipEvent_t* ev = sd ? sd->ev : nullptr;
  auto fork = [&](int i) { return S == M ? RSDET_OK : edge(M, S, ev[i]); };
  // ================= MLP half: out = x1 + ls2 (fc2(h3) + bf2)
  VB_CHECK(rsdet_van_chan_reduce_f32(grad_out, nullptr, N, C, P, 0, tabg, M));
  VB_CHECK(fork(0));
  // (the three row folds and the three depthwise finishing passes feed nothing in the chain: they are issued as ONE launch
  //  each behind the backward, below)
  rsdet_van_rows_fold folds[3];
  VB_CHECK(rsdet_van_wgrad_f32(grad_out, s.h3, C, R, P, N, part_r0, S));                                         // side
  folds[0] = rsdet_van_rows_fold{part_r0, b->ls2, b->wf2, tabg, b->bf2, nullptr, nullptr, nullptr, g.wf2, g.bf2, g.ls2, S_cr, C, R, ns, 2, 0};
  VB_CHECK(rsdet_van_gemm_f32(s.w5t, grad_out, R, C, P, N, 6, nullptr, nullptr, nullptr, nullptr, s.d3, nullptr, gh2, nullptr, M));
  VB_CHECK(fork(1));
  VB_CHECK(rsdet_dwconv2d_backward_weight_partial_f32(gh2, s.h, nullptr, N, R, H, W, 3, 1, dwp3, dw_part_one(d, R, 3) * 4, S));  // side
  if (dir_r)
    VB_CHECK(rsdet_dwconv2d_backward_data_f32(gh2, b->wd3, N, R, H, W, 3, 1, gh, nullptr, gsh, (size_t)R * ns_r * 4, M));
  else
    VB_CHECK(rsdet_dwconv2d_backward_data_f32(gh2, b->wd3, N, R, H, W, 3, 1, gh, gsh, dws, dw_bytes, M));
  VB_CHECK(rsdet_van_wgrad_f32(s.x1, gh, C, R, P, N, part, M));
  {
    rsdet_van_bn_fold f{part, s.w4t, gsh, nullptr, nullptr, s.st2, s.st2 + C, s.st2 + 2 * C, s.st2 + 3 * C, g.wf1, g.bf1,
                        g.g2, g.be2, vec2, vec2 + C, vec2 + 2 * C, vec2 + 3 * C, S_cr, C, R, ns_r, 1, 0, cnt};
    VB_CHECK(rsdet_van_fold_bn_f32(&f, M));
  }
  VB_CHECK(rsdet_van_gemm_f32(s.w4t, gh, C, R, P, N, 4, vec2, vec2 + C, vec2 + 2 * C, vec2 + 3 * C, grad_out, s.x1, G, nullptr, M));
  // ================= attention half: x1 = x + ls1 (proj_2(gt) + bp2 + xn)
  VB_CHECK(rsdet_van_chan_reduce_f32(G, x, N, C, P, 0, tabr, M));
  VB_CHECK(fork(2));
  VB_CHECK(rsdet_van_wgrad_f32(G, s.gt, C, C, P, N, part_r1, S));                                                // side
  folds[1] = rsdet_van_rows_fold{part_r1, b->ls1, b->wp2, tabr, b->bp2, tabr, s.st1 + 2 * C, s.st1 + 3 * C, g.wp2, g.bp2, g.ls1,
                                 S_cc, C, C, ns, 2, ns};
  VB_CHECK(rsdet_van_gemm_f32(s.w3t, G, C, C, P, N, 5, nullptr, nullptr, nullptr, nullptr, s.u, s.a2, ga2, gug, M));
  VB_CHECK(fork(3));
  VB_CHECK(rsdet_van_chan_reduce_f32(ga2, nullptr, N, C, P, 0, tab2, S));                                        // side
  VB_CHECK(rsdet_van_wgrad_f32(ga2, s.a1, C, C, P, N, part_r2, S));                                              // side
  folds[2] = rsdet_van_rows_fold{part_r2, nullptr, nullptr, tab2, nullptr, nullptr, nullptr, nullptr, g.wc1, g.bc1, nullptr,
                                 S_cc, C, C, ns, 2, 0};
  VB_CHECK(rsdet_van_gemm_f32(s.w2t, ga2, C, C, P, N, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, ga1, nullptr, M));
  VB_CHECK(fork(4));
  VB_CHECK(rsdet_dwconv2d_backward_weight_partial_f32(ga1, s.a0, nullptr, N, C, H, W, 7, 3, dwp7, dw_part_one(d, C, 7) * 4, S));  // side
  VB_CHECK(rsdet_dwconv2d_backward_data_f32(ga1, b->wd7, N, C, H, W, 7, 3, ga0, nullptr, nullptr, 0, M));
  VB_CHECK(fork(5));
  VB_CHECK(rsdet_dwconv2d_backward_weight_partial_f32(ga0, s.u, nullptr, N, C, H, W, 5, 1, dwp5, dw_part_one(d, C, 5) * 4, S));   // side
  if (dir_c)
    VB_CHECK(rsdet_dwconv2d_backward_data_act_f32(ga0, b->wd5, N, C, H, W, 5, 1, gug, s.t1, gt1, nullptr, gs1,
                                                  (size_t)C * ns_c * 4, M));
  else
    VB_CHECK(rsdet_dwconv2d_backward_data_act_f32(ga0, b->wd5, N, C, H, W, 5, 1, gug, s.t1, gt1, gs1, dws, dw_bytes, M));
  VB_CHECK(rsdet_van_wgrad_f32(x, gt1, C, C, P, N, part, M));
  {
    rsdet_van_bn_fold f{part, s.w1t, gs1, tabr, b->ls1, s.st1, s.st1 + C, s.st1 + 2 * C, s.st1 + 3 * C, g.wp1, g.bp1,
                        g.g1, g.be1, vec1, vec1 + C, vec1 + 2 * C, vec1 + 3 * C, S_cc, C, C, ns_c, 1, ns, cnt};
    VB_CHECK(rsdet_van_fold_bn_f32(&f, M));
  }
  if (grad_x)
    VB_CHECK(rsdet_van_gemm_f32(s.w1t, gt1, C, C, P, N, 4, vec1, vec1 + C, vec1 + 2 * C, vec1 + 3 * C, G, x, grad_x, nullptr, M));
  // ---- the parameter gradients nothing above waited for: three row folds in one launch, three depthwise sums in one
  VB_CHECK(rsdet_van_fold_rows_multi_f32(folds, 3, S));
  {
    const void* wsp[3] = {dwp3, dwp7, dwp5};
    const int n3[3] = {N, N, N}, c3[3] = {R, C, C}, h3[3] = {H, H, H}, w3[3] = {W, W, W}, k3[3] = {3, 7, 5};
    float* gw3[3] = {g.wd3, g.wd7, g.wd5};
    float* gb3[3] = {g.bd3, g.bd7, g.bd5};
    VB_CHECK(rsdet_dwconv2d_wgrad_finish_multi_f32(3, wsp, n3, c3, h3, w3, k3, gw3, gb3, S));
  }
  // the caller's stream owns every buffer again once the side stream's work is behind it
  if (S != M) VB_CHECK(edge(S, M, ev[6]));
  return RSDET_OK;
}
