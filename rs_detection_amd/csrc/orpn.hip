// orpn.hip -- the control path of the Oriented R-CNN heads as a handful of launches.
//
// Reference: /root/reference/python/jdet/models/roi_heads/oriented_rpn_head.py:135-222 (_get_bboxes_single: per-level
// top-nms_pre, MidpointOffsetCoder.decode, min-size filter, per-level offset, horizontal NMS, first nms_post),
// models/boxes/coder.py:372-433 (MidpointOffsetCoder.decode), ops/bbox_transforms.py:501-671 (rectpoly2obb, obb2hbb,
// regular_obb, regular_theta), models/boxes/sampler.py:57-180 (RandomSampler: a uniform k-subset of the positives, then of
// the negatives).  As tensor operations those are ~280 launches per image for the proposals and ~70 per sampler call, all
// a few microseconds of work each: 1 400 launches and 7 ms of a 60 ms Oriented R-CNN / VAN-B3 step.
//
// Both need "the k largest of n" with n up to ~1e6 and k <= 2 000, with the tie order of a STABLE sort (equal values: lower
// index first).  That is an exact RADIX SELECT here, not a sort:
//   * P passes over the n keys (3 for 32-bit keys, 6 for 64-bit), many workgroups each: pass p histograms digit p (11 bits
//     from the top) of the keys whose higher digits equal the digits chosen so far, in LDS (wave-aggregated: the two most
//     frequent digits of a wave cost one atomic each -- scores and uniform draws both put most keys of a wave in one bin),
//     then adds its non-empty bins to the pass's global histogram;
//   * no pass leaves state behind except its histogram: every workgroup of pass p + 1 re-derives the digits chosen by
//     passes 0..p from those histograms (a 2 048-bin suffix search each), so there is no "last workgroup" protocol, no
//     fence and no single-workgroup launch between passes -- the kernel boundary is the only ordering used;
//   * an emit pass appends every key above the threshold to a list (exactly k - need of them) and the keys EQUAL to it to a
//     tie list; the consumer takes the `need` lowest-indexed ties (by rank counting in LDS; when the tie list overflowed --
//     a degenerate input such as all scores equal -- by an ordered scan of the input, slow and exact).
// The consumers: sampler_final (both index lists in ascending order, the reference's `.unique()` order) and
// orpn_sort_decode (one workgroup per image: decode, min-size mask, extent of the real boxes, a bitonic sort of <= 16 384
// 64-bit keys in 128 KB of LDS, sorted boxes for the NMS) + orpn_finish (keep flags -> the first nms_post rows).
#include <stdint.h>

#include "rsdet_api_internal.h"

namespace rsdet {

typedef unsigned long long u64;

constexpr int SEL_T = 256;         // threads per workgroup of the counting / emitting passes
constexpr int SEL_CHUNK = 8192;    // keys per workgroup
constexpr int SEL_BINS = 2048;     // 11-bit digits
constexpr int SEL_TIE_CAP = 2048;  // ties kept per class before the ordered-scan fallback
constexpr int SEL_MAXP = 6;

template <class K>
struct SelKey;
template <>
struct SelKey<unsigned> {
  static constexpr int BITS = 32, P = 3;
};
template <>
struct SelKey<u64> {
  static constexpr int BITS = 64, P = 6;
};
template <class K>
__device__ __forceinline__ int sel_shift(int p) {
  const int s = SelKey<K>::BITS - 11 * (p + 1);
  return s < 0 ? 0 : s;
}
template <class K>
__device__ __forceinline__ int sel_width(int p) {
  return p == SelKey<K>::P - 1 ? SelKey<K>::BITS - 11 * (SelKey<K>::P - 1) : 11;
}
template <class K>
__device__ __forceinline__ unsigned sel_digit(K key, int p) {
  return (unsigned)(key >> sel_shift<K>(p)) & ((1u << sel_width<K>(p)) - 1u);
}
// order-preserving integer images of the floating-point values
__device__ __forceinline__ unsigned sel_key(float v) {
  const unsigned u = __float_as_uint(v);
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ u64 sel_key(double v) {
  const u64 u = (u64)__double_as_longlong(v);
  return u ^ ((u >> 63) ? ~0ull : (1ull << 63));
}

// per job; all zero on entry (the entry point's memset)
template <int NC>
struct SelWs {
  unsigned hist[SEL_MAXP][NC][SEL_BINS];
  unsigned fill[NC][2];  // emitted: [c][0] keys above the threshold, [c][1] keys equal to it
  // left by workgroup 0 of the emit pass for the consumer:
  u64 thr[NC];
  unsigned need[NC], all[NC], total[NC];
};

struct SelPick {
  u64 prefix;          // the digits chosen so far (key >> shift of the last chosen digit)
  unsigned remaining;  // how many keys are still to be taken from those that match the prefix
  unsigned all;        // the class has fewer than k members: every one is taken
};

// One suffix search over a 2 048-bin histogram by the whole workgroup (T threads, T | 2048): the highest bin b with
// sum_{j >= b} hist[j] >= r.  -> bin, above = sum_{j > b}, total = sum of all bins; found = 0 when total < r.
template <int T>
__device__ __forceinline__ void sel_pick_bin(const unsigned* __restrict__ hist, unsigned r, unsigned* s_scr, unsigned& bin,
                                             unsigned& above, unsigned& total, unsigned& found) {
  constexpr int BPT = SEL_BINS / T, NW = T / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned h[BPT], v = 0;
#pragma unroll
  for (int j = 0; j < BPT; ++j) {
    h[j] = hist[tid * BPT + j];
    v += h[j];
  }
  unsigned x = v;  // inclusive suffix sum inside the wave
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned t = __shfl_down(x, off);
    if (lane + off < 64) x += t;
  }
  __syncthreads();  // (s_scr may still be read from the previous call)
  if (lane == 0) s_scr[wave] = x;
  if (tid == 0) s_scr[NW + 2] = 0u;
  __syncthreads();
  unsigned hi = 0, all = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const unsigned t = s_scr[w];
    all += t;
    if (w > wave) hi += t;
  }
  const unsigned incl = x + hi, excl = incl - v;
  if (r >= 1u && excl < r && incl >= r) {
    unsigned acc = excl;
#pragma unroll
    for (int j = BPT - 1; j >= 0; --j) {
      if (acc < r && acc + h[j] >= r) {
        s_scr[NW] = (unsigned)(tid * BPT + j);
        s_scr[NW + 1] = acc;
        s_scr[NW + 2] = 1u;
      }
      acc += h[j];
    }
  }
  __syncthreads();
  bin = s_scr[NW];
  above = s_scr[NW + 1];
  found = s_scr[NW + 2];
  total = all;
}

// The digits chosen by passes 0 .. done-1, re-derived from their histograms by every workgroup that needs them.
template <class K, int NC, int T, class Plan>
__device__ __forceinline__ void sel_resolve(const SelWs<NC>* __restrict__ ws, int done, const Plan& plan, int job,
                                            unsigned* s_scr, SelPick* pick, unsigned* total) {
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    pick[c].prefix = 0;
    pick[c].remaining = 0;
    pick[c].all = 0;
    total[c] = 0;
  }
  for (int p = 0; p < done; ++p) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (p > 0 && pick[c].all) continue;
      const unsigned r = p == 0 ? plan.k(job, c, total) : pick[c].remaining;
      unsigned bin, above, tot, found;
      sel_pick_bin<T>(ws->hist[p][c], r, s_scr, bin, above, tot, found);
      if (p == 0) total[c] = tot;
      if (r == 0u) {  // nothing wanted: a threshold no key exceeds, no ties taken
        pick[c].prefix = (pick[c].prefix << sel_width<K>(p)) | ((1u << sel_width<K>(p)) - 1u);
        pick[c].remaining = 0;
      } else if (!found) {
        pick[c].all = 1;  // (only at p == 0: fewer members than k)
      } else {
        pick[c].prefix = (pick[c].prefix << sel_width<K>(p)) | bin;
        pick[c].remaining = r - above;
      }
    }
  }
}

// one wave's keys into the LDS histogram: the wave's two most frequent digits cost one atomic each
__device__ __forceinline__ void sel_count(unsigned* hist, bool active, unsigned digit) {
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const u64 m = __ballot(active);
    if (!m) return;
    const int lead = __ffsll((long long)m) - 1;
    const unsigned d0 = __shfl(digit, lead);
    const u64 same = __ballot(active && digit == d0);
    if ((int)(threadIdx.x & 63) == lead) atomicAdd(hist + d0, (unsigned)__popcll(same));
    active = active && digit != d0;
  }
  if (active) atomicAdd(hist + digit, 1u);
}

template <class K, int NC, class Src, class Plan>
__global__ __launch_bounds__(SEL_T) void sel_hist_kernel(Src src, Plan plan, SelWs<NC>* __restrict__ wsb, int pass) {
  const int job = blockIdx.y, n = src.count(job), tid = threadIdx.x;
  const long long base = (long long)blockIdx.x * SEL_CHUNK;
  if (base >= n) return;
  SelWs<NC>* ws = wsb + job;
  __shared__ unsigned s_hist[NC][SEL_BINS];
  __shared__ unsigned s_scr[SEL_T / 64 + 4];
  SelPick pick[NC];
  unsigned total[NC];
  sel_resolve<K, NC, SEL_T>(ws, pass, plan, job, s_scr, pick, total);
  for (int i = tid; i < NC * SEL_BINS; i += SEL_T) (&s_hist[0][0])[i] = 0u;
  __syncthreads();
  const int end = (int)(base + SEL_CHUNK < n ? base + SEL_CHUNK : n);
  for (int i0 = (int)base; i0 < end; i0 += SEL_T) {  // (whole waves stay in the loop: sel_count ballots)
    const int i = i0 + tid;
    int cls = -1;
    K key = 0;
    if (i < end) src.load(job, i, cls, key);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const bool act = cls == c && !pick[c].all && (pass == 0 || (key >> sel_shift<K>(pass - 1)) == (K)pick[c].prefix);
      sel_count(s_hist[c], act, sel_digit<K>(key, pass));
    }
  }
  __syncthreads();
  for (int i = tid; i < NC * SEL_BINS; i += SEL_T) {
    const unsigned v = (&s_hist[0][0])[i];
    if (v) atomicAdd(&ws->hist[pass][0][0] + i, v);
  }
}

template <class K, int NC, class Src, class Plan>
__global__ __launch_bounds__(SEL_T) void sel_emit_kernel(Src src, Plan plan, SelWs<NC>* __restrict__ wsb,
                                                         u64* __restrict__ above, int cap, unsigned* __restrict__ ties) {
  const int job = blockIdx.y, n = src.count(job), tid = threadIdx.x, lane = tid & 63;
  const long long base = (long long)blockIdx.x * SEL_CHUNK;
  if (base >= n) return;
  SelWs<NC>* ws = wsb + job;
  __shared__ unsigned s_scr[SEL_T / 64 + 4];
  SelPick pick[NC];
  unsigned total[NC];
  sel_resolve<K, NC, SEL_T>(ws, SelKey<K>::P, plan, job, s_scr, pick, total);
  if (blockIdx.x == 0 && tid == 0) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      ws->thr[c] = pick[c].prefix;
      ws->need[c] = pick[c].all ? 0u : pick[c].remaining;
      ws->all[c] = pick[c].all;
      ws->total[c] = total[c];
    }
  }
  const int end = (int)(base + SEL_CHUNK < n ? base + SEL_CHUNK : n);
  const u64 lt = lane ? (~0ull >> (64 - lane)) : 0ull;
  for (int i0 = (int)base; i0 < end; i0 += SEL_T) {
    const int i = i0 + tid;
    int cls = -1;
    K key = 0;
    if (i < end) src.load(job, i, cls, key);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const bool mine = cls == c;
      const bool up = mine && (pick[c].all || key > (K)pick[c].prefix);
      const bool eq = mine && !pick[c].all && key == (K)pick[c].prefix;
      const u64 mu = __ballot(up), me = __ballot(eq);
      if (mu) {
        unsigned b = 0;
        if (lane == 0) b = atomicAdd(&ws->fill[c][0], (unsigned)__popcll(mu));
        b = __shfl(b, 0) + (unsigned)__popcll(mu & lt);
        if (up && b < (unsigned)cap) above[((long long)job * NC + c) * cap + b] = ((u64)(unsigned)key << 32) | (unsigned)i;
      }
      if (me) {
        unsigned b = 0;
        if (lane == 0) b = atomicAdd(&ws->fill[c][1], (unsigned)__popcll(me));
        b = __shfl(b, 0) + (unsigned)__popcll(me & lt);
        if (eq && b < (unsigned)SEL_TIE_CAP) ties[((long long)job * NC + c) * SEL_TIE_CAP + b] = (unsigned)i;
      }
    }
  }
}

// The `need` lowest-indexed keys equal to the threshold of class c, written to dst[0 .. need) in ASCENDING index order;
// the whole workgroup (T threads) calls it.  ties: the emit pass's list (n_ties entries were seen, at most SEL_TIE_CAP
// stored); when it overflowed, the input itself is scanned in index order.
template <class K, int NC, int T, class Src>
__device__ __forceinline__ void sel_take_ties(const Src& src, int job, int c, K thr, unsigned need, unsigned n_ties,
                                              const unsigned* __restrict__ ties, unsigned* s_tie, unsigned* s_w,
                                              unsigned* dst) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (need == 0u) return;
  if (n_ties <= (unsigned)SEL_TIE_CAP) {
    __syncthreads();
    for (unsigned t = tid; t < n_ties; t += T) s_tie[t] = ties[t];
    __syncthreads();
    for (unsigned t = tid; t < n_ties; t += T) {
      const unsigned me = s_tie[t];
      unsigned rank = 0;
      for (unsigned u = 0; u < n_ties; ++u) rank += s_tie[u] < me ? 1u : 0u;
      if (rank < need) dst[rank] = me;
    }
    __syncthreads();
    return;
  }
  const int n = src.count(job);
  const u64 lt = lane ? (~0ull >> (64 - lane)) : 0ull;
  unsigned base = 0;
  for (int i0 = 0; i0 < n && base < need; i0 += T) {
    const int i = i0 + tid;
    int cls = -1;
    K key = 0;
    if (i < n) src.load(job, i, cls, key);
    const bool eq = cls == c && key == thr;
    const u64 m = __ballot(eq);
    __syncthreads();
    if (lane == 0) s_w[wave] = (unsigned)__popcll(m);
    __syncthreads();
    unsigned off = 0, tot = 0;
    for (int w = 0; w < T / 64; ++w) {
      const unsigned t = s_w[w];
      tot += t;
      if (w < wave) off += t;
    }
    const unsigned r = base + off + (unsigned)__popcll(m & lt);
    if (eq && r < need) dst[r] = (unsigned)i;
    base += tot;
  }
  __syncthreads();
}

// in-place bitonic sort of NP (a power of two) values in LDS by T threads; DESC: descending
template <class V, int T, bool DESC>
__device__ __forceinline__ void lds_bitonic(V* s, int NP) {
  const int tid = threadIdx.x;
  for (int k = 2; k <= NP; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (int t = tid; t < (NP >> 1); t += T) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
        const V a = s[i], b = s[l];
        const bool up = ((i & k) == 0) != DESC;
        if ((a > b) == up) {
          s[i] = b;
          s[l] = a;
        }
      }
    }
  }
  __syncthreads();
}

// ============================ RandomSampler on masks (sampler.py:57-180) ============================================
struct SamplerSrc {
  const int* gt_inds;          // (n_props) assigned gt index + 1, 0 = negative, -1 = ignore
  const unsigned char* valid;  // (n_props) or NULL: rows that are padding are neither positive nor negative
  const void* pri;             // (k_gt + n_props) one draw per candidate, float or double
  int k_gt, n, f64;
  __device__ __forceinline__ int count(int) const { return n; }
  __device__ __forceinline__ int gt_of(int i) const {
    if (i < k_gt) return i + 1;
    return (valid == nullptr || valid[i - k_gt]) ? gt_inds[i - k_gt] : -1;
  }
  __device__ __forceinline__ void load(int, int i, int& cls, unsigned& key) const {
    const int g = gt_of(i);
    cls = g > 0 ? 0 : (g == 0 ? 1 : -1);
    key = sel_key(static_cast<const float*>(pri)[i]);
  }
  __device__ __forceinline__ void load(int, int i, int& cls, u64& key) const {
    const int g = gt_of(i);
    cls = g > 0 ? 0 : (g == 0 ? 1 : -1);
    key = sel_key(static_cast<const double*>(pri)[i]);
  }
};

struct SamplerPlan {
  int kp, kn, num;  // kp = min(int(num * pos_fraction), n), kn = min(num, n)
  float ub;         // neg_pos_ub (< 0: none)
  __device__ __forceinline__ unsigned k(int, int c, const unsigned* total) const {
    if (c == 0) return (unsigned)kp;
    const unsigned npos = total[0] < (unsigned)kp ? total[0] : (unsigned)kp;
    long long quota = (long long)num - (long long)npos;
    if (ub >= 0.f) {
      const long long q2 = (long long)(ub * (float)(npos ? npos : 1u));
      quota = quota < q2 ? quota : q2;
    }
    if (quota < 0) quota = 0;
    if (quota > kn) quota = kn;
    return (unsigned)quota;
  }
};

template <class K>
__global__ __launch_bounds__(1024) void sampler_final_kernel(SamplerSrc src, const SelWs<2>* __restrict__ ws,
                                                             const u64* __restrict__ above, const unsigned* __restrict__ ties,
                                                             int cap, int num, long long* __restrict__ inds,
                                                             unsigned char* __restrict__ is_pos, unsigned char* __restrict__ val,
                                                             long long* __restrict__ assigned, long long* __restrict__ counts) {
  constexpr int T = 1024;
  __shared__ unsigned s_list[2][1024];
  __shared__ unsigned s_tie[SEL_TIE_CAP];
  __shared__ unsigned s_w[T / 64];
  const int tid = threadIdx.x;
  unsigned cnt[2];
  for (int c = 0; c < 2; ++c) {
    s_list[c][tid] = 0xFFFFFFFFu;
    __syncthreads();
    const unsigned na = ws->fill[c][0] < (unsigned)cap ? ws->fill[c][0] : (unsigned)cap;
    const unsigned need = ws->need[c], nt = ws->fill[c][1];
    if ((unsigned)tid < na) s_list[c][tid] = (unsigned)above[(long long)c * cap + tid];
    sel_take_ties<K, 2, T>(src, 0, c, (K)ws->thr[c], need, nt, ties + (long long)c * SEL_TIE_CAP, s_tie, s_w,
                           s_list[c] + na);
    cnt[c] = na + need;
    lds_bitonic<unsigned, T, false>(s_list[c], 1024);
  }
  const unsigned np = cnt[0], nn = cnt[1];
  if (tid < num) {
    const unsigned s = (unsigned)tid;
    unsigned idx = 0;
    unsigned char p = 0, v = 0;
    if (s < np) {
      idx = s_list[0][s];
      p = v = 1;
    } else if (s < np + nn) {
      idx = s_list[1][s - np];
      v = 1;
    }
    inds[s] = (long long)idx;
    is_pos[s] = p;
    val[s] = v;
    const int g = src.gt_of((int)idx) - 1;
    assigned[s] = g > 0 ? (long long)g : 0ll;
  }
  if (tid == 0) {
    counts[0] = (long long)np;
    counts[1] = (long long)nn;
  }
}

// ============================ Oriented RPN proposals (oriented_rpn_head.py:135-222) =================================
struct F6 {
  float v[6];
};
struct F5v {
  float v[5];
};

// regular_theta(t) = remainder(t + pi/2, pi) - pi/2 with Python-style remainder (bbox_transforms.py:501-505)
__device__ __forceinline__ float orpn_regular_theta(float t) {
  const float half_pi = 1.57079632679489661923f, pi = 3.14159265358979323846f;
  const float x = t + half_pi;  // (theta - start, start = -pi/2)
  float r = fmodf(x, pi);
  if (r != 0.f && r < 0.f) r += pi;
  return r + (-half_pi);
}

// MidpointOffsetCoder.decode for one anchor (coder.py:372-433) followed by rectpoly2obb + regular_obb
// (bbox_transforms.py:525-548, :507-514): hbb anchor a (x1, y1, x2, y2) + 6 deltas -> obb (x, y, w, h, theta), w >= h.
__device__ __forceinline__ void orpn_decode(const float* a, const float* dl, const F6& mean, const F6& stdv, float max_ratio,
                                            float* o) {
  float d[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) d[k] = dl[k] * stdv.v[k] + mean.v[k];
  const float dx = d[0], dy = d[1];
  const float dw = fminf(fmaxf(d[2], -max_ratio), max_ratio), dh = fminf(fmaxf(d[3], -max_ratio), max_ratio);
  const float px = (a[0] + a[2]) * 0.5f, py = (a[1] + a[3]) * 0.5f, pw = a[2] - a[0], ph = a[3] - a[1];
  const float gw = pw * expf(dw), gh = ph * expf(dh);
  const float gx = px + pw * dx, gy = py + ph * dy;
  const float x1 = gx - gw * 0.5f, y1 = gy - gh * 0.5f, x2 = gx + gw * 0.5f, y2 = gy + gh * 0.5f;
  const float da = fminf(fmaxf(d[4], -0.5f), 0.5f), db = fminf(fmaxf(d[5], -0.5f), 0.5f);
  const float ga = gx + da * gw, ga_ = gx - da * gw, gb = gy + db * gh, gb_ = gy - db * gh;
  float cx[4] = {ga - gx, x2 - gx, ga_ - gx, x1 - gx}, cy[4] = {y1 - gy, gb - gy, y2 - gy, gb_ - gy};
  float diag[4], dmax = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    diag[k] = sqrtf(cx[k] * cx[k] + cy[k] * cy[k]);
    dmax = k == 0 ? diag[0] : fmaxf(dmax, diag[k]);
  }
  float qx[4], qy[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float sc = dmax / diag[k];
    qx[k] = cx[k] * sc + gx;
    qy[k] = cy[k] * sc + gy;
  }
  // rectpoly2obb
  const float theta = atan2f(-(qy[1] - qy[0]), qx[1] - qx[0]);
  const float Cos = cosf(theta), Sin = sinf(theta);
  const float x = (qx[0] + qx[1] + qx[2] + qx[3]) / 4.f, y = (qy[0] + qy[1] + qy[2] + qy[3]) / 4.f;
  float r0min = 0.f, r0max = 0.f, r1min = 0.f, r1max = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float ux = qx[k] - x, uy = qy[k] - y;
    const float r0 = ux * Cos + uy * (-Sin), r1 = ux * Sin + uy * Cos;
    r0min = k == 0 ? r0 : fminf(r0min, r0);
    r0max = k == 0 ? r0 : fmaxf(r0max, r0);
    r1min = k == 0 ? r1 : fminf(r1min, r1);
    r1max = k == 0 ? r1 : fmaxf(r1max, r1);
  }
  const float w = r0max - r0min, h = r1max - r1min;
  const bool wide = w > h;
  o[0] = x;
  o[1] = y;
  o[2] = wide ? w : h;
  o[3] = wide ? h : w;
  o[4] = orpn_regular_theta(wide ? theta : theta + 1.57079632679489661923f);
}

// obb2hbb (bbox_transforms.py:572-578)
__device__ __forceinline__ void orpn_obb2hbb(const float* b, float* o) {
  const float Cos = cosf(b[4]), Sin = sinf(b[4]);
  const float bx = fabsf(b[2] / 2.f * Cos) + fabsf(b[3] / 2.f * Sin), by = fabsf(b[2] / 2.f * Sin) + fabsf(b[3] / 2.f * Cos);
  o[0] = b[0] - bx;
  o[1] = b[1] - by;
  o[2] = b[0] + bx;
  o[3] = b[1] + by;
}

__global__ void orpn_decode_kernel(const float* __restrict__ anchors, const float* __restrict__ deltas, int n, F6 mean,
                                   F6 stdv, float max_ratio, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float a[4], d[6], o[5];
#pragma unroll
  for (int k = 0; k < 4; ++k) a[k] = anchors[(long long)i * 4 + k];
#pragma unroll
  for (int k = 0; k < 6; ++k) d[k] = deltas[(long long)i * 6 + k];
  orpn_decode(a, d, mean, stdv, max_ratio, o);
#pragma unroll
  for (int k = 0; k < 5; ++k) out[(long long)i * 5 + k] = o[k];
}

__global__ void orpn_obb2hbb_kernel(const float* __restrict__ obb, int n, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float b[5], o[4];
#pragma unroll
  for (int k = 0; k < 5; ++k) b[k] = obb[(long long)i * 5 + k];
  orpn_obb2hbb(b, o);
#pragma unroll
  for (int k = 0; k < 4; ++k) out[(long long)i * 4 + k] = o[k];
}

struct LevelSrc {  // job = image * L + level; element i of a job = pixel * A + a: the score maps lie PIXEL-MAJOR (N, H, W, A),
                   // the reference's flattening (cls.permute(1, 2, 0).reshape(-1)), so that ties go to the lower index of THAT
  const float* score[8];
  int cnt[8];  // A * HW
  int L;
  __device__ __forceinline__ int count(int job) const { return cnt[job % L]; }
  __device__ __forceinline__ void load(int job, int i, int& cls, unsigned& key) const {
    const int l = job % L, b = job / L;
    cls = 0;
    key = sel_key(score[l][(long long)b * cnt[l] + i]);
  }
};

struct LevelPlan {
  int nms_pre;
  __device__ __forceinline__ unsigned k(int, int, const unsigned*) const { return (unsigned)nms_pre; }
};

struct OrpnArgs {
  LevelSrc src;
  const float* reg[8];      // (N, A * 6, H, W)
  const float* anchors[8];  // (HW * A, 4), index pixel * A + a
  int hw[8];
  int A, nms_pre, n_tot;
  F6 mean, stdv;
  float max_ratio, min_size;
};

constexpr int ORPN_NP = 16384;  // keys sorted per image (128 KB of LDS)

// composite sort key: (score bits + 1 for a real box, 0 for a too-small one) : 7 - level : 0x1FFFFF - index in the level
// (index = pixel * A + a, the reference's flattening) -- descending order of it is the reference's stable argsort of
// where(ok, score, -1) over the level-major list.
__device__ __forceinline__ void orpn_unpack(u64 key, int& l, int& orig) {
  l = 7 - (int)((key >> 21) & 7u);
  orig = 0x1FFFFF - (int)(key & 0x1FFFFFu);
}

__device__ __forceinline__ float orpn_box_at(const OrpnArgs& a, int b, int l, int orig, float* obb, float* hbb, bool& ok) {
  const int A = a.A, hw = a.hw[l], pix = orig / A, an = orig - pix * A;
  float av[4], d[6];
#pragma unroll
  for (int k = 0; k < 4; ++k) av[k] = a.anchors[l][(long long)orig * 4 + k];
  const float* r = a.reg[l] + ((long long)b * A * 6 + an * 6) * hw + pix;
#pragma unroll
  for (int k = 0; k < 6; ++k) d[k] = r[(long long)k * hw];
  orpn_decode(av, d, a.mean, a.stdv, a.max_ratio, obb);
  orpn_obb2hbb(obb, hbb);
  ok = a.min_size < 0.f || (obb[2] > a.min_size && obb[3] > a.min_size);
  return a.src.score[l][(long long)b * a.src.cnt[l] + orig];
}

__global__ __launch_bounds__(1024) void orpn_sort_decode_kernel(OrpnArgs a, const SelWs<1>* __restrict__ wsb,
                                                                const u64* __restrict__ above,
                                                                const unsigned* __restrict__ ties, float* __restrict__ dets,
                                                                float* __restrict__ boxes, unsigned char* __restrict__ okf) {
  constexpr int T = 1024;
  extern __shared__ u64 s_key[];  // ORPN_NP
  __shared__ unsigned s_tie[SEL_TIE_CAP];
  __shared__ unsigned s_sel[SEL_TIE_CAP];
  __shared__ unsigned s_w[T / 64];
  __shared__ float s_red[2][T / 64];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, L = a.src.L;
  // ---- 1. the selected keys of every level -> composite keys
  int at = 0;
  for (int l = 0; l < L; ++l) {
    const int job = b * L + l;
    const SelWs<1>* ws = wsb + job;
    const unsigned na = ws->fill[0][0] < (unsigned)a.nms_pre ? ws->fill[0][0] : (unsigned)a.nms_pre;
    const unsigned need = ws->need[0], nt = ws->fill[0][1];
    const unsigned thr = (unsigned)ws->thr[0];
    sel_take_ties<unsigned, 1, T>(a.src, job, 0, thr, need, nt, ties + (long long)job * SEL_TIE_CAP, s_tie, s_w, s_sel);
    for (unsigned t = tid; t < na + need; t += T) {
      unsigned e, kbits;
      if (t < na) {
        const u64 v = above[(long long)job * a.nms_pre + t];
        e = (unsigned)v;
        kbits = (unsigned)(v >> 32);
      } else {
        e = s_sel[t - na];
        kbits = thr;
      }
      const unsigned orig = e;  // (the score maps lie pixel-major: an element's position IS the reference's index)
      const unsigned sb = kbits ^ 0x80000000u;  // scores are >= 0: the float's own bits
      s_key[at + t] = ((u64)(sb + 1u) << 24) | ((u64)(7 - l) << 21) | (u64)(0x1FFFFFu - orig);
    }
    at += (int)(na + need);
    __syncthreads();
  }
  const int n = at;  // == a.n_tot
  for (int t = n + tid; t < ORPN_NP; t += T) s_key[t] = 0ull;
  // ---- 2. decode once for the min-size mask and the extent of the real boxes
  float vmax = -INFINITY, vmin = INFINITY;
  for (int t = tid; t < n; t += T) {
    int l, orig;
    orpn_unpack(s_key[t], l, orig);
    float obb[5], hbb[4];
    bool ok;
    orpn_box_at(a, b, l, orig, obb, hbb, ok);
    if (ok) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        vmax = fmaxf(vmax, hbb[k]);
        vmin = fminf(vmin, hbb[k]);
      }
    } else {
      s_key[t] &= 0xFFFFFFull;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    vmax = fmaxf(vmax, __shfl_xor(vmax, off));
    vmin = fminf(vmin, __shfl_xor(vmin, off));
  }
  if (lane == 0) {
    s_red[0][wave] = vmax;
    s_red[1][wave] = vmin;
  }
  // ---- 3. sort (lds_bitonic opens with a barrier)
  lds_bitonic<u64, T, true>(s_key, ORPN_NP);
  vmax = s_red[0][0];
  vmin = s_red[1][0];
  for (int w = 1; w < T / 64; ++w) {
    vmax = fmaxf(vmax, s_red[0][w]);
    vmin = fminf(vmin, s_red[1][w]);
  }
  const float step = (vmax - vmin) + 1.f;  // max_coordinate + 1
  // ---- 4. the sorted list: (obb, score) rows, offset horizontal boxes for the NMS, the min-size mask
  for (int t = tid; t < n; t += T) {
    int l, orig;
    orpn_unpack(s_key[t], l, orig);
    float obb[5], hbb[4];
    bool ok;
    const float sc = orpn_box_at(a, b, l, orig, obb, hbb, ok);
    float* d = dets + ((long long)b * a.n_tot + t) * 6;
#pragma unroll
    for (int k = 0; k < 5; ++k) d[k] = obb[k];
    d[5] = sc;
    const float off = (float)l * step;
    float* h = boxes + ((long long)b * a.n_tot + t) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) h[k] = hbb[k] + off;
    okf[(long long)b * a.n_tot + t] = ok ? 1 : 0;
  }
}

// keep flags (sorted order) -> the first P kept real boxes, zero rows behind them (oriented_rpn_head.py:219-222)
__global__ __launch_bounds__(1024) void orpn_finish_kernel(const unsigned char* __restrict__ keep,
                                                           const unsigned char* __restrict__ okf,
                                                           const float* __restrict__ dets, int n, int P,
                                                           float* __restrict__ out, unsigned char* __restrict__ flags) {
  constexpr int T = 1024;
  __shared__ unsigned s_w[T / 64];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  keep += (long long)b * n;
  okf += (long long)b * n;
  dets += (long long)b * n * 6;
  out += (long long)b * P * 6;
  flags += (long long)b * P;
  const u64 lt = lane ? (~0ull >> (64 - lane)) : 0ull;
  unsigned base = 0;
  for (int i0 = 0; i0 < n && base < (unsigned)P; i0 += T) {
    const int i = i0 + tid;
    const bool kept = i < n && keep[i] && okf[i];
    const u64 m = __ballot(kept);
    __syncthreads();
    if (lane == 0) s_w[wave] = (unsigned)__popcll(m);
    __syncthreads();
    unsigned off = 0, tot = 0;
    for (int w = 0; w < T / 64; ++w) {
      const unsigned t = s_w[w];
      tot += t;
      if (w < wave) off += t;
    }
    const unsigned slot = base + off + (unsigned)__popcll(m & lt);
    if (kept && slot < (unsigned)P) {
#pragma unroll
      for (int k = 0; k < 6; ++k) out[(long long)slot * 6 + k] = dets[(long long)i * 6 + k];
      flags[slot] = 1;
    }
    base += tot;
  }
  if (base > (unsigned)P) base = (unsigned)P;
  for (int s = (int)base + tid; s < P; s += T) {
#pragma unroll
    for (int k = 0; k < 6; ++k) out[(long long)s * 6 + k] = 0.f;
    flags[s] = 0;
  }
}


// ============================ the Oriented RPN's losses on the SAMPLES (oriented_rpn_head.py:274-480) ===============
// The reference builds four dense target maps per image (labels, label weights, encoded boxes, box weights over all
// 611 072 anchors), cuts them into levels and evaluates both losses densely -- with weights that are zero outside the
// <= 256 sampled anchors of each image.  The same sums over the samples alone: one workgroup looks every sample up
// (level, pixel, anchor), encodes the positives' targets (MidpointOffsetCoder.encode, coder.py:334-370), evaluates the
// weighted BCE-with-logits and smooth-L1 terms, reduces them per level, and leaves the per-sample derivatives for a
// backward pass that scatters them into zero gradient maps.
struct OrpnLossArgs {
  const float* cls[8];  // forward: (N, A, H, W) logits;          backward: gradient maps (written)
  const float* reg[8];  // forward: (N, 6 A, H, W) predictions;   backward: gradient maps (written)
  int hw[8];
  int n_img, L, A, num;
  const float* anchors;      // (total, 4) level-major
  const long long* inside;   // (n_inside) or NULL
  const float* gt[16];
  int k_gt[16];
  const long long* inds;
  const unsigned char* is_pos;
  const unsigned char* val;
  const long long* assigned;
  const long long* counts;
  F6 mean, stdv;
  float beta, w_cls, w_box, pos_weight;
};

constexpr int ORPN_REC = 12;  // words per sample: level, cls offset, reg offset, hw, dcls, dreg[6], pad

// MidpointOffsetCoder.encode (coder.py:334-370): hbb proposal p, obb ground truth g -> 6 normalised deltas
__device__ __forceinline__ void orpn_encode(const float* p, const float* g, const F6& mean, const F6& stdv, float* o) {
  const float px = (p[0] + p[2]) * 0.5f, py = (p[1] + p[3]) * 0.5f, pw = p[2] - p[0], ph = p[3] - p[1];
  float hbb[4];
  orpn_obb2hbb(g, hbb);
  const float Cos = cosf(g[4]), Sin = sinf(g[4]);
  const float v1x = g[2] / 2.f * Cos, v1y = -g[2] / 2.f * Sin, v2x = -g[3] / 2.f * Sin, v2y = -g[3] / 2.f * Cos;
  const float qx[4] = {g[0] + v1x + v2x, g[0] + v1x - v2x, g[0] - v1x - v2x, g[0] - v1x + v2x};
  const float qy[4] = {g[1] + v1y + v2y, g[1] + v1y - v2y, g[1] - v1y - v2y, g[1] - v1y + v2y};
  const float gx = (hbb[0] + hbb[2]) * 0.5f, gy = (hbb[1] + hbb[3]) * 0.5f, gw = hbb[2] - hbb[0], gh = hbb[3] - hbb[1];
  const float y_min = fminf(fminf(qy[0], qy[1]), fminf(qy[2], qy[3]));
  const float x_max = fmaxf(fmaxf(qx[0], qx[1]), fmaxf(qx[2], qx[3]));
  float ga = -INFINITY, gb = -INFINITY;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    ga = fmaxf(ga, fabsf(qy[k] - y_min) > 0.1f ? -1000.f : qx[k]);
    gb = fmaxf(gb, fabsf(qx[k] - x_max) > 0.1f ? -1000.f : qy[k]);
  }
  const float d[6] = {(gx - px) / pw, (gy - py) / ph, logf(gw / pw), logf(gh / ph), (ga - gx) / gw, (gb - gy) / gh};
#pragma unroll
  for (int k = 0; k < 6; ++k) o[k] = (d[k] - mean.v[k]) / stdv.v[k];
}

__global__ __launch_bounds__(1024) void orpn_loss_fwd_kernel(OrpnLossArgs a, float* __restrict__ losses,
                                                             float* __restrict__ rec) {
  constexpr int T = 1024;
  __shared__ float s_part[2][8][T / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, S = a.n_img * a.num;
  float avg = 0.f;
  {
    long long np = 0, nn = 0;
    for (int b = 0; b < a.n_img; ++b) {
      const long long p = a.counts[2 * b], n = a.counts[2 * b + 1];
      np += p > 1 ? p : 1;
      nn += n > 1 ? n : 1;
    }
    avg = (float)(np + nn);
  }
  float acc_c[8], acc_b[8];
#pragma unroll
  for (int l = 0; l < 8; ++l) acc_c[l] = acc_b[l] = 0.f;
  for (int s = tid; s < S; s += T) {
    const int b = s / a.num;
    int level = -1, cls_off = 0, reg_off = 0, hw = 0;
    float dcls = 0.f, dreg[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (a.val[s]) {
      const long long i = a.inds[s], flat = a.inside ? a.inside[i] : i;
      long long start = 0;
      int l = 0;
      for (; l < a.L - 1; ++l) {
        const long long c = (long long)a.A * a.hw[l];
        if (flat < start + c) break;
        start += c;
      }
      const int j = (int)(flat - start);
      hw = a.hw[l];
      const int pix = j / a.A, an = j - pix * a.A;
      level = l;
      cls_off = (b * a.A + an) * hw + pix;
      reg_off = (b * a.A + an) * 6 * hw + pix;
      const bool pos = a.is_pos[s] != 0;
      const float x = a.cls[l][cls_off], y = pos ? 1.f : 0.f;
      const float w = (pos && a.pos_weight > 0.f) ? a.pos_weight : 1.f;
      // binary_cross_entropy_with_logits: (1 - y) x - log_sigmoid(x), log_sigmoid(x) = min(x, 0) - log1p(exp(-|x|))
      const float bce = (1.f - y) * x - (fminf(x, 0.f) - log1pf(expf(-fabsf(x))));
      acc_c[l] += w * bce;
      dcls = w * (1.f / (1.f + expf(-x)) - y) / avg * a.w_cls;
      if (pos) {
        float anc[4], g[5], t[6];
#pragma unroll
        for (int k = 0; k < 4; ++k) anc[k] = a.anchors[flat * 4 + k];
        const float* gp = a.gt[b] + a.assigned[s] * 5;
#pragma unroll
        for (int k = 0; k < 5; ++k) g[k] = gp[k];
        orpn_encode(anc, g, a.mean, a.stdv, t);
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          const float df = a.reg[l][reg_off + k * hw] - t[k], ad = fabsf(df);
          sum += ad < a.beta ? 0.5f * ad * ad / a.beta : ad - 0.5f * a.beta;
          const float gk = ad < a.beta ? df / a.beta : (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f));
          dreg[k] = gk / avg * a.w_box;
        }
        acc_b[l] += sum;
      }
    }
    float* r = rec + (long long)s * ORPN_REC;
    r[0] = __int_as_float(level);
    r[1] = __int_as_float(cls_off);
    r[2] = __int_as_float(reg_off);
    r[3] = __int_as_float(hw);
    r[4] = dcls;
#pragma unroll
    for (int k = 0; k < 6; ++k) r[5 + k] = dreg[k];
    r[11] = 0.f;
  }
#pragma unroll
  for (int l = 0; l < 8; ++l) {
    float c = acc_c[l], bx = acc_b[l];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      c += __shfl_xor(c, off);
      bx += __shfl_xor(bx, off);
    }
    if (lane == 0) {
      s_part[0][l][wave] = c;
      s_part[1][l][wave] = bx;
    }
  }
  __syncthreads();
  if (tid < 2 * a.L) {
    const int which = tid / a.L, l = tid - which * a.L;
    float v = 0.f;
    for (int w = 0; w < T / 64; ++w) v += s_part[which][l][w];
    losses[tid] = v / avg * (which ? a.w_box : a.w_cls);
  }
}

__global__ void orpn_loss_bwd_kernel(OrpnLossArgs a, const float* __restrict__ rec, const float* __restrict__ g) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= a.n_img * a.num) return;
  const float* r = rec + (long long)s * ORPN_REC;
  const int l = __float_as_int(r[0]);
  if (l < 0) return;
  const int cls_off = __float_as_int(r[1]), reg_off = __float_as_int(r[2]), hw = __float_as_int(r[3]);
  const float gc = g[l], gb = g[a.L + l];
  const_cast<float*>(a.cls[l])[cls_off] = r[4] * gc;
#pragma unroll
  for (int k = 0; k < 6; ++k) const_cast<float*>(a.reg[l])[reg_off + k * hw] = r[5 + k] * gb;
}


// ============================ OrientedHead: the sampled RoIs and their targets (oriented_head.py:426-496, :566-588) =
// One image: the `num` sampled rows of the gt-extended proposal list -> RoIs (image index, obb), labels, label weights,
// OrientedDeltaXYWHTCoder.encode targets (coder.py:447-470) and their weights, written into the image's rows of the batch.
struct RoiTargetArgs {
  const float* props;  // (n_props, stride) rows starting with (x, y, w, h, theta)
  const float* gt;     // (k_gt, 5)
  const long long* gt_labels;
  const long long* inds;
  const unsigned char* is_pos;
  const unsigned char* val;
  const long long* assigned;
  int stride, k_gt, num, image, num_classes;
  F5v mean, stdv;
  float pos_weight;
  float* rois;
  long long* labels;
  float* label_weights;
  float* bbox_targets;
  float* bbox_weights;
};

__global__ void roi_targets_kernel(RoiTargetArgs a) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= a.num) return;
  const long long idx = a.inds[s];
  const float* bp = idx < a.k_gt ? a.gt + idx * 5 : a.props + (idx - a.k_gt) * a.stride;
  float p[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) p[k] = bp[k];
  a.rois[s * 6] = (float)a.image;
#pragma unroll
  for (int k = 0; k < 5; ++k) a.rois[s * 6 + 1 + k] = p[k];
  const bool pos = a.is_pos[s] != 0, used = a.val[s] != 0;
  a.labels[s] = pos ? a.gt_labels[a.assigned[s]] : (long long)a.num_classes;
  a.label_weights[s] = used ? ((pos && a.pos_weight > 0.f) ? a.pos_weight : 1.f) : 0.f;
  float t[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  if (pos) {
    const float* g = a.gt + a.assigned[s] * 5;
    const float half_pi = 1.57079632679489661923f;
    const float d1 = orpn_regular_theta(g[4] - p[4]), d2 = orpn_regular_theta(g[4] - p[4] + half_pi);
    const bool first = fabsf(d1) < fabsf(d2);
    const float gw = first ? g[2] : g[3], gh = first ? g[3] : g[2], dth = first ? d1 : d2;
    const float c = cosf(-p[4]), sn = sinf(-p[4]);
    const float ex = g[0] - p[0], ey = g[1] - p[1];
    const float d[5] = {(c * ex + sn * ey) / p[2], (-sn * ex + c * ey) / p[3], logf(gw / p[2]), logf(gh / p[3]), dth};
#pragma unroll
    for (int k = 0; k < 5; ++k) t[k] = (d[k] - a.mean.v[k]) / a.stdv.v[k];
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    a.bbox_targets[s * 5 + k] = t[k];
    a.bbox_weights[s * 5 + k] = pos ? 1.f : 0.f;
  }
}


// ============================ MaxIoUAssigner on horizontal boxes without the (K, A) matrix ==========================
// The Oriented RPN assigns 611 072 horizontal anchors to the K horizontal hulls of an image's ground truth
// (oriented_rpn_head.py:292-300 -> assigner.py:65-170 with BboxOverlaps2D, iou_calculator.py:164-257).  As
// rsdet_bbox_overlaps_f32 + rsdet_assign_wrt_overlaps_f32 that is a K x A fp32 matrix written once and read twice (977 MB at
// K = 400): 0.6 ms of the Oriented R-CNN step.  A horizontal IoU is a dozen flops, so both passes recompute it instead:
//   pass 1 (row maxima): one thread per anchor walks the K ground truths (box + area staged in LDS, broadcast reads).  Only
//     pairs that overlap can raise a row maximum above its initial 0, and they are rare (a ground truth touches a few
//     thousand of the 611 072 anchors), so the division and the update sit behind `overlap > 0`: the update is an LDS
//     atomicMax of one 64-bit key (IoU bits : ~index -- IoUs are >= 0, so integer order = float order and the lowest index
//     wins ties), tried only when the IoU reaches the row's current LDS value; workgroups meet in a global atomicMax per
//     ground truth they touched.  A row no anchor overlaps keeps key 0 = (IoU 0, first anchor), what the ascending
//     reference argmax gives.  (First form: thread = ground truth x anchor slice over 2 048 staged anchors, one wave per
//     SIMD and a division per pair: 175 us at K = 100; this one: see profiles/r06_hba_ab.txt.)
//   pass 2 (columns): one thread per anchor walks the K ground truths (LDS broadcast), keeps max / first argmax, the LAST
//     row whose IoU equals its row maximum (the low-quality rule of the ascending reference loop), applies the thresholds.
// Same arithmetic as bbox_overlaps_kernel (fp32, no contraction; 0 / x = +0 for the skipped pairs), so gt_inds equal the
// matrix route's bit for bit (tests/test_gpu_orpn.py).  K <= 1024; finite boxes.
constexpr int HBA_NT = 256, HBA_MAXK = 1024;

__device__ __forceinline__ float hba_overlap(float gx1, float gy1, float gx2, float gy2, float cx1, float cy1, float cx2,
                                             float cy2) {
  const float w = fmaxf(fminf(gx2, cx2) - fmaxf(gx1, cx1), 0.f);
  const float h = fmaxf(fminf(gy2, cy2) - fmaxf(gy1, cy1), 0.f);
  return w * h;
}

__device__ __forceinline__ float hba_iou(float ov, float ga, float ca, float eps) {
  const float uni = (ga + ca) - ov;
  return ov / fmaxf(uni, eps);
}

// ground truths as one 32-byte record each (x1, y1, x2, y2, area, -, -, -): the walk over k is uniform across a wave, so
// the record arrives through the scalar cache (s_load_dwordx8) and costs no LDS or vector-memory issue slot -- staged in
// LDS, the two broadcast reads per pair were the bound (K = 400: 250 us per pass against a 37 us VALU floor)
__global__ __launch_bounds__(HBA_NT) void hba_prep_kernel(const float* __restrict__ gt, int K, int gstride,
                                                          float* __restrict__ tab, u64* __restrict__ rowkey) {
  const int k = blockIdx.x * HBA_NT + threadIdx.x;
  if (k >= ((K + 7) & ~7)) return;
  float* t = tab + (long long)k * 8;
  if (k >= K) {  // padding up to a multiple of eight: empty boxes at the origin overlap nothing
    t[0] = t[1] = t[2] = t[3] = t[4] = t[5] = t[6] = t[7] = 0.f;
    return;
  }
  const float* p = gt + (long long)k * gstride;
  const float x1 = p[0], y1 = p[1], x2 = p[2], y2 = p[3];
  t[0] = x1, t[1] = y1, t[2] = x2, t[3] = y2, t[4] = (x2 - x1) * (y2 - y1), t[5] = 0.f, t[6] = 0.f, t[7] = 0.f;
  rowkey[k] = 0ull;
}

// Both passes walk the ground truths eight at a time: the eight records are fetched together (one wait), the eight
// overlaps are computed branch-free, and only a lane that overlaps one of the eight goes on to divisions / updates -- per
// ground truth, one scalar-load latency and one divergent branch had left the loop latency-bound (K = 400: 245 us a pass).
constexpr int HBA_U = 8;

__device__ __forceinline__ bool hba_overlaps8(const float* __restrict__ tab, int k0, float cx1, float cy1, float cx2,
                                              float cy2, float (&ov)[HBA_U], float (&ga)[HBA_U]) {
  const float4* g = reinterpret_cast<const float4*>(tab + (long long)k0 * 8);   // (the table is padded to whole eights)
  float4 r[HBA_U];
#pragma unroll
  for (int u = 0; u < HBA_U; ++u) r[u] = g[2 * u], ga[u] = g[2 * u + 1].x;      // scalar loads in flight together, one wait
  bool any = false;
#pragma unroll
  for (int u = 0; u < HBA_U; ++u) {
    ov[u] = hba_overlap(r[u].x, r[u].y, r[u].z, r[u].w, cx1, cy1, cx2, cy2);
    any |= ov[u] > 0.f;
  }
  return any;
}

// maximum of a non-negative value over the wave, the same in every lane: rows of 16 on the vector pipe, the four rows
// through scalar registers (inactive lanes read as 0)
__device__ __forceinline__ float hba_wave_max(float v) {
#define HBA_DPP(ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true))
  v = fmaxf(v, HBA_DPP(0xB1));   // quad_perm [1, 0, 3, 2]
  v = fmaxf(v, HBA_DPP(0x4E));   // quad_perm [2, 3, 0, 1]
  v = fmaxf(v, HBA_DPP(0x141));  // row_half_mirror
  v = fmaxf(v, HBA_DPP(0x140));  // row_mirror
#undef HBA_DPP
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// Workgroups are taken from the END of the anchor list first: the list is level-major, its last few thousand anchors are
// the coarse levels' (256-512 px boxes that overlap a third of the ground truths and take the division for each), and
// started last they were the whole tail of both kernels (K = 400: 63 of 1 023 workgroups, 427 of 484 us).
__global__ __launch_bounds__(HBA_NT) void hba_rowmax_kernel(const float* __restrict__ tab, int K,
                                                            const float* __restrict__ anchors, int A, int astride, float eps,
                                                            u64* __restrict__ rowkey) {
  __shared__ u64 s_key[HBA_MAXK];
  for (int k = threadIdx.x; k < K; k += HBA_NT) s_key[k] = 0ull;
  __syncthreads();
  const int j = (int)(gridDim.x - 1 - blockIdx.x) * HBA_NT + threadIdx.x;
  const bool live = j < A;                                   // (every lane walks the loop: the wave reductions want them)
  const float* q = anchors + (long long)(live ? j : A - 1) * astride;
  const float cx1 = q[0], cy1 = q[1], cx2 = q[2], cy2 = q[3], ca = (cx2 - cx1) * (cy2 - cy1);
  const u64 low = (u64)(0xFFFFFFFFu - (unsigned)j);
  const int lane = threadIdx.x & 63;
  for (int k0 = 0; k0 < K; k0 += HBA_U) {
    float ov[HBA_U], ga[HBA_U];
    const bool any = hba_overlaps8(tab, k0, cx1, cy1, cx2, cy2, ov, ga) && live;
    if (!__builtin_amdgcn_ballot_w64(any)) continue;         // uniform
    int hits = 0;                                            // ground truths of the eight that some lane overlaps
#pragma unroll
    for (int u = 0; u < HBA_U; ++u) hits += __builtin_amdgcn_ballot_w64(live && ov[u] > 0.f) != 0;
    if (hits >= 3) {
      // a wave of coarse anchors meets most ground truths: all eight quotients (0 / x = +0 where nothing overlaps) and
      // all eight reductions side by side -- one after the other behind a branch each, their latencies added up on a wave
      // that has its SIMD to itself (63 such workgroups were 180 of the 185 us at K = 400)
      float v[HBA_U], m[HBA_U];
#pragma unroll
      for (int u = 0; u < HBA_U; ++u) v[u] = live ? hba_iou(ov[u], ga[u], ca, eps) : 0.f;
#pragma unroll
      for (int u = 0; u < HBA_U; ++u) m[u] = hba_wave_max(v[u]);
#pragma unroll
      for (int u = 0; u < HBA_U; ++u)
        if (m[u] > 0.f && (int)__builtin_ctzll(__builtin_amdgcn_ballot_w64(v[u] == m[u])) == lane)
          atomicMax(&s_key[k0 + u], ((u64)__float_as_uint(v[u]) << 32) | low);
      continue;
    }
#pragma unroll
    for (int u = 0; u < HBA_U; ++u) {
      const bool hit = live && ov[u] > 0.f;
      if (!__builtin_amdgcn_ballot_w64(hit)) continue;       // uniform
      const int k = k0 + u;
      float v = 0.f;
      if (hit) v = hba_iou(ov[u], ga[u], ca, eps);
      // one candidate per wave and ground truth: the largest IoU, at its lowest anchor (lanes are in anchor order) --
      // with a 64-bit LDS atomic per overlapping LANE, a wave of coarse anchors serialised 64-fold on every k
      const float m = hba_wave_max(v);
      // (m > 0: a quotient that underflows to 0 must not displace the "first anchor" answer of an all-zero row)
      if (m > 0.f && (int)__builtin_ctzll(__builtin_amdgcn_ballot_w64(v == m)) == lane)
        atomicMax(&s_key[k], ((u64)__float_as_uint(v) << 32) | low);   // (no return value: nothing waits for it)
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += HBA_NT) {
    const u64 key = s_key[k];
    if (key && key > __atomic_load_n(rowkey + k, __ATOMIC_RELAXED)) atomicMax(rowkey + k, key);
  }
}

__global__ __launch_bounds__(HBA_NT) void hba_col_kernel(const float* __restrict__ tab, int K,
                                                         const float* __restrict__ anchors, int A, int astride, float eps,
                                                         const u64* __restrict__ rowkey, float pos_thr, float neg_lo,
                                                         float neg_hi, float min_pos_iou, int match_low_quality,
                                                         int gt_max_assign_all, int* __restrict__ gt_inds,
                                                         float* __restrict__ max_ov) {
  const int j = (int)(gridDim.x - 1 - blockIdx.x) * HBA_NT + threadIdx.x;
  const bool live = j < A;
  const float* q = anchors + (long long)(live ? j : A - 1) * astride;
  const float cx1 = q[0], cy1 = q[1], cx2 = q[2], cy2 = q[3], ca = (cx2 - cx1) * (cy2 - cy1);
  // an anchor that overlaps none of the eight has IoU +0 with each: after k = 0 that raises no maximum, and it meets the
  // low-quality rule only through a row maximum of 0, which min_pos_iou > 0 rules out -- otherwise take the full path
  const bool zero_rows_matter = match_low_quality && !(min_pos_iou > 0.f);
  float best = 0.f;                                          // = IoU with ground truth 0 once k = 0 has been seen
  int arg = 0, lowq = -1;
  for (int k0 = 0; k0 < K; k0 += HBA_U) {
    float ov[HBA_U], ga[HBA_U];
    const bool any = hba_overlaps8(tab, k0, cx1, cy1, cx2, cy2, ov, ga);
    if (!zero_rows_matter && !__builtin_amdgcn_ballot_w64(any)) continue;      // uniform
    const ulonglong2* rk = reinterpret_cast<const ulonglong2*>(rowkey + k0);   // (allocation padded past whole eights)
    ulonglong2 keys[HBA_U / 2];
#pragma unroll
    for (int u = 0; u < HBA_U / 2; ++u) keys[u] = rk[u];                        // scalar loads, in flight together
    int hits = 0;
#pragma unroll
    for (int u = 0; u < HBA_U; ++u) hits += __builtin_amdgcn_ballot_w64(ov[u] > 0.f) != 0;
    const bool dense = hits >= 3 || zero_rows_matter;       // uniform; dense: the eight quotients side by side
    float vd[HBA_U];
    if (dense) {
#pragma unroll
      for (int u = 0; u < HBA_U; ++u) vd[u] = hba_iou(ov[u], ga[u], ca, eps);    // 0 / x = +0 where nothing overlaps
    }
#pragma unroll
    for (int u = 0; u < HBA_U; ++u) {
      const int k = k0 + u;
      if (k >= K) break;
      float v = 0.f;                                                   // 0 / x = +0: the same bits without the division
      if (dense) {
        v = vd[u];
      } else {
        if (k > 0 && !__builtin_amdgcn_ballot_w64(ov[u] > 0.f)) continue;      // uniform
        if (ov[u] > 0.f) v = hba_iou(ov[u], ga[u], ca, eps);
      }
      if (k == 0 || v > best) best = v, arg = k;
      if (match_low_quality) {
        const u64 key = (u & 1) ? keys[u / 2].y : keys[u / 2].x;
        const float rm = __uint_as_float((unsigned)(key >> 32));
        const int ra = key ? (int)(0xFFFFFFFFu - (unsigned)key) : 0;   // untouched row: IoU 0 everywhere, first anchor
        if (rm >= min_pos_iou)
          if (gt_max_assign_all ? (v == rm) : (ra == j)) lowq = k;
      }
    }
  }
  if (!live) return;
  int gi = -1;
  if (best >= neg_lo && best < neg_hi) gi = 0;   // assigner.py:138-145
  if (best >= pos_thr) gi = arg + 1;             // :147-148
  if (lowq >= 0) gi = lowq + 1;                  // :151-158
  gt_inds[j] = gi;
  if (max_ov) max_ov[j] = best;
}

}  // namespace rsdet

using namespace rsdet;

static inline size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }

template <class K, int NC, class Src, class Plan>
static void sel_run(const Src& src, const Plan& plan, int jobs, int n_max, SelWs<NC>* ws, u64* above, int cap,
                    unsigned* ties, hipStream_t s) {
  (void)hipMemsetAsync(ws, 0, sizeof(SelWs<NC>) * (size_t)jobs, s);
  const dim3 grid((unsigned)((n_max + SEL_CHUNK - 1) / SEL_CHUNK), (unsigned)jobs);
  for (int p = 0; p < SelKey<K>::P; ++p)
    hipLaunchKernelGGL((sel_hist_kernel<K, NC, Src, Plan>), grid, dim3(SEL_T), 0, s, src, plan, ws, p);
  hipLaunchKernelGGL((sel_emit_kernel<K, NC, Src, Plan>), grid, dim3(SEL_T), 0, s, src, plan, ws, above, cap, ties);
}

// ---- RandomSampler.sample_masked ------------------------------------------------------------------------------------
extern "C" size_t rsdet_sample_masked_ws_size(int num) {
  if (num <= 0) return 0;
  return up256(sizeof(SelWs<2>)) + up256((size_t)2 * num * sizeof(u64)) + up256((size_t)2 * SEL_TIE_CAP * sizeof(unsigned));
}

extern "C" int rsdet_sample_masked(const int32_t* gt_inds, const uint8_t* valid, int n_props, int k_gt, const void* pri,
                                   int pri_f64, int num, int num_pos, float neg_pos_ub, int64_t* inds, uint8_t* is_pos,
                                   uint8_t* val, int64_t* assigned, int64_t* counts, void* ws, size_t ws_bytes,
                                   void* stream) {
  if (n_props < 0 || k_gt < 0 || num <= 0 || num > 1024 || num_pos < 0) return RSDET_EINVAL;
  const long long n = (long long)n_props + k_gt;
  if (n <= 0 || n > 0x7FFFFFFFll) return RSDET_EINVAL;
  if ((n_props && !gt_inds) || !pri || !inds || !is_pos || !val || !assigned || !counts || !ws ||
      ws_bytes < rsdet_sample_masked_ws_size(num) || ((uintptr_t)ws & 15))
    return RSDET_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  SamplerSrc src{gt_inds, valid, pri, k_gt, (int)n, pri_f64};
  SamplerPlan plan{(int)(num_pos < n ? num_pos : n), (int)(num < n ? num : n), num, neg_pos_ub};
  char* w = (char*)ws;
  SelWs<2>* st = (SelWs<2>*)w;
  w += up256(sizeof(SelWs<2>));
  u64* above = (u64*)w;
  w += up256((size_t)2 * num * sizeof(u64));
  unsigned* ties = (unsigned*)w;
  if (pri_f64) {
    sel_run<u64, 2>(src, plan, 1, (int)n, st, above, num, ties, s);
    hipLaunchKernelGGL(sampler_final_kernel<u64>, dim3(1), dim3(1024), 0, s, src, st, above, ties, num, num,
                       (long long*)inds, is_pos, val, (long long*)assigned, (long long*)counts);
  } else {
    sel_run<unsigned, 2>(src, plan, 1, (int)n, st, above, num, ties, s);
    hipLaunchKernelGGL(sampler_final_kernel<unsigned>, dim3(1), dim3(1024), 0, s, src, st, above, ties, num, num,
                       (long long*)inds, is_pos, val, (long long*)assigned, (long long*)counts);
  }
  return rsdet_launch_status();
}

// ---- MidpointOffsetCoder.decode, obb2hbb ---------------------------------------------------------------------------
static F6 load6(const float* host, float dflt) {
  F6 f;
  for (int k = 0; k < 6; ++k) f.v[k] = host ? host[k] : dflt;
  return f;
}

extern "C" int rsdet_midpoint_offset_decode_f32(const float* anchors, const float* deltas, int n, const float* means,
                                                const float* stds, float max_ratio, float* out, void* stream) {
  if (n < 0) return RSDET_EINVAL;
  if (n == 0) return RSDET_OK;
  if (!anchors || !deltas || !out) return RSDET_EINVAL;
  hipLaunchKernelGGL(orpn_decode_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, anchors, deltas, n,
                     load6(means, 0.f), load6(stds, 1.f), max_ratio, out);
  return rsdet_launch_status();
}

extern "C" int rsdet_obb2hbb_f32(const float* obb, int n, float* out, void* stream) {
  if (n < 0) return RSDET_EINVAL;
  if (n == 0) return RSDET_OK;
  if (!obb || !out) return RSDET_EINVAL;
  hipLaunchKernelGGL(orpn_obb2hbb_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, obb, n, out);
  return rsdet_launch_status();
}

// ---- the proposals of a batch ---------------------------------------------------------------------------------------
static int orpn_n_tot(const rsdet_orpn_levels* d) {
  long long t = 0;
  for (int l = 0; l < d->n_levels; ++l) {
    const long long c = (long long)d->A * d->hw[l];
    t += c < d->nms_pre ? c : d->nms_pre;
  }
  return t > 0x7FFFFFFF ? -1 : (int)t;
}

extern "C" int rsdet_orpn_proposals_supported(const rsdet_orpn_levels* d) {
  if (!d || d->n_img <= 0 || d->n_levels <= 0 || d->n_levels > 7 || d->A <= 0 || d->nms_pre <= 0 ||
      d->nms_pre > SEL_TIE_CAP || d->nms_post <= 0)
    return 0;
  for (int l = 0; l < d->n_levels; ++l)
    if (d->hw[l] <= 0 || (long long)d->A * d->hw[l] > 0x1FFFFF) return 0;
  const int n = orpn_n_tot(d);
  return n > 0 && n <= ORPN_NP;
}

extern "C" int rsdet_orpn_proposals_n(const rsdet_orpn_levels* d) { return rsdet_orpn_proposals_supported(d) ? orpn_n_tot(d) : -1; }

struct OrpnWs {
  size_t st, above, ties, dets, boxes, okf, keep, nms, nms_each, total;
};
static OrpnWs orpn_ws(const rsdet_orpn_levels* d) {
  OrpnWs o;
  const size_t jobs = (size_t)d->n_img * d->n_levels, n = (size_t)orpn_n_tot(d), N = (size_t)d->n_img;
  size_t at = 0;
  o.st = at, at += up256(sizeof(SelWs<1>) * jobs);
  o.above = at, at += up256(jobs * d->nms_pre * sizeof(u64));
  o.ties = at, at += up256(jobs * SEL_TIE_CAP * sizeof(unsigned));
  o.dets = at, at += up256(N * n * 6 * sizeof(float));
  o.boxes = at, at += up256(N * n * 4 * sizeof(float));
  o.okf = at, at += up256(N * n);
  o.keep = at, at += up256(N * n);
  o.nms_each = up256(rsdet_nms_hbb_ws_size((int)n));
  o.nms = at, at += o.nms_each * N;
  o.total = at;
  return o;
}

extern "C" size_t rsdet_orpn_proposals_ws_size(const rsdet_orpn_levels* d) {
  return rsdet_orpn_proposals_supported(d) ? orpn_ws(d).total : 0;
}

extern "C" int rsdet_orpn_proposals_f32(const rsdet_orpn_levels* d, float* out, uint8_t* flags, void* ws, size_t ws_bytes,
                                        void* stream) {
  if (!rsdet_orpn_proposals_supported(d)) return RSDET_EINVAL;
  const OrpnWs o = orpn_ws(d);
  if (!out || !flags || !ws || ws_bytes < o.total || ((uintptr_t)ws & 15)) return RSDET_EINVAL;
  OrpnArgs a;
  int n_max = 0;
  for (int l = 0; l < 8; ++l) {
    const bool in = l < d->n_levels;
    if (in && (!d->score[l] || !d->reg[l] || !d->anchors[l])) return RSDET_EINVAL;
    a.src.score[l] = in ? d->score[l] : nullptr;
    a.src.cnt[l] = in ? d->A * d->hw[l] : 0;
    a.reg[l] = in ? d->reg[l] : nullptr;
    a.anchors[l] = in ? d->anchors[l] : nullptr;
    a.hw[l] = in ? d->hw[l] : 0;
    if (a.src.cnt[l] > n_max) n_max = a.src.cnt[l];
  }
  a.src.L = d->n_levels;
  a.A = d->A;
  a.nms_pre = d->nms_pre;
  a.n_tot = orpn_n_tot(d);
  a.mean = load6(d->means, 0.f);
  a.stdv = load6(d->stds, 1.f);
  a.max_ratio = d->max_ratio;
  a.min_size = d->min_size;
  hipStream_t s = (hipStream_t)stream;
  char* w = (char*)ws;
  SelWs<1>* st = (SelWs<1>*)(w + o.st);
  u64* above = (u64*)(w + o.above);
  unsigned* ties = (unsigned*)(w + o.ties);
  float *dets = (float*)(w + o.dets), *boxes = (float*)(w + o.boxes);
  unsigned char *okf = (unsigned char*)(w + o.okf), *keep = (unsigned char*)(w + o.keep);
  const int jobs = d->n_img * d->n_levels, n = a.n_tot;
  sel_run<unsigned, 1>(a.src, LevelPlan{d->nms_pre}, jobs, n_max, st, above, d->nms_pre, ties, s);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)orpn_sort_decode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            ORPN_NP * (int)sizeof(u64)) != hipSuccess)
      return RSDET_ELAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL(orpn_sort_decode_kernel, dim3(d->n_img), dim3(1024), ORPN_NP * sizeof(u64), s, a, st, above, ties,
                     dets, boxes, okf);
  for (int b = 0; b < d->n_img; ++b) {
    const int rc = rsdet_nms_hbb_sorted_f32(boxes + (size_t)b * n * 4, n, d->nms_thr, 1, keep + (size_t)b * n,
                                            w + o.nms + o.nms_each * b, o.nms_each, stream);
    if (rc != RSDET_OK) return rc;
  }
  hipLaunchKernelGGL(orpn_finish_kernel, dim3(d->n_img), dim3(1024), 0, s, keep, okf, dets, n, d->nms_post, out, flags);
  return rsdet_launch_status();
}

// ---- the Oriented RPN's losses on the samples -----------------------------------------------------------------------
static int orpn_loss_args(const rsdet_orpn_loss* d, OrpnLossArgs& a, bool forward) {
  if (!d || d->n_img <= 0 || d->n_img > 16 || d->n_levels <= 0 || d->n_levels > 8 || d->A <= 0 || d->num <= 0) return 0;
  for (int l = 0; l < 8; ++l) {
    const bool in = l < d->n_levels;
    if (in && (d->hw[l] <= 0 || !d->cls[l] || !d->reg[l])) return 0;
    if (in && (long long)d->n_img * d->A * 6 * d->hw[l] > 0x7FFFFFFFll) return 0;
    a.cls[l] = in ? d->cls[l] : nullptr;
    a.reg[l] = in ? d->reg[l] : nullptr;
    a.hw[l] = in ? d->hw[l] : 0;
  }
  a.n_img = d->n_img, a.L = d->n_levels, a.A = d->A, a.num = d->num;
  if (!forward) return 1;
  if (!d->anchors || !d->inds || !d->is_pos || !d->val || !d->assigned || !d->counts || !(d->beta > 0.f)) return 0;
  for (int b = 0; b < 16; ++b) {
    a.gt[b] = b < d->n_img ? d->gt[b] : nullptr;
    a.k_gt[b] = b < d->n_img ? d->k_gt[b] : 0;
  }
  a.anchors = d->anchors;
  a.inside = (const long long*)d->inside;
  a.inds = (const long long*)d->inds;
  a.is_pos = d->is_pos;
  a.val = d->val;
  a.assigned = (const long long*)d->assigned;
  a.counts = (const long long*)d->counts;
  a.mean = load6(d->means, 0.f);
  a.stdv = load6(d->stds, 1.f);
  a.beta = d->beta, a.w_cls = d->w_cls, a.w_box = d->w_box, a.pos_weight = d->pos_weight;
  return 1;
}

extern "C" int rsdet_orpn_loss_rec_floats(int n_img, int num) { return n_img > 0 && num > 0 ? n_img * num * ORPN_REC : 0; }

extern "C" int rsdet_orpn_loss_forward_f32(const rsdet_orpn_loss* d, float* losses, float* rec, void* stream) {
  OrpnLossArgs a;
  if (!orpn_loss_args(d, a, true) || !losses || !rec) return RSDET_EINVAL;
  hipLaunchKernelGGL(orpn_loss_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a, losses, rec);
  return rsdet_launch_status();
}

extern "C" int rsdet_orpn_loss_backward_f32(const rsdet_orpn_loss* d, const float* rec, const float* grad_losses,
                                            void* stream) {
  OrpnLossArgs a;
  if (!orpn_loss_args(d, a, false) || !rec || !grad_losses) return RSDET_EINVAL;
  const int S = d->n_img * d->num;
  hipLaunchKernelGGL(orpn_loss_bwd_kernel, dim3((S + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, rec, grad_losses);
  return rsdet_launch_status();
}

// ---- OrientedHead: sampled RoIs and targets of one image --------------------------------------------------------------
extern "C" int rsdet_orcnn_roi_targets_f32(const float* props, int prop_stride, int n_props, const float* gt,
                                           const int64_t* gt_labels, int k_gt, const int64_t* inds, const uint8_t* is_pos,
                                           const uint8_t* val, const int64_t* assigned, int num, int image, int num_classes,
                                           const float* means, const float* stds, float pos_weight, float* rois,
                                           int64_t* labels, float* label_weights, float* bbox_targets, float* bbox_weights,
                                           void* stream) {
  if (num <= 0 || prop_stride < 5 || n_props < 0 || k_gt < 0 || (n_props && !props) || (k_gt && (!gt || !gt_labels)) ||
      !inds || !is_pos || !val || !assigned || !rois || !labels || !label_weights || !bbox_targets || !bbox_weights)
    return RSDET_EINVAL;
  RoiTargetArgs a;
  a.props = props, a.gt = gt, a.gt_labels = (const long long*)gt_labels, a.inds = (const long long*)inds;
  a.is_pos = is_pos, a.val = val, a.assigned = (const long long*)assigned;
  a.stride = prop_stride, a.k_gt = k_gt, a.num = num, a.image = image, a.num_classes = num_classes;
  for (int k = 0; k < 5; ++k) {
    a.mean.v[k] = means ? means[k] : 0.f;
    a.stdv.v[k] = stds ? stds[k] : 1.f;
  }
  a.pos_weight = pos_weight;
  a.rois = rois, a.labels = (long long*)labels, a.label_weights = label_weights, a.bbox_targets = bbox_targets;
  a.bbox_weights = bbox_weights;
  hipLaunchKernelGGL(roi_targets_kernel, dim3((num + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
  return rsdet_launch_status();
}

// ---- MaxIoUAssigner on horizontal boxes, no matrix ---------------------------------------------------------------------
extern "C" size_t rsdet_hbb_assign_ws_size(int K) {
  return K > 0 && K <= HBA_MAXK ? up256((size_t)K * sizeof(u64)) + up256((size_t)((K + 7) & ~7) * 32) : 0;
}

extern "C" int rsdet_hbb_assign_f32(const float* gt, int K, int gt_stride, const float* anchors, int A, int anchor_stride,
                                    float eps, float pos_iou_thr, float neg_lo, float neg_hi, float min_pos_iou,
                                    int match_low_quality, int gt_max_assign_all, int32_t* gt_inds, float* max_overlaps,
                                    void* ws, size_t ws_bytes, void* stream) {
  if (K < 1 || K > HBA_MAXK || A < 1 || gt_stride < 4 || anchor_stride < 4 || !gt || !anchors || !gt_inds || !ws ||
      ws_bytes < rsdet_hbb_assign_ws_size(K) || ((uintptr_t)ws & 31))
    return RSDET_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  u64* rowkey = (u64*)ws;
  float* tab = (float*)((char*)ws + up256((size_t)K * sizeof(u64)));
  hipLaunchKernelGGL(hba_prep_kernel, dim3((K + 7 + HBA_NT - 1) / HBA_NT), dim3(HBA_NT), 0, s, gt, K, gt_stride, tab, rowkey);
  hipLaunchKernelGGL(hba_rowmax_kernel, dim3((A + HBA_NT - 1) / HBA_NT), dim3(HBA_NT), 0, s, tab, K, anchors, A,
                     anchor_stride, eps, rowkey);
  hipLaunchKernelGGL(hba_col_kernel, dim3((A + HBA_NT - 1) / HBA_NT), dim3(HBA_NT), 0, s, tab, K, anchors, A, anchor_stride,
                     eps, rowkey, pos_iou_thr, neg_lo, neg_hi, min_pos_iou, match_low_quality, gt_max_assign_all, gt_inds,
                     max_overlaps);
  return rsdet_launch_status();
}
