// K4: Hungarian-matcher cost matrices, and the point sampler shared with the loss.
// Replaces Mask2FormerHungarianMatcher.forward up to the scipy solver -- transformers
// modeling_mask2former.py:444-472 -- with sample_point (:245-274), the pair-wise sigmoid CE
// (:350-374) and the pair-wise dice (:328-347).
//
// The reference loops over images and prediction levels, launching ~10 small ops and forcing one
// device->host copy per (image, level): B x 10 syncs per step.  Here every level of one image is
// covered by two launches (target sampling, cost), nothing syncs, and the caller copies ALL cost
// matrices to the host once.
//
// Numerics: sampled values, softplus and sigmoid are fp32 (as in the reference); each thread sums
// its ~49 of the P points in fp32, the cross-thread reduction and the cost arithmetic are fp64,
// so the cost is within fp32 round-off of the exact value -- assignments are stable under that
// (SURVEY.md F8).
#include "common.h"

namespace wm2f {

// grid_sample(bilinear, zeros, align_corners=False) of one (H, W) map at normalised (x, y).
template <typename T>
__device__ __forceinline__ float bilinear_zeros(const T* __restrict__ img, int H, int W, float lx, float ly) {
  const float gx = 2.f * lx - 1.f, gy = 2.f * ly - 1.f;
  const float x = ((gx + 1.f) * (float)W - 1.f) * 0.5f;
  const float y = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
  if (!(x > -1.f && x < (float)W && y > -1.f && y < (float)H)) return 0.f;
  const float x0f = floorf(x), y0f = floorf(y);
  const int x0 = (int)x0f, y0 = (int)y0f;
  const float fx1 = x - x0f, fy1 = y - y0f, fx0 = 1.f - fx1, fy0 = 1.f - fy1;
  const bool xl = x0 >= 0, xr = x0 + 1 < W, yt = y0 >= 0, yb = y0 + 1 < H;
  const T* p = img + (int64_t)y0 * W + x0;
  float r = 0.f;
  if (yt && xl) r += (float)p[0] * (fx0 * fy0);
  if (yt && xr) r += (float)p[1] * (fx1 * fy0);
  if (yb && xl) r += (float)p[W] * (fx0 * fy1);
  if (yb && xr) r += (float)p[W + 1] * (fx1 * fy1);
  return r;
}

// out[m][p] = bilinear(feat[map_index ? map_index[m] : m], pts[m][p])          grid (ceil(P/256), M)
template <typename T>
__global__ __launch_bounds__(256) void point_sample_fwd_kernel(const T* __restrict__ feat,
                                                               const float* __restrict__ pts,
                                                               const int32_t* __restrict__ map_index,
                                                               float* __restrict__ out, int H, int W, int P) {
  const int p = blockIdx.x * 256 + threadIdx.x, m = blockIdx.y;
  if (p >= P) return;
  const int64_t src = map_index ? map_index[m] : m;
  const float* pp = pts + ((int64_t)m * P + p) * 2;
  out[(int64_t)m * P + p] = bilinear_zeros(feat + src * H * W, H, W, pp[0], pp[1]);
}

__global__ __launch_bounds__(256) void point_sample_bwd_kernel(const float* __restrict__ grad_out,
                                                               const float* __restrict__ pts,
                                                               const int32_t* __restrict__ map_index,
                                                               float* __restrict__ grad_feat, int H, int W, int P) {
  const int p = blockIdx.x * 256 + threadIdx.x, m = blockIdx.y;
  if (p >= P) return;
  const int64_t n = map_index ? map_index[m] : m;
  const float* pp = pts + ((int64_t)m * P + p) * 2;
  const float go = grad_out[(int64_t)m * P + p];
  const float gx = 2.f * pp[0] - 1.f, gy = 2.f * pp[1] - 1.f;
  const float x = ((gx + 1.f) * (float)W - 1.f) * 0.5f;
  const float y = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
  if (!(x > -1.f && x < (float)W && y > -1.f && y < (float)H)) return;
  const float x0f = floorf(x), y0f = floorf(y);
  const int x0 = (int)x0f, y0 = (int)y0f;
  const float fx1 = x - x0f, fy1 = y - y0f, fx0 = 1.f - fx1, fy0 = 1.f - fy1;
  const bool xl = x0 >= 0, xr = x0 + 1 < W, yt = y0 >= 0, yb = y0 + 1 < H;
  float* g = grad_feat + n * H * W + (int64_t)y0 * W + x0;
  if (yt && xl) atomicAdd(g, go * fx0 * fy0);
  if (yt && xr) atomicAdd(g + 1, go * fx1 * fy0);
  if (yb && xl) atomicAdd(g + W, go * fx0 * fy1);
  if (yb && xr) atomicAdd(g + W + 1, go * fx1 * fy1);
}

// Targets of ONE image, all levels: tm[lvl][t][p].          grid (ceil(P/256), T_b, NL)
template <typename T>
__global__ __launch_bounds__(256) void matcher_sample_targets_kernel(const T* __restrict__ tgt /* (T_b,Ht,Wt) */,
                                                                     const float* __restrict__ points /* +b */,
                                                                     float* __restrict__ tm /* + tgt_offset*P */,
                                                                     int Ht, int Wt, int P, int64_t pts_stride_lvl,
                                                                     int64_t tm_stride_lvl) {
  const int p = blockIdx.x * 256 + threadIdx.x, t = blockIdx.y, lvl = blockIdx.z;
  if (p >= P) return;
  const float* pp = points + lvl * pts_stride_lvl + (int64_t)p * 2;
  tm[lvl * tm_stride_lvl + (int64_t)t * P + p] = bilinear_zeros(tgt + (int64_t)t * Ht * Wt, Ht, Wt, pp[0], pp[1]);
}

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWave);
  return v;
}

// Cost rows of EVERY image and level in one launch: block = (group of QG queries, level, image);
// loops over target chunks of TC.  Per-thread partial sums (P/256 ~ 49 terms) are fp32, the
// cross-thread reduction and the cost arithmetic are fp64.
constexpr int kQG = 2, kTC = 16, kMaxImg = 64;

struct MatcherImages {
  int off[kMaxImg + 1];  // target offsets per image
};
constexpr int kMaxMatchLevels = 16;
struct MatcherLevels {  // base pointer of each prediction level's (B, Q, h, w) logits: the levels are NOT stacked
  const float* p[kMaxMatchLevels];
};

__global__ __launch_bounds__(256) void matcher_cost_kernel(
    MatcherLevels levels, const float* __restrict__ class_logits, const float* __restrict__ tm,
    const int64_t* __restrict__ tgt_classes, const float* __restrict__ points, float* __restrict__ cost,
    MatcherImages im, int B, int Q, int C1, int h, int w, int P, int Tsum, int Tmax, float w_class, float w_mask,
    float w_dice) {
  constexpr int NV = kQG * kTC * 3 + kTC + kQG * 2;  // values reduced per chunk
  __shared__ float red[4][NV];
  const int b = blockIdx.z, lvl = blockIdx.y, q0 = blockIdx.x * kQG;
  const int t_begin = im.off[b], T = im.off[b + 1] - t_begin;
  if (T <= 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* ml = levels.p[lvl] + (int64_t)b * Q * h * w;
  const float* pts = points + ((int64_t)lvl * B + b) * P * 2;
  const float* tml = tm + ((int64_t)lvl * Tsum + t_begin) * P;
  const float* clb = class_logits + ((int64_t)lvl * B + b) * Q * C1;
  float* costb = cost + ((int64_t)lvl * B + b) * Q * Tmax;

  for (int t0 = 0; t0 < T; t0 += kTC) {
    float a_pos[kQG][kTC], a_neg[kQG][kTC], a_sig[kQG][kTC], s_t[kTC], s_neg[kQG], s_sig[kQG];
#pragma unroll
    for (int c = 0; c < kTC; ++c) {
      s_t[c] = 0.f;
#pragma unroll
      for (int qq = 0; qq < kQG; ++qq) a_pos[qq][c] = a_neg[qq][c] = a_sig[qq][c] = 0.f;
    }
#pragma unroll
    for (int qq = 0; qq < kQG; ++qq) s_neg[qq] = s_sig[qq] = 0.f;

    for (int p = threadIdx.x; p < P; p += 256) {
      const float2 pt = *reinterpret_cast<const float2*>(pts + 2 * p);
      float tv[kTC];
#pragma unroll
      for (int c = 0; c < kTC; ++c) {
        tv[c] = (t0 + c < T) ? tml[(int64_t)(t0 + c) * P + p] : 0.f;
        s_t[c] += tv[c];
      }
#pragma unroll
      for (int qq = 0; qq < kQG; ++qq) {
        int qi = q0 + qq;
        if (qi > Q - 1) qi = Q - 1;
        const float x = bilinear_zeros(ml + (int64_t)qi * h * w, h, w, pt.x, pt.y);
        const float e = expf(-fabsf(x));
        const float lg = log1pf(e);
        const float pos = fmaxf(-x, 0.f) + lg;  // BCEWithLogits(x, 1)
        const float neg = fmaxf(x, 0.f) + lg;   // BCEWithLogits(x, 0)
        const float r = 1.f / (1.f + e);
        const float sig = x >= 0.f ? r : e * r;
        s_neg[qq] += neg;
        s_sig[qq] += sig;
#pragma unroll
        for (int c = 0; c < kTC; ++c) {
          a_pos[qq][c] = fmaf(pos, tv[c], a_pos[qq][c]);
          a_neg[qq][c] = fmaf(neg, tv[c], a_neg[qq][c]);
          a_sig[qq][c] = fmaf(sig, tv[c], a_sig[qq][c]);
        }
      }
    }
    // ---- block reduction: wave shuffles, then 4 partials through LDS
    int vi = 0;
    auto put = [&](float v) {
      v = wave_sum_f(v);
      if (lane == 0) red[wave][vi] = v;
      ++vi;
    };
#pragma unroll
    for (int qq = 0; qq < kQG; ++qq)
#pragma unroll
      for (int c = 0; c < kTC; ++c) {
        put(a_pos[qq][c]);
        put(a_neg[qq][c]);
        put(a_sig[qq][c]);
      }
#pragma unroll
    for (int c = 0; c < kTC; ++c) put(s_t[c]);
#pragma unroll
    for (int qq = 0; qq < kQG; ++qq) {
      put(s_neg[qq]);
      put(s_sig[qq]);
    }
    __syncthreads();
    if (threadIdx.x < kQG * kTC) {
      const int qq = threadIdx.x / kTC, c = threadIdx.x % kTC;
      const int qi = q0 + qq, t = t0 + c;
      if (qi < Q && t < T) {
        auto R = [&](int i) { return (double)red[0][i] + (double)red[1][i] + (double)red[2][i] + (double)red[3][i]; };
        const double apos = R((qq * kTC + c) * 3 + 0), aneg = R((qq * kTC + c) * 3 + 1);
        const double asig = R((qq * kTC + c) * 3 + 2);
        const double st = R(kQG * kTC * 3 + c);
        const double sneg = R(kQG * kTC * 3 + kTC + qq * 2), ssig = R(kQG * kTC * 3 + kTC + qq * 2 + 1);
        const double cost_mask = (apos + (sneg - aneg)) / (double)P;
        const double cost_dice = 1.0 - (2.0 * asig + 1.0) / (ssig + st + 1.0);
        // -softmax(class_logits)[target class], HF:445-449
        const float* cl = clb + (int64_t)qi * C1;
        double mx = -1e300;
        for (int k2 = 0; k2 < C1; ++k2) mx = fmax(mx, (double)cl[k2]);
        double den = 0.0;
        for (int k2 = 0; k2 < C1; ++k2) den += exp((double)cl[k2] - mx);
        const int64_t tc = tgt_classes[t_begin + t];
        const double prob = (tc >= 0 && tc < C1) ? exp((double)cl[tc] - mx) / den : 0.0;
        const double cst = (double)w_mask * cost_mask - (double)w_class * prob + (double)w_dice * cost_dice;
        float cf = (float)cst;
        cf = fminf(cf, 1e10f);
        cf = fmaxf(cf, -1e10f);
        if (cf != cf) cf = 0.f;
        costb[(int64_t)qi * Tmax + t] = cf;
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------- band form
// The kernel above gathers every (query, point) sample from global memory: 4 scattered 4-byte loads per sample, 200 M samples per
// config-2 step -- the texture path's line rate, not the arithmetic, was its time (3.5 ms, plus a 1.4 ms sort of the points by pixel
// that made the gathers tolerable).  Here the maps come to the points instead: the P points of an (image, level) are GROUPED by the
// band of map rows their footprint starts in (a stable counting sort inside one workgroup: same order every run, so the sums
// are reproducible), and the cost workgroup of a query pair walks the bands -- stage rows [k R, k R + R] of its two maps in LDS
// with coalesced 16-byte loads (every map byte read once: 4.2 GB per step, the floor of any form since 12 544 random points touch
// every 64-byte line of a 256 x 256 map), then sample that band's points from LDS.  Two workgroups per CU (<= 72 KiB each), one
// staging while the other computes.  The sample arithmetic is bilinear_zeros' own, term by term.
constexpr int kMaxBands = 64;
__device__ const float4 g_zero_page_match[1] = {{0.f, 0.f, 0.f, 0.f}};  // LDS-DMA source for rows below the map
typedef const __attribute__((address_space(1))) void* mgptr_t;
typedef __attribute__((address_space(3))) void* mlptr_t;
constexpr int kBandLdsBytes = 72 * 1024;

__device__ __forceinline__ int point_band(float ly, int h, int R) {  // band of the row the footprint starts in (row -1 -> band 0)
  const float gy = 2.f * ly - 1.f;
  const float y = ((gy + 1.f) * (float)h - 1.f) * 0.5f;
  int y0 = (int)floorf(y);
  y0 = y0 < 0 ? 0 : (y0 > h - 1 ? h - 1 : y0);
  return y0 / R;
}

// column bucket of a point, 0 .. XB - 1: the secondary key of the grouping (points of a band that are neighbours in memory are then
// neighbours in the maps: the target sampler's 64 lanes read a dozen cache lines, not 64)
__device__ __forceinline__ int point_xbucket(float lx, int w, int XB) {
  const float gx = 2.f * lx - 1.f;
  const float x = ((gx + 1.f) * (float)w - 1.f) * 0.5f;
  int x0 = (int)floorf(x);
  x0 = x0 < 0 ? 0 : (x0 > w - 1 ? w - 1 : x0);
  return x0 * XB / w;
}

// grid = NL * B segments, 1024 threads.  out: the segment's points, stably grouped by (band, column bucket); band_off[seg][0 .. NB]
__global__ __launch_bounds__(1024) void matcher_group_points_kernel(const float* __restrict__ pts_in, float* __restrict__ pts_out,
                                                                    int* __restrict__ band_off, int P, int h, int w, int R, int NB,
                                                                    int XB) {
  __shared__ int hist[kMaxBands], base[kMaxBands], wcount[16][kMaxBands];
  const int n_bins = NB * XB;  // <= kMaxBands
  const int seg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float2* in = reinterpret_cast<const float2*>(pts_in) + (int64_t)seg * P;
  float2* out = reinterpret_cast<float2*>(pts_out) + (int64_t)seg * P;
  if (tid < kMaxBands) hist[tid] = 0;
  __syncthreads();
  for (int p = tid; p < P; p += 1024) {  // integer counts: order-independent
    const float2 pt = in[p];
    atomicAdd(&hist[point_band(pt.y, h, R) * XB + point_xbucket(pt.x, w, XB)], 1);
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int b = 0; b < n_bins; ++b) {
      base[b] = run;
      if (b % XB == 0) band_off[seg * (kMaxBands + 1) + b / XB] = run;
      run += hist[b];
    }
    band_off[seg * (kMaxBands + 1) + NB] = run;
  }
  __syncthreads();
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  for (int c0 = 0; c0 < P; c0 += 1024) {
    const int p = c0 + tid;
    float2 pt = make_float2(0.f, 0.f);
    int band = -1;
    if (p < P) {
      pt = in[p];
      band = point_band(pt.y, h, R) * XB + point_xbucket(pt.x, w, XB);
    }
    int rank = 0;
    for (int b = 0; b < n_bins; ++b) {  // (uniform loop) position of this point among the wave's points of its band, in index order
      const unsigned long long m = __builtin_amdgcn_ballot_w64(band == b);
      if (lane == 0) wcount[wave][b] = __builtin_popcountll(m);
      if (band == b) rank = __builtin_popcountll(m & lt);
    }
    __syncthreads();
    if (band >= 0) {
      int pos = base[band] + rank;
      for (int w2 = 0; w2 < wave; ++w2) pos += wcount[w2][band];
      out[pos] = pt;
    }
    __syncthreads();
    if (tid < n_bins) {
      int add = 0;
#pragma unroll
      for (int w2 = 0; w2 < 16; ++w2) add += wcount[w2][tid];
      base[tid] += add;
    }
    __syncthreads();
  }
}

template <bool VEC4, int QG>
__global__ __launch_bounds__(256) void matcher_cost_band_kernel(
    MatcherLevels levels, const float* __restrict__ class_logits, const float* __restrict__ tm,
    const int64_t* __restrict__ tgt_classes, const float* __restrict__ points, const int* __restrict__ band_off,
    float* __restrict__ cost, MatcherImages im, int B, int Q, int C1, int h, int w, int P, int Tsum, int Tmax, int R, int NB,
    float w_class, float w_mask, float w_dice) {
  constexpr int NV = QG * kTC * 3 + kTC + QG * 2;  // values reduced per chunk
  __shared__ float red[4][NV];
  extern __shared__ __attribute__((aligned(16))) float band[];  // [QG][(R + 1) * w rounded up to 256]
  const int b = blockIdx.z, lvl = blockIdx.y, q0 = blockIdx.x * QG;
  const int t_begin = im.off[b], T = im.off[b + 1] - t_begin;
  if (T <= 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* ml = levels.p[lvl] + (int64_t)b * Q * h * w;
  const int seg = lvl * B + b;
  const float* pts = points + (int64_t)seg * P * 2;
  const int* boff = band_off + seg * (kMaxBands + 1);
  const float* tml = tm + ((int64_t)lvl * Tsum + t_begin) * P;
  const float* clb = class_logits + ((int64_t)lvl * B + b) * Q * C1;
  float* costb = cost + ((int64_t)lvl * B + b) * Q * Tmax;
  const int band_elems = (R + 1) * w, band_stride = (band_elems + 255) & ~255;  // whole 1-KiB pieces per map

  for (int t0 = 0; t0 < T; t0 += kTC) {
    float a_pos[QG][kTC], a_neg[QG][kTC], a_sig[QG][kTC], s_t[kTC], s_neg[QG], s_sig[QG];
#pragma unroll
    for (int c = 0; c < kTC; ++c) {
      s_t[c] = 0.f;
#pragma unroll
      for (int qq = 0; qq < QG; ++qq) a_pos[qq][c] = a_neg[qq][c] = a_sig[qq][c] = 0.f;
    }
#pragma unroll
    for (int qq = 0; qq < QG; ++qq) s_neg[qq] = s_sig[qq] = 0.f;

    for (int k = 0; k < NB; ++k) {
      const int p_lo = boff[k], p_hi = boff[k + 1];
      if (p_lo == p_hi) continue;  // (uniform)
      __syncthreads();             // the previous band's samples are taken
      const int row0 = k * R;
#pragma unroll
      for (int qq = 0; qq < QG; ++qq) {
        int qi = q0 + qq;
        if (qi > Q - 1) qi = Q - 1;
        const float* src = ml + (int64_t)qi * h * w + (int64_t)row0 * w;
        const int n_valid = (h - row0 < R + 1 ? h - row0 : R + 1) * w;  // rows below the map: zeros
        float* dst = band + qq * band_stride;
        if (VEC4) {
          // w % 4 == 0: map rows and the band are 16-byte aligned -> LDS-DMA, a wave-instruction moves 1 KiB (64 lanes x 16 B,
          // lane-linear in LDS); all of a band's requests are in flight at once and no register holds them (a load / ds_write
          // loop kept ONE 16-byte load in flight per lane: 18 trips to HBM per band, 4.5 ms per step).  Rows below the map come
          // from a page of zeros.
          for (int c = wave; c * 256 < band_elems; c += 4) {
            const int i = c * 256 + lane * 4;
            const float* sp = i < n_valid ? src + i : reinterpret_cast<const float*>(g_zero_page_match);
            __builtin_amdgcn_global_load_lds((mgptr_t)sp, (mlptr_t)(dst + c * 256), 16, 0, 0);
          }
        } else {
          for (int i = threadIdx.x; i < band_elems; i += 256) dst[i] = i < n_valid ? src[i] : 0.f;
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      for (int p = p_lo + (int)threadIdx.x; p < p_hi; p += 256) {
        const float2 pt = *reinterpret_cast<const float2*>(pts + 2 * p);
        float tv[kTC];
#pragma unroll
        for (int c = 0; c < kTC; ++c) {
          tv[c] = (t0 + c < T) ? tml[(int64_t)(t0 + c) * P + p] : 0.f;
          s_t[c] += tv[c];
        }
        // bilinear_zeros' arithmetic on the band: same terms, same order
        const float gx = 2.f * pt.x - 1.f, gy = 2.f * pt.y - 1.f;
        const float x = ((gx + 1.f) * (float)w - 1.f) * 0.5f;
        const float y = ((gy + 1.f) * (float)h - 1.f) * 0.5f;
        const bool in = x > -1.f && x < (float)w && y > -1.f && y < (float)h;
        const float x0f = floorf(x), y0f = floorf(y);
        const int x0 = (int)x0f, y0 = (int)y0f;
        const float fx1 = x - x0f, fy1 = y - y0f, fx0 = 1.f - fx1, fy0 = 1.f - fy1;
        const bool xl = in && x0 >= 0, xr = in && x0 + 1 < w, yt = y0 >= 0, yb = y0 + 1 < h;
        const int o = in ? (y0 - row0) * w + x0 : 0;
#pragma unroll
        for (int qq = 0; qq < QG; ++qq) {
          const float* pb = band + qq * band_stride + o;
          float xs = 0.f;
          if (yt && xl) xs += pb[0] * (fx0 * fy0);
          if (yt && xr) xs += pb[1] * (fx1 * fy0);
          if (yb && xl) xs += pb[w] * (fx0 * fy1);
          if (yb && xr) xs += pb[w + 1] * (fx1 * fy1);
          const float e = expf(-fabsf(xs));
          const float lg = log1pf(e);
          const float pos = fmaxf(-xs, 0.f) + lg;  // BCEWithLogits(x, 1)
          const float neg = fmaxf(xs, 0.f) + lg;   // BCEWithLogits(x, 0)
          const float r = 1.f / (1.f + e);
          const float sig = xs >= 0.f ? r : e * r;
          s_neg[qq] += neg;
          s_sig[qq] += sig;
#pragma unroll
          for (int c = 0; c < kTC; ++c) {
            a_pos[qq][c] = fmaf(pos, tv[c], a_pos[qq][c]);
            a_neg[qq][c] = fmaf(neg, tv[c], a_neg[qq][c]);
            a_sig[qq][c] = fmaf(sig, tv[c], a_sig[qq][c]);
          }
        }
      }
    }
    // ---- block reduction and cost arithmetic: as matcher_cost_kernel
    int vi = 0;
    auto put = [&](float v) {
      v = wave_sum_f(v);
      if (lane == 0) red[wave][vi] = v;
      ++vi;
    };
#pragma unroll
    for (int qq = 0; qq < QG; ++qq)
#pragma unroll
      for (int c = 0; c < kTC; ++c) {
        put(a_pos[qq][c]);
        put(a_neg[qq][c]);
        put(a_sig[qq][c]);
      }
#pragma unroll
    for (int c = 0; c < kTC; ++c) put(s_t[c]);
#pragma unroll
    for (int qq = 0; qq < QG; ++qq) {
      put(s_neg[qq]);
      put(s_sig[qq]);
    }
    __syncthreads();
    if (threadIdx.x < QG * kTC) {
      const int qq = threadIdx.x / kTC, c = threadIdx.x % kTC;
      const int qi = q0 + qq, t = t0 + c;
      if (qi < Q && t < T) {
        auto Rd = [&](int i) { return (double)red[0][i] + (double)red[1][i] + (double)red[2][i] + (double)red[3][i]; };
        const double apos = Rd((qq * kTC + c) * 3 + 0), aneg = Rd((qq * kTC + c) * 3 + 1);
        const double asig = Rd((qq * kTC + c) * 3 + 2);
        const double st = Rd(QG * kTC * 3 + c);
        const double sneg = Rd(QG * kTC * 3 + kTC + qq * 2), ssig = Rd(QG * kTC * 3 + kTC + qq * 2 + 1);
        const double cost_mask = (apos + (sneg - aneg)) / (double)P;
        const double cost_dice = 1.0 - (2.0 * asig + 1.0) / (ssig + st + 1.0);
        const float* cl = clb + (int64_t)qi * C1;
        double mx = -1e300;
        for (int k2 = 0; k2 < C1; ++k2) mx = fmax(mx, (double)cl[k2]);
        double den = 0.0;
        for (int k2 = 0; k2 < C1; ++k2) den += exp((double)cl[k2] - mx);
        const int64_t tc = tgt_classes[t_begin + t];
        const double prob = (tc >= 0 && tc < C1) ? exp((double)cl[tc] - mx) / den : 0.0;
        const double cst = (double)w_mask * cost_mask - (double)w_class * prob + (double)w_dice * cost_dice;
        float cf = (float)cst;
        cf = fminf(cf, 1e10f);
        cf = fmaxf(cf, -1e10f);
        if (cf != cf) cf = 0.f;
        costb[(int64_t)qi * Tmax + t] = cf;
      }
    }
    __syncthreads();
  }
}

// rows per band for a map of width w (0: the band form does not apply)
constexpr int kBandQG = 2;  // queries per workgroup in the band form.  Every query group re-reads the (level, image)'s sampled targets
                            // from L2 (803 KB at config 2): 4 per workgroup halve that traffic (6.4 -> 3.2 GB per step next to the
                            // 4.2 GB of maps) but measured 3.98 against 2.34 ms -- 192 accumulators per lane and half the bands' rows
static inline int band_rows(int h, int w, int P) {
  if (P < 1024) return 0;  // few points: the gathers are not the time
  int R = kBandLdsBytes / (kBandQG * w * 4) - 1;
  if (R > h) R = h;
  if (R < 8) return 0;
  return (h + R - 1) / R <= kMaxBands ? R : 0;
}

}  // namespace wm2f

using namespace wm2f;

// [sampled targets NL * Tsum * P floats][grouped points NL * B * P * 2 floats][band offsets NL * B * 65 ints]
extern "C" int64_t wm2f_matcher_workspace(int NL, int B, int Q, int P, int Tsum) {
  (void)Q;
  if (NL <= 0 || P <= 0 || Tsum <= 0 || B <= 0) return 0;
  return (int64_t)NL * Tsum * P * 4 + (int64_t)NL * B * P * 8 + (int64_t)NL * B * (kMaxBands + 1) * 4 + 64;
}

static int matcher_cost_impl(const MatcherLevels& levels, const void* class_logits, const void* tgt_masks,
                             int tgt_dtype, const int32_t* tgt_offset, const void* tgt_classes, const void* points,
                             void* cost, void* workspace, int NL, int B, int Q, int C1, int h, int w, int Ht, int Wt,
                             int P, int Tmax, float w_class, float w_mask, float w_dice, void* stream, const char* who) {
  WM2F_REQUIRE(class_logits && tgt_offset && points && cost, "%s: null pointer", who);
  WM2F_REQUIRE(NL > 0 && B > 0 && Q > 0 && C1 > 0 && h > 0 && w > 0 && Ht > 0 && Wt > 0 && P > 0,
               "%s: non-positive size", who);
  WM2F_REQUIRE(tgt_dtype == 0 || tgt_dtype == 1, "%s: tgt_dtype must be 0 (fp32) or 1 (uint8)", who);
  WM2F_REQUIRE(NL <= kMaxMatchLevels, "%s: at most %d levels per call", who, kMaxMatchLevels);
  const int Tsum = tgt_offset[B];
  WM2F_REQUIRE(tgt_offset[0] == 0 && Tsum >= 0, "%s: bad tgt_offset", who);
  if (Tsum == 0) return WM2F_OK;
  WM2F_REQUIRE(tgt_masks && tgt_classes && workspace, "%s: null pointer", who);
  hipStream_t st = (hipStream_t)stream;
  float* tm = (float*)workspace;
  const int64_t tm_stride_lvl = (int64_t)Tsum * P;
  const int64_t pts_stride_lvl = (int64_t)B * P * 2;
  // band form: group every (level, image)'s points by map band first; targets and costs are then sampled at the grouped points
  // (the cost is a sum over the points: any order gives the same matrix up to fp32 summation order)
  const int R = band_rows(h, w, P), NB = R ? (h + R - 1) / R : 0;
  int* band_off = nullptr;
  if (R) {
    float* grouped = tm + (int64_t)NL * Tsum * P;
    band_off = (int*)(((uintptr_t)(grouped + (int64_t)NL * B * P * 2) + 15) & ~(uintptr_t)15);
    hipLaunchKernelGGL(matcher_group_points_kernel, dim3(NL * B), dim3(1024), 0, st, (const float*)points, grouped, band_off, P, h, w, R, NB,
                       kMaxBands / NB);
    points = grouped;
  }
  WM2F_REQUIRE(B <= kMaxImg, "%s: at most %d images per call", who, kMaxImg);
  MatcherImages im;
  for (int b = 0; b <= B; ++b) im.off[b] = tgt_offset[b];
  for (int b = 0; b < B; ++b) {
    const int T = tgt_offset[b + 1] - tgt_offset[b];
    WM2F_REQUIRE(T >= 0 && T <= Tmax && T <= 65535, "%s: image %d has %d targets (Tmax=%d)", who, b, T, Tmax);
    if (T == 0) continue;
    const float* pts_b = (const float*)points + (int64_t)b * P * 2;
    float* tm_b = tm + (int64_t)tgt_offset[b] * P;
    dim3 ga(ceil_div(P, 256), T, NL);
    if (tgt_dtype == 0)
      hipLaunchKernelGGL(matcher_sample_targets_kernel<float>, ga, dim3(256), 0, st,
                         (const float*)tgt_masks + (int64_t)tgt_offset[b] * Ht * Wt, pts_b, tm_b, Ht, Wt, P,
                         pts_stride_lvl, tm_stride_lvl);
    else
      hipLaunchKernelGGL(matcher_sample_targets_kernel<uint8_t>, ga, dim3(256), 0, st,
                         (const uint8_t*)tgt_masks + (int64_t)tgt_offset[b] * Ht * Wt, pts_b, tm_b, Ht, Wt, P,
                         pts_stride_lvl, tm_stride_lvl);
  }
  dim3 gb(ceil_div(Q, kQG), NL, B);
  if (R) {
    gb.x = ceil_div(Q, kBandQG);
    const size_t lds = (size_t)kBandQG * ((((size_t)(R + 1) * w) + 255) & ~(size_t)255) * 4;
    auto kf = (w % 4 == 0) ? matcher_cost_band_kernel<true, kBandQG> : matcher_cost_band_kernel<false, kBandQG>;
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      set_error("%s: cannot raise dynamic LDS to %zu", who, lds);
      return WM2F_ELAUNCH;
    }
    hipLaunchKernelGGL(kf, gb, dim3(256), lds, st, levels, (const float*)class_logits, (const float*)tm, (const int64_t*)tgt_classes,
                       (const float*)points, (const int*)band_off, (float*)cost, im, B, Q, C1, h, w, P, Tsum, Tmax, R, NB, w_class,
                       w_mask, w_dice);
  } else {
    hipLaunchKernelGGL(matcher_cost_kernel, gb, dim3(256), 0, st, levels,
                       (const float*)class_logits, (const float*)tm, (const int64_t*)tgt_classes,
                       (const float*)points, (float*)cost, im, B, Q, C1, h, w, P, Tsum, Tmax, w_class, w_mask, w_dice);
  }
  WM2F_CHECK_LAUNCH(who);
  return WM2F_OK;
}

extern "C" int wm2f_matcher_cost(const void* mask_logits, const void* class_logits, const void* tgt_masks,
                                 int tgt_dtype, const int32_t* tgt_offset, const void* tgt_classes,
                                 const void* points, void* cost, void* workspace, int NL, int B, int Q, int C1, int h,
                                 int w, int Ht, int Wt, int P, int Tmax, float w_class, float w_mask, float w_dice,
                                 void* stream) {
  const char* who = "wm2f_matcher_cost";
  WM2F_REQUIRE(mask_logits, "%s: null pointer", who);
  WM2F_REQUIRE(NL > 0 && NL <= kMaxMatchLevels && B > 0 && Q > 0 && h > 0 && w > 0, "%s: bad size", who);
  MatcherLevels lv;
  for (int l = 0; l < kMaxMatchLevels; ++l)
    lv.p[l] = (const float*)mask_logits + (int64_t)(l < NL ? l : 0) * B * Q * h * w;
  return matcher_cost_impl(lv, class_logits, tgt_masks, tgt_dtype, tgt_offset, tgt_classes, points, cost, workspace, NL, B,
                           Q, C1, h, w, Ht, Wt, P, Tmax, w_class, w_mask, w_dice, stream, who);
}

extern "C" int wm2f_matcher_cost_levels(const void* const* mask_levels, const void* class_logits, const void* tgt_masks,
                                        int tgt_dtype, const int32_t* tgt_offset, const void* tgt_classes,
                                        const void* points, void* cost, void* workspace, int NL, int B, int Q, int C1,
                                        int h, int w, int Ht, int Wt, int P, int Tmax, float w_class, float w_mask,
                                        float w_dice, void* stream) {
  const char* who = "wm2f_matcher_cost_levels";
  WM2F_REQUIRE(mask_levels, "%s: null pointer", who);
  WM2F_REQUIRE(NL > 0 && NL <= kMaxMatchLevels, "%s: 1..%d levels", who, kMaxMatchLevels);
  MatcherLevels lv;
  for (int l = 0; l < kMaxMatchLevels; ++l) lv.p[l] = (const float*)mask_levels[l < NL ? l : 0];
  for (int l = 0; l < NL; ++l) WM2F_REQUIRE(lv.p[l], "%s: null level pointer", who);
  return matcher_cost_impl(lv, class_logits, tgt_masks, tgt_dtype, tgt_offset, tgt_classes, points, cost, workspace, NL, B,
                           Q, C1, h, w, Ht, Wt, P, Tmax, w_class, w_mask, w_dice, stream, who);
}

extern "C" int wm2f_point_sample_fwd(const void* feat, int feat_dtype, const void* pts, const void* map_index,
                                     void* out, int M, int H, int W, int P, void* stream) {
  const char* who = "wm2f_point_sample_fwd";
  WM2F_REQUIRE(feat && pts && out, "%s: null pointer", who);
  WM2F_REQUIRE(M > 0 && H > 0 && W > 0 && P > 0 && M <= 65535, "%s: bad size", who);
  WM2F_REQUIRE(feat_dtype == 0 || feat_dtype == 1, "%s: feat_dtype must be 0 (fp32) or 1 (uint8)", who);
  dim3 grid(ceil_div(P, 256), M);
  if (feat_dtype == 0)
    hipLaunchKernelGGL(point_sample_fwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)feat,
                       (const float*)pts, (const int32_t*)map_index, (float*)out, H, W, P);
  else
    hipLaunchKernelGGL(point_sample_fwd_kernel<uint8_t>, grid, dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t*)feat, (const float*)pts, (const int32_t*)map_index, (float*)out, H, W, P);
  WM2F_CHECK_LAUNCH(who);
  return WM2F_OK;
}

extern "C" int wm2f_point_sample_bwd(const void* grad_out, const void* pts, const void* map_index, void* grad_feat,
                                     int M, int H, int W, int P, void* stream) {
  const char* who = "wm2f_point_sample_bwd";
  WM2F_REQUIRE(grad_out && pts && grad_feat, "%s: null pointer", who);
  WM2F_REQUIRE(M > 0 && H > 0 && W > 0 && P > 0 && M <= 65535, "%s: bad size", who);
  hipLaunchKernelGGL(point_sample_bwd_kernel, dim3(ceil_div(P, 256), M), dim3(256), 0, (hipStream_t)stream,
                     (const float*)grad_out, (const float*)pts, (const int32_t*)map_index, (float*)grad_feat, H, W, P);
  WM2F_CHECK_LAUNCH(who);
  return WM2F_OK;
}
