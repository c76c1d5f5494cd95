// Point-sampled mask loss, batched over the prediction levels.
// Replaces, for all 10 levels of a step at once, what transformers 5.15.0 modeling_mask2former.py does per level:
//   sample_point on the matched prediction maps (:245-274, called from :602-640 and :671-724),
//   sigmoid_cross_entropy_loss (:308-324) and dice_loss (:278-305) over the sampled points.
// The level tensors stay where the mask predictor wrote them: the kernels take a table of base pointers
// (at most 16 levels, passed by value) instead of a stacked copy (10 x 210 MB at config 2).
//   point_sample_levels_fwd : out[l][m][p] = bilinear(maps[l][index[l][m]], pts[l][m][p])  (or -|.|: the
//                             uncertainty of :688-690)
//   point_sample_levels_bwd : scatter of grad_out into the level gradient maps (fp32 atomics; 4 per point)
//   mask_loss_rows_fwd      : per matched mask r: mean BCE-with-logits over its points, dice = 1 - (2 sum(p t) + 1) /
//                             (sum p + sum t + 1); row sums kept for the backward
//   mask_loss_rows_bwd      : d/dlogit of g_bce[r] * bce_mean[r] + g_dice[r] * dice[r]
// HBM-bound streaming / gather passes over (levels x masks x points) = 1280 x 12544 values; no roofline claim.
#include "common.h"

namespace wm2f {
namespace {

constexpr int kMaxLossLevels = 16;
struct LevelTable {
  const float* p[kMaxLossLevels];
};
struct LevelTableMut {
  float* p[kMaxLossLevels];
};

// grid_sample(bilinear, zeros, align_corners=False) at normalised (x, y) in [0, 1] (HF:245-274 maps them to 2 x - 1)
__device__ __forceinline__ float sample_zeros(const float* __restrict__ img, int H, int W, float lx, float ly) {
  const float gx = 2.f * lx - 1.f, gy = 2.f * ly - 1.f;
  const float x = ((gx + 1.f) * (float)W - 1.f) * 0.5f;
  const float y = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
  if (!(x > -1.f && x < (float)W && y > -1.f && y < (float)H)) return 0.f;
  const float x0f = floorf(x), y0f = floorf(y);
  const int x0 = (int)x0f, y0 = (int)y0f;
  const float fx1 = x - x0f, fy1 = y - y0f, fx0 = 1.f - fx1, fy0 = 1.f - fy1;
  const bool xl = x0 >= 0, xr = x0 + 1 < W, yt = y0 >= 0, yb = y0 + 1 < H;
  const float* p = img + (int64_t)y0 * W + x0;
  float r = 0.f;
  if (yt && xl) r += p[0] * (fx0 * fy0);
  if (yt && xr) r += p[1] * (fx1 * fy0);
  if (yb && xl) r += p[W] * (fx0 * fy1);
  if (yb && xr) r += p[W + 1] * (fx1 * fy1);
  return r;
}

__global__ __launch_bounds__(256) void point_sample_levels_fwd_kernel(LevelTable maps, const float* __restrict__ pts,
                                                                      const int32_t* __restrict__ index,
                                                                      float* __restrict__ out, int M, int H, int W, int P,
                                                                      int neg_abs) {
  const int p = blockIdx.x * 256 + threadIdx.x, m = blockIdx.y, l = blockIdx.z;
  if (p >= P) return;
  const int64_t row = (int64_t)l * M + m;
  const float* pp = pts + (row * P + p) * 2;
  const float v = sample_zeros(maps.p[l] + (int64_t)index[row] * H * W, H, W, pp[0], pp[1]);
  out[row * P + p] = neg_abs ? -fabsf(v) : v;
}

// Band form of the forward for many points per map (the uncertainty sampling: 37 632 random points on each matched 256 x 256 map).
// The gather form reads 4 scattered floats per point -- 64 different 128-byte lines per wave instruction, 32 x the bytes it uses
// (2 ms per config-2 step for 1.5 GB of useful floats).  Here a workgroup loads a BAND of rows [y_lo, y_hi] of its map into LDS
// with coalesced float4 loads (the map moves once), walks ALL the row's points (coalesced float2 loads) and samples those whose
// top corner row lies in its band from LDS; every point is written by exactly one band (points outside the image by band 0).
// Same arithmetic in the same order as sample_zeros: identical bits.
__global__ __launch_bounds__(1024) void point_sample_levels_fwd_band_kernel(LevelTable maps, const float* __restrict__ pts,
                                                                            const int32_t* __restrict__ index, float* __restrict__ out,
                                                                            int M, int H, int W, int P, int neg_abs, int band_rows) {
  extern __shared__ __attribute__((aligned(16))) float band[];  // rows y_lo .. y_hi (inclusive) of the map
  const int bnd = blockIdx.x, m = blockIdx.y, l = blockIdx.z, tid = threadIdx.x;
  const int y_lo = bnd * band_rows, y_hi = min(H - 1, y_lo + band_rows);  // one row beyond the band's last top-corner row
  const int n_px = (y_hi - y_lo + 1) * W;
  const int64_t row = (int64_t)l * M + m;
  const float* src = maps.p[l] + (int64_t)index[row] * H * W + (int64_t)y_lo * W;
  if ((n_px & 3) == 0 && (W & 3) == 0) {
    for (int i = tid; i < n_px / 4; i += 1024) reinterpret_cast<float4*>(band)[i] = reinterpret_cast<const float4*>(src)[i];
  } else {
    for (int i = tid; i < n_px; i += 1024) band[i] = src[i];
  }
  __syncthreads();
  const float2* pp = reinterpret_cast<const float2*>(pts) + row * P;
  float* op = out + row * P;
  const int own_hi = min(H, y_lo + band_rows);  // this band owns top-corner rows [y_lo, own_hi) -- band 0 also row -1
  for (int p = tid; p < P; p += 1024) {
    const float2 pt = pp[p];
    const float gx = 2.f * pt.x - 1.f, gy = 2.f * pt.y - 1.f;
    const float x = ((gx + 1.f) * (float)W - 1.f) * 0.5f;
    const float y = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
    if (!(x > -1.f && x < (float)W && y > -1.f && y < (float)H)) {
      if (bnd == 0) op[p] = 0.f;  // (neg_abs of 0 is 0)
      continue;
    }
    const float x0f = floorf(x), y0f = floorf(y);
    const int x0 = (int)x0f, y0 = (int)y0f;
    if (!((y0 >= y_lo && y0 < own_hi) || (bnd == 0 && y0 < 0))) continue;
    const float fx1 = x - x0f, fy1 = y - y0f, fx0 = 1.f - fx1, fy0 = 1.f - fy1;
    const bool xl = x0 >= 0, xr = x0 + 1 < W, yt = y0 >= 0, yb = y0 + 1 < H;
    const float* q = band + (y0 - y_lo) * W + x0;
    float r = 0.f;
    if (yt && xl) r += q[0] * (fx0 * fy0);
    if (yt && xr) r += q[1] * (fx1 * fy0);
    if (yb && xl) r += q[W] * (fx0 * fy1);
    if (yb && xr) r += q[W + 1] * (fx1 * fy1);
    op[p] = neg_abs ? -fabsf(r) : r;
  }
}

__global__ __launch_bounds__(256) void point_sample_levels_bwd_kernel(const float* __restrict__ grad_out,
                                                                      const float* __restrict__ pts,
                                                                      const int32_t* __restrict__ index, LevelTableMut grads,
                                                                      int M, int H, int W, int P) {
  const int p = blockIdx.x * 256 + threadIdx.x, m = blockIdx.y, l = blockIdx.z;
  if (p >= P) return;
  const int64_t row = (int64_t)l * M + m;
  const float* pp = pts + (row * P + p) * 2;
  const float go = grad_out[row * P + p];
  const float x = ((2.f * pp[0] - 1.f + 1.f) * (float)W - 1.f) * 0.5f;
  const float y = ((2.f * pp[1] - 1.f + 1.f) * (float)H - 1.f) * 0.5f;
  if (!(x > -1.f && x < (float)W && y > -1.f && y < (float)H)) return;
  const float x0f = floorf(x), y0f = floorf(y);
  const int x0 = (int)x0f, y0 = (int)y0f;
  const float fx1 = x - x0f, fy1 = y - y0f, fx0 = 1.f - fx1, fy0 = 1.f - fy1;
  const bool xl = x0 >= 0, xr = x0 + 1 < W, yt = y0 >= 0, yb = y0 + 1 < H;
  float* g = grads.p[l] + (int64_t)index[row] * H * W + (int64_t)y0 * W + x0;
  if (yt && xl) atomicAdd(g, go * fx0 * fy0);
  if (yt && xr) atomicAdd(g + 1, go * fx1 * fy0);
  if (yb && xl) atomicAdd(g + W, go * fx0 * fy1);
  if (yb && xr) atomicAdd(g + W + 1, go * fx1 * fy1);
}

// The same scatter when every (level, index) pair is distinct (the matched rows of a one-to-one assignment): a workgroup owns
// a BAND of one indexed map in LDS (16 K pixels = 64 KiB), walks all P points of the row, adds the corners that fall into its
// band with LDS atomics, and writes the band back with plain coalesced stores -- the map is OVERWRITTEN (no clearing needed,
// no global atomics: 64 M of them per step cost 3 ms at config 2).
__global__ __launch_bounds__(256) void point_sample_levels_bwd_band_kernel(const float* __restrict__ grad_out,
                                                                           const float* __restrict__ pts,
                                                                           const int32_t* __restrict__ index, LevelTableMut grads,
                                                                           int M, int H, int W, int P, int band_rows) {
  extern __shared__ __attribute__((aligned(16))) float band[];  // [band_rows][W]
  const int bnd = blockIdx.x, m = blockIdx.y, l = blockIdx.z, tid = threadIdx.x;
  const int y_lo = bnd * band_rows, y_hi = min(H, y_lo + band_rows);
  const int n_px = (y_hi - y_lo) * W;
  for (int i = tid; i < n_px; i += 256) band[i] = 0.f;
  __syncthreads();
  const int64_t row = (int64_t)l * M + m;
  const float2* pp = reinterpret_cast<const float2*>(pts) + row * P;
  const float* gp = grad_out + row * P;
  for (int p = tid; p < P; p += 256) {
    const float2 pt = pp[p];
    const float go = gp[p];
    const float x = ((2.f * pt.x - 1.f + 1.f) * (float)W - 1.f) * 0.5f;
    const float y = ((2.f * pt.y - 1.f + 1.f) * (float)H - 1.f) * 0.5f;
    if (!(x > -1.f && x < (float)W && y > -1.f && y < (float)H)) continue;
    const float x0f = floorf(x), y0f = floorf(y);
    const int x0 = (int)x0f, y0 = (int)y0f;
    if (y0 + 1 < y_lo || y0 >= y_hi) continue;
    const float fx1 = x - x0f, fy1 = y - y0f, fx0 = 1.f - fx1, fy0 = 1.f - fy1;
    const bool xl = x0 >= 0, xr = x0 + 1 < W;
    const bool yt = y0 >= y_lo, yb = y0 + 1 < y_hi;  // inside the band implies inside the image
    float* g = band + (y0 - y_lo) * W + x0;
    if (yt && xl) atomicAdd(g, go * fx0 * fy0);
    if (yt && xr) atomicAdd(g + 1, go * fx1 * fy0);
    if (yb && xl) atomicAdd(g + W, go * fx0 * fy1);
    if (yb && xr) atomicAdd(g + W + 1, go * fx1 * fy1);
  }
  __syncthreads();
  float* dst = grads.p[l] + (int64_t)index[row] * H * W + (int64_t)y_lo * W;
  if ((n_px & 3) == 0 && (W & 3) == 0) {
    for (int i = tid; i < n_px / 4; i += 256) reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(band)[i];
  } else {
    for (int i = tid; i < n_px; i += 256) dst[i] = band[i];
  }
}

__device__ __forceinline__ float block_sum256(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  const float t = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  return t;
}

// sums (R, 4): sum of BCE terms, sum p t, sum p, sum t;  bce_mean (R), dice (R)
__global__ __launch_bounds__(256) void mask_loss_rows_fwd_kernel(const float* __restrict__ logits,
                                                                 const float* __restrict__ labels, float* __restrict__ sums,
                                                                 float* __restrict__ bce_mean, float* __restrict__ dice,
                                                                 int P) {
  __shared__ float red[4];
  const int64_t r = blockIdx.x;
  const float *x = logits + r * P, *t = labels + r * P;
  float sb = 0.f, spt = 0.f, sp = 0.f, st = 0.f;
  for (int i = threadIdx.x; i < P; i += 256) {
    const float xi = x[i], ti = t[i];
    // binary_cross_entropy_with_logits: max(x, 0) - x t + log1p(exp(-|x|))
    sb += fmaxf(xi, 0.f) - xi * ti + log1pf(__expf(-fabsf(xi)));
    const float pi = 1.f / (1.f + __expf(-xi));
    spt += pi * ti;
    sp += pi;
    st += ti;
  }
  sb = block_sum256(sb, red);
  spt = block_sum256(spt, red);
  sp = block_sum256(sp, red);
  st = block_sum256(st, red);
  if (threadIdx.x == 0) {
    sums[r * 4 + 0] = sb;
    sums[r * 4 + 1] = spt;
    sums[r * 4 + 2] = sp;
    sums[r * 4 + 3] = st;
    bce_mean[r] = sb / (float)P;                              // HF:321-323: mean over points
    dice[r] = 1.f - (2.f * spt + 1.f) / (sp + st + 1.f);      // HF:299-303
  }
}

__global__ __launch_bounds__(256) void mask_loss_rows_bwd_kernel(const float* __restrict__ logits,
                                                                 const float* __restrict__ labels,
                                                                 const float* __restrict__ sums,
                                                                 const float* __restrict__ g_bce, const float* __restrict__ g_dice,
                                                                 float* __restrict__ grad, int P) {
  const int64_t r = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const float xi = logits[r * P + i], ti = labels[r * P + i];
  const float pi = 1.f / (1.f + __expf(-xi));
  const float N = 2.f * sums[r * 4 + 1] + 1.f, Dn = sums[r * 4 + 2] + sums[r * 4 + 3] + 1.f;
  // d bce_mean / dx = (p - t) / P;   d dice / dp = -(2 t Dn - N) / Dn^2,  dp/dx = p (1 - p)
  const float gd = -(2.f * ti * Dn - N) / (Dn * Dn) * pi * (1.f - pi);
  grad[r * P + i] = g_bce[r] * (pi - ti) / (float)P + g_dice[r] * gd;
}

}  // namespace
}  // namespace wm2f

using namespace wm2f;

// ---------------------------------------------------------------------------------- importance sampling: the k most uncertain points
// HF:688-704 (sample_points_using_uncertainty): idx = topk(uncertainty, k)[1]; points = gather(coords, idx).  A top-k with k in the
// thousands is a full sort in the stock library (one segmented radix sort of 2560 x 37632 keys: 1.4 ms per config-2 step).  The
// losses that follow are sums over the points, so only the SET matters: a workgroup per row holds the row's keys in LDS,
// finds the k-th largest by radix selection (four 8-bit digits, most significant first, one 256-bin histogram each) and writes
// the selected points in index order (equal keys at the threshold: lowest indices first).  NaN ranks highest, as in torch.topk.
namespace wm2f {
namespace {

constexpr int kSelThreads = 1024;
constexpr int kSelMaxN = 38400;  // keys per row that fit LDS (150 KiB)

__device__ __forceinline__ unsigned sel_key(float v) {  // larger float (NaN largest) -> larger unsigned
  const unsigned u = __float_as_uint(v);
  if (v != v) return 0xffffffffu;
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(kSelThreads) void select_top_points_kernel(const float* __restrict__ score, const float* __restrict__ pts,
                                                                        float* __restrict__ out, int n, int k, int out_row_floats) {
  extern __shared__ unsigned keys[];  // [n]
  __shared__ int hist[256];
  __shared__ unsigned s_prefix;
  __shared__ int s_krem, s_krem_next, wcnt[2][kSelThreads / 64];  // [greater / equal][wave]
  __shared__ int hsuf[256], wtot[4];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* sc = score + (int64_t)row * n;
  const float2* pr = reinterpret_cast<const float2*>(pts) + (int64_t)row * n;
  float2* po = reinterpret_cast<float2*>(out + (int64_t)row * out_row_floats);
  // (eight loads in flight per lane: one at a time, this loop alone was 37 trips to memory per row)
  for (int i0 = tid; i0 < n; i0 += 8 * kSelThreads) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = i0 + e * kSelThreads < n ? sc[i0 + e * kSelThreads] : 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (i0 + e * kSelThreads < n) keys[i0 + e * kSelThreads] = sel_key(v[e]);
  }
  if (tid == 0) {
    s_prefix = 0u;
    s_krem = k;
  }
  // radix selection: after the round of digit d, s_prefix holds the top (d + 1) digits of the k-th largest key and s_krem how
  // many keys with exactly that prefix are still to be taken
  for (int shift = 24; shift >= 0; shift -= 8) {
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const unsigned prefix = s_prefix, himask = shift == 24 ? 0u : (0xffffffffu << (shift + 8));
    for (int i0 = 0; i0 < n; i0 += kSelThreads) {
      const int i = i0 + tid;
      const unsigned key = i < n ? keys[i] : 0u;
      bool live = i < n && (key & himask) == prefix;
      const int bin = (int)((key >> shift) & 255u);
      // The scores of one row share sign and exponent: in the first rounds nearly every lane of a wave hits ONE bin, and 64 LDS
      // atomics on one address run one after the other (the first form of this kernel took as long as the sort it replaces).
      // "The first live lane's bin: one add of its population count" takes those lanes out; what is left is spread.
      const unsigned long long m = __builtin_amdgcn_ballot_w64(live);
      if (m == 0) continue;  // (uniform: from the third digit on almost every wave, almost every time)
      const int first = __builtin_ctzll(m), lead = __builtin_amdgcn_readlane(bin, first);
      const unsigned long long same = __builtin_amdgcn_ballot_w64(live && bin == lead);
      if (lane == first) atomicAdd(&hist[lead], __builtin_popcountll(same));
      if (live && bin != lead) atomicAdd(&hist[bin], 1);
    }
    __syncthreads();
    // which digit holds the rank-th largest: bin d with (keys in bins above d) < rank <= (keys in bins above d) + hist[d].
    // 256 lanes, one bin each; counts of the bins above by a suffix sum inside the wave + the totals of the waves above
    // (thread 0 walking the 256 bins one dependent LDS read at a time was a quarter of the kernel)
    if (tid < 256) {
      const int hme = hist[tid];
      int suf = hme;  // inclusive suffix sum over the wave's lanes >= lane
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_down(suf, o, 64);
        if (lane + o < 64) suf += up;
      }
      if (lane == 0) wtot[wave] = suf;  // the wave's total
      // (waves 0..3 only: a named barrier is not needed -- they meet at the workgroup barrier below; the totals are read after it)
      hsuf[tid] = suf - hme;  // keys in this wave's bins above this one
    }
    __syncthreads();
    if (tid < 256) {
      int above = hsuf[tid];
      for (int w2 = wave + 1; w2 < 4; ++w2) above += wtot[w2];
      const int rem = s_krem, hme = hist[tid];
      if (above < rem && rem <= above + hme) {  // exactly one bin
        s_prefix = prefix | ((unsigned)tid << shift);
        s_krem_next = rem - above;
      }
    }
    __syncthreads();
    if (tid == 0) s_krem = s_krem_next;
    __syncthreads();
  }
  const unsigned thr = s_prefix;  // the k-th largest key; eq_take of the keys equal to it are taken, lowest indices first
  const int eq_take = s_krem;
  // compaction in index order: position = selected before me = greater before me + min(equal before me, eq_take).  Every wave
  // owns a contiguous segment of the row: it counts its segment, ONE barrier publishes the counts, and it then walks the
  // segment again with running counters (a prefix over the 16 waves per 1024-key chunk cost 32 LDS reads and a barrier per chunk).
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  constexpr int kW = kSelThreads / 64;
  const int seg = ((n + kW - 1) / kW + 63) & ~63, s_lo = wave * seg, s_hi = s_lo + seg < n ? s_lo + seg : n;
  int gt_cnt = 0, eq_cnt = 0;
  for (int i0 = s_lo; i0 < s_hi; i0 += 64) {
    const int i = i0 + lane;
    const unsigned key = i < s_hi ? keys[i] : 0u;
    gt_cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(i < s_hi && key > thr));
    eq_cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(i < s_hi && key == thr));
  }
  if (lane == 0) {
    wcnt[0][wave] = gt_cnt;
    wcnt[1][wave] = eq_cnt;
  }
  __syncthreads();
  int gt_seen = 0, eq_seen = 0;
  for (int w2 = 0; w2 < wave; ++w2) {
    gt_seen += wcnt[0][w2];
    eq_seen += wcnt[1][w2];
  }
  for (int i0 = s_lo; i0 < s_hi; i0 += 64) {
    const int i = i0 + lane;
    const unsigned key = i < s_hi ? keys[i] : 0u;
    const bool gt = i < s_hi && key > thr, eq = i < s_hi && key == thr;
    const unsigned long long mg = __builtin_amdgcn_ballot_w64(gt), me = __builtin_amdgcn_ballot_w64(eq);
    const int gt_before = gt_seen + __builtin_popcountll(mg & lt), eq_before = eq_seen + __builtin_popcountll(me & lt);
    // the output position replaces the key (0xffffffff: not selected); the copies follow in a loop of their own
    if (i < s_hi) keys[i] = (gt || (eq && eq_before < eq_take)) ? (unsigned)(gt_before + (eq_before < eq_take ? eq_before : eq_take)) : 0xffffffffu;
    gt_seen += __builtin_popcountll(mg);
    eq_seen += __builtin_popcountll(me);
  }
  __syncthreads();
  for (int i0 = tid; i0 < n; i0 += 8 * kSelThreads) {
    float2 v[8];
    unsigned pos[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int i = i0 + e * kSelThreads;
      pos[e] = i < n ? keys[i] : 0xffffffffu;
      if (pos[e] != 0xffffffffu) v[e] = pr[i];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (pos[e] != 0xffffffffu) po[pos[e]] = v[e];
  }
}

}  // namespace
}  // namespace wm2f

extern "C" int wm2f_select_top_points(const void* score, const void* pts, void* out, int rows, int n, int k, int out_row_points,
                                      void* stream) {
  const char* who = "wm2f_select_top_points";
  WM2F_REQUIRE(score && pts && out, "%s: null pointer", who);
  WM2F_REQUIRE(rows > 0 && n > 0 && k > 0 && k <= n && out_row_points >= k, "%s: sizes", who);
  if (n > wm2f::kSelMaxN) {
    wm2f::set_error("%s: %d candidates per row exceed %d (use a sort)", who, n, wm2f::kSelMaxN);
    return WM2F_EUNSUPPORTED;
  }
  const size_t lds = (size_t)n * 4;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)wm2f::select_top_points_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, wm2f::kSelMaxN * 4) != hipSuccess) {
      wm2f::set_error("%s: cannot reserve %d bytes of LDS", who, wm2f::kSelMaxN * 4);
      return WM2F_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(wm2f::select_top_points_kernel, dim3(rows), dim3(wm2f::kSelThreads), lds, (hipStream_t)stream, (const float*)score,
                     (const float*)pts, (float*)out, n, k, out_row_points * 2);
  WM2F_CHECK_LAUNCH(who);
  return WM2F_OK;
}

extern "C" int wm2f_point_sample_levels_fwd(const void* const* level_maps, int n_levels, const void* pts, const int32_t* index,
                                            void* out, int M, int H, int W, int P, int neg_abs, void* stream) {
  const char* who = "wm2f_point_sample_levels_fwd";
  WM2F_REQUIRE(level_maps && pts && index && out, "%s: null pointer", who);
  WM2F_REQUIRE(n_levels > 0 && n_levels <= kMaxLossLevels, "%s: 1..%d levels", who, kMaxLossLevels);
  WM2F_REQUIRE(M > 0 && M < 65536 && H > 0 && W > 0 && P > 0, "%s: bad size", who);
  LevelTable tab;
  for (int l = 0; l < kMaxLossLevels; ++l) tab.p[l] = (const float*)level_maps[l < n_levels ? l : 0];
  for (int l = 0; l < n_levels; ++l) WM2F_REQUIRE(tab.p[l], "%s: null level pointer", who);
  // many points per map and a map of at most two LDS bands: the band form (the map moves once, the points twice)
  const int band_rows = (H + 1) / 2;
  const size_t band_bytes = (size_t)(band_rows + 1) * W * sizeof(float);
  if (P >= 4096 && band_bytes <= 150 * 1024 && H >= 2) {
    static bool attr_set = false;
    if (!attr_set) {
      if (hipFuncSetAttribute((const void*)point_sample_levels_fwd_band_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) {
        set_error("%s: cannot reserve LDS for the band form", who);
        return WM2F_ELAUNCH;
      }
      attr_set = true;
    }
    hipLaunchKernelGGL(point_sample_levels_fwd_band_kernel, dim3(ceil_div(H, band_rows), M, n_levels), dim3(1024), band_bytes,
                       (hipStream_t)stream, tab, (const float*)pts, index, (float*)out, M, H, W, P, neg_abs, band_rows);
  } else {
    hipLaunchKernelGGL(point_sample_levels_fwd_kernel, dim3(ceil_div(P, 256), M, n_levels), dim3(256), 0, (hipStream_t)stream,
                       tab, (const float*)pts, index, (float*)out, M, H, W, P, neg_abs);
  }
  WM2F_CHECK_LAUNCH(who);
  return WM2F_OK;
}

extern "C" int wm2f_point_sample_levels_bwd(const void* grad_out, const void* pts, const int32_t* index,
                                            void* const* level_grads, int n_levels, int M, int H, int W, int P,
                                            void* stream) {
  const char* who = "wm2f_point_sample_levels_bwd";
  WM2F_REQUIRE(grad_out && pts && index && level_grads, "%s: null pointer", who);
  WM2F_REQUIRE(n_levels > 0 && n_levels <= kMaxLossLevels, "%s: 1..%d levels", who, kMaxLossLevels);
  WM2F_REQUIRE(M > 0 && M < 65536 && H > 0 && W > 0 && P > 0, "%s: bad size", who);
  LevelTableMut tab;
  for (int l = 0; l < kMaxLossLevels; ++l) tab.p[l] = (float*)level_grads[l < n_levels ? l : 0];
  for (int l = 0; l < n_levels; ++l) WM2F_REQUIRE(tab.p[l], "%s: null level pointer", who);
  hipLaunchKernelGGL(point_sample_levels_bwd_kernel, dim3(ceil_div(P, 256), M, n_levels), dim3(256), 0, (hipStream_t)stream,
                     (const float*)grad_out, (const float*)pts, index, tab, M, H, W, P);
  WM2F_CHECK_LAUNCH(who);
  return WM2F_OK;
}

extern "C" int wm2f_mask_loss_rows_fwd(const void* logits, const void* labels, void* sums, void* bce_mean, void* dice,
                                       int R, int P, void* stream) {
  const char* who = "wm2f_mask_loss_rows_fwd";
  WM2F_REQUIRE(logits && labels && sums && bce_mean && dice, "%s: null pointer", who);
  WM2F_REQUIRE(R > 0 && P > 0, "%s: non-positive size", who);
  hipLaunchKernelGGL(mask_loss_rows_fwd_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, (const float*)logits,
                     (const float*)labels, (float*)sums, (float*)bce_mean, (float*)dice, P);
  WM2F_CHECK_LAUNCH(who);
  return WM2F_OK;
}

extern "C" int wm2f_mask_loss_rows_bwd(const void* logits, const void* labels, const void* sums, const void* g_bce,
                                       const void* g_dice, void* grad, int R, int P, void* stream) {
  const char* who = "wm2f_mask_loss_rows_bwd";
  WM2F_REQUIRE(logits && labels && sums && g_bce && g_dice && grad, "%s: null pointer", who);
  WM2F_REQUIRE(R > 0 && R < 65536 && P > 0, "%s: bad size", who);
  hipLaunchKernelGGL(mask_loss_rows_bwd_kernel, dim3(ceil_div(P, 256), R), dim3(256), 0, (hipStream_t)stream,
                     (const float*)logits, (const float*)labels, (const float*)sums, (const float*)g_bce,
                     (const float*)g_dice, (float*)grad, P);
  WM2F_CHECK_LAUNCH(who);
  return WM2F_OK;
}

extern "C" int wm2f_point_sample_levels_bwd_unique(const void* grad_out, const void* pts, const int32_t* index,
                                                   void* const* level_grads, int n_levels, int M, int H, int W, int P,
                                                   void* stream) {
  const char* who = "wm2f_point_sample_levels_bwd_unique";
  WM2F_REQUIRE(grad_out && pts && index && level_grads, "%s: null pointer", who);
  WM2F_REQUIRE(n_levels > 0 && n_levels <= kMaxLossLevels, "%s: 1..%d levels", who, kMaxLossLevels);
  WM2F_REQUIRE(M > 0 && M < 65536 && H > 0 && W > 0 && P > 0, "%s: bad size", who);
  if (W > 16384) {
    set_error("%s: maps wider than 16384 pixels do not fit a band", who);
    return WM2F_EUNSUPPORTED;
  }
  LevelTableMut tab;
  for (int l = 0; l < kMaxLossLevels; ++l) tab.p[l] = (float*)level_grads[l < n_levels ? l : 0];
  for (int l = 0; l < n_levels; ++l) WM2F_REQUIRE(tab.p[l], "%s: null level pointer", who);
  int band_rows = 16384 / W;  // 64 KiB of LDS
  if (band_rows > H) band_rows = H;
  const int bands = ceil_div(H, band_rows);
  const size_t lds = (size_t)band_rows * W * 4;
  hipLaunchKernelGGL(point_sample_levels_bwd_band_kernel, dim3(bands, M, n_levels), dim3(256), lds, (hipStream_t)stream,
                     (const float*)grad_out, (const float*)pts, index, tab, M, H, W, P, band_rows);
  WM2F_CHECK_LAUNCH(who);
  return WM2F_OK;
}
