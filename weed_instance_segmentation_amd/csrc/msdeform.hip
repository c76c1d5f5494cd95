// K1: multi-scale deformable attention core (+ fused module prologue) and its backward.
// Replaces multi_scale_deformable_attention(), transformers modeling_mask2former.py:798-837,
// and (fused variant) the location / softmax prologue at :983-1002.
//
// Data layout in HBM (all fp32, row-major):
//   value (B, S, heads, D)  one token row = heads*D floats; one head slice = D floats
//                           (D = 32 -> 128 B = exactly one cache line)
//   loc   (B, Q, heads, L, P, 2), attn_w (B, Q, heads, L, P), out (B, Q, heads*D)
//
// Work decomposition: D/4 lanes own one (query, head) pair, each lane a float4 of channels, so a
// bilinear corner is ONE 16-B load per lane and the D/4 lanes of a pair read one contiguous
// head slice.  With heads*D = 256 a wave covers one query's 8 heads: its loc / attn_w reads and
// its 1-KiB output row are fully contiguous.
//
// XCD-aware launch: blocks that share an XCD (blockIdx % 8 equal) take a CONTIGUOUS range of
// queries in raster order, so at B = 8 every XCD's private 4-MiB L2 serves one image, and the
// rows being sampled by the blocks in flight stay L2-resident (speed only, never correctness).
//
// Roofline: HBM.  Algorithmic bytes per call = value + loc + attn_w + out
//   = 4*(B*S*heads*D + B*Q*heads*L*P*3 + B*Q*heads*D)  (550 502 400 B at config 2).
#include "common.h"

namespace wm2f {

template <int D>
struct PairMap {
  static constexpr int LPG = D / 4;        // lanes per (query, head) pair
  static constexpr int PPW = kWave / LPG;  // pairs per wave
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

__device__ __forceinline__ void fma4(float4& acc, float s, const float4& v) {
  acc.x = fmaf(s, v.x, acc.x);
  acc.y = fmaf(s, v.y, acc.y);
  acc.z = fmaf(s, v.z, acc.z);
  acc.w = fmaf(s, v.w, acc.w);
}

// One sampling point: accumulate aw * bilinear(value_level, (lx, ly)) into acc.
// Pixel coordinates follow grid_sample(align_corners=False): x = ((2*lx-1 + 1) * W - 1) / 2.
__device__ __forceinline__ void sample_point_acc(float4& acc, const float* __restrict__ vlev, int Hl, int Wl,
                                                 int row_stride, float lx, float ly, float aw) {
  const float gx = 2.f * lx - 1.f, gy = 2.f * ly - 1.f;
  const float x = ((gx + 1.f) * (float)Wl - 1.f) * 0.5f;
  const float y = ((gy + 1.f) * (float)Hl - 1.f) * 0.5f;
  if (!(x > -1.f && x < (float)Wl && y > -1.f && y < (float)Hl)) return;  // all 4 corners are padding
  const float x0f = floorf(x), y0f = floorf(y);
  const int x0 = (int)x0f, y0 = (int)y0f;
  const float fx1 = x - x0f, fy1 = y - y0f;  // weight of the +1 neighbour
  const float fx0 = 1.f - fx1, fy0 = 1.f - fy1;
  const bool xl = x0 >= 0, xr = x0 + 1 < Wl, yt = y0 >= 0, yb = y0 + 1 < Hl;
  const float* p00 = vlev + (int64_t)(y0 * Wl + x0) * row_stride;
  float4 v00 = make_float4(0.f, 0.f, 0.f, 0.f), v01 = v00, v10 = v00, v11 = v00;
  if (yt && xl) v00 = ld4(p00);
  if (yt && xr) v01 = ld4(p00 + row_stride);
  if (yb && xl) v10 = ld4(p00 + (int64_t)Wl * row_stride);
  if (yb && xr) v11 = ld4(p00 + (int64_t)(Wl + 1) * row_stride);
  fma4(acc, aw * fy0 * fx0, v00);
  fma4(acc, aw * fy0 * fx1, v01);
  fma4(acc, aw * fy1 * fx0, v10);
  fma4(acc, aw * fy1 * fx1, v11);
}

// FUSED = false: loc_in = loc, w_in = attn_w.   FUSED = true: loc_in = raw offsets, w_in = raw logits.
template <int D, bool FUSED, int PC /* compile-time P, 0 = runtime */>
__global__ __launch_bounds__(256) void msdeform_fwd_kernel(const float* __restrict__ value,
                                                           const float* __restrict__ loc_in,
                                                           const float* __restrict__ w_in,
                                                           const float* __restrict__ ref, float* __restrict__ out,
                                                           LevelInfo lv, int S, int Q, int heads, int L, int Prt,
                                                           int64_t n_pairs, int blocks_per_xcd, int n_blocks) {
  using M = PairMap<D>;
  const int P = PC ? PC : Prt;
  const int lb = xcd_contiguous_id(blockIdx.x, blocks_per_xcd);
  if (lb >= n_blocks) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t pair = ((int64_t)lb * 4 + wave) * M::PPW + lane / M::LPG;
  if (pair >= n_pairs) return;
  const int sub = lane % M::LPG;
  const int h = (int)(pair % heads);
  const int64_t bq = pair / heads;
  const int q = (int)(bq % Q);
  const int b = (int)(bq / Q);
  const int row_stride = heads * D;
  const float* vb = value + ((int64_t)b * S * heads + h) * D + sub * 4;
  const float* lp = loc_in + pair * (int64_t)(L * P * 2);
  const float* wp = w_in + pair * (int64_t)(L * P);

  float wmax = 0.f, winv = 1.f;
  if (FUSED) {  // softmax over the L*P logits of this (query, head), HF:986-991
    wmax = -INFINITY;
    for (int i = 0; i < L * P; ++i) wmax = fmaxf(wmax, wp[i]);
    float s = 0.f;
    for (int i = 0; i < L * P; ++i) s += __expf(wp[i] - wmax);
    winv = 1.f / s;
  }

  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int l = 0; l < L; ++l) {
    const int Hl = lv.h[l], Wl = lv.w[l];
    const float* vlev = vb + (int64_t)lv.start[l] * row_stride;
    float rx = 0.f, ry = 0.f;
    if (FUSED) {
      rx = ref[(q * L + l) * 2 + 0];
      ry = ref[(q * L + l) * 2 + 1];
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      float lx = lp[(l * P + p) * 2 + 0];
      float ly = lp[(l * P + p) * 2 + 1];
      float aw = wp[l * P + p];
      if (FUSED) {  // loc = ref + offset / (W_l, H_l), HF:993-1002
        lx = rx + lx / (float)Wl;
        ly = ry + ly / (float)Hl;
        aw = __expf(aw - wmax) * winv;
      }
      sample_point_acc(acc, vlev, Hl, Wl, row_stride, lx, ly, aw);
    }
  }
  *reinterpret_cast<float4*>(out + pair * D + sub * 4) = acc;
}

// ------------------------------------------------------------------------------------ backward
// grad_value is accumulated with float atomics (caller zeroes it); grad_loc / grad_attn_w are
// reduced over the D/4 lanes of a pair with wave shuffles and written by lane 0 of the pair.
template <int LPG>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int m = LPG / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWave);
  return v;
}

__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
  return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
}

__device__ __forceinline__ void atomic_add4(float* p, float s, const float4& g) {
  atomicAdd(p + 0, s * g.x);
  atomicAdd(p + 1, s * g.y);
  atomicAdd(p + 2, s * g.z);
  atomicAdd(p + 3, s * g.w);
}

template <int D>
__global__ __launch_bounds__(256) void msdeform_bwd_kernel(const float* __restrict__ value,
                                                           const float* __restrict__ loc,
                                                           const float* __restrict__ attn_w,
                                                           const float* __restrict__ grad_out,
                                                           float* __restrict__ grad_value, float* __restrict__ grad_loc,
                                                           float* __restrict__ grad_w, LevelInfo lv, int S, int Q,
                                                           int heads, int L, int P, int64_t n_pairs,
                                                           int blocks_per_xcd, int n_blocks) {
  using M = PairMap<D>;
  const int lb = xcd_contiguous_id(blockIdx.x, blocks_per_xcd);
  if (lb >= n_blocks) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int64_t pair = ((int64_t)lb * 4 + wave) * M::PPW + lane / M::LPG;
  const bool live = pair < n_pairs;  // keep every lane in the shuffles
  if (!live) pair = n_pairs - 1;
  const int sub = lane % M::LPG;
  const int h = (int)(pair % heads);
  const int64_t bq = pair / heads;
  const int b = (int)(bq / Q);
  const int row_stride = heads * D;
  const int64_t voff = ((int64_t)b * S * heads + h) * D + sub * 4;
  const float* lp = loc + pair * (int64_t)(L * P * 2);
  const float* wp = attn_w + pair * (int64_t)(L * P);
  const float4 go = ld4(grad_out + pair * D + sub * 4);

  for (int l = 0; l < L; ++l) {
    const int Hl = lv.h[l], Wl = lv.w[l];
    const int64_t lev_off = voff + (int64_t)lv.start[l] * row_stride;
    for (int p = 0; p < P; ++p) {
      const float lx = lp[(l * P + p) * 2 + 0], ly = lp[(l * P + p) * 2 + 1];
      const float aw = wp[l * P + p];
      const float gx = 2.f * lx - 1.f, gy = 2.f * ly - 1.f;
      const float x = ((gx + 1.f) * (float)Wl - 1.f) * 0.5f;
      const float y = ((gy + 1.f) * (float)Hl - 1.f) * 0.5f;
      float g_w = 0.f, g_x = 0.f, g_y = 0.f;
      if (x > -1.f && x < (float)Wl && y > -1.f && y < (float)Hl) {
        const float x0f = floorf(x), y0f = floorf(y);
        const int x0 = (int)x0f, y0 = (int)y0f;
        const float fx1 = x - x0f, fy1 = y - y0f, fx0 = 1.f - fx1, fy0 = 1.f - fy1;
        const bool xl = x0 >= 0, xr = x0 + 1 < Wl, yt = y0 >= 0, yb = y0 + 1 < Hl;
        const int64_t o00 = lev_off + (int64_t)(y0 * Wl + x0) * row_stride;
        const int64_t o01 = o00 + row_stride, o10 = o00 + (int64_t)Wl * row_stride, o11 = o10 + row_stride;
        float4 z = make_float4(0.f, 0.f, 0.f, 0.f), v00 = z, v01 = z, v10 = z, v11 = z;
        if (yt && xl) v00 = ld4(value + o00);
        if (yt && xr) v01 = ld4(value + o01);
        if (yb && xl) v10 = ld4(value + o10);
        if (yb && xr) v11 = ld4(value + o11);
        const float d00 = dot4(go, v00), d01 = dot4(go, v01), d10 = dot4(go, v10), d11 = dot4(go, v11);
        g_w = fy0 * (fx0 * d00 + fx1 * d01) + fy1 * (fx0 * d10 + fx1 * d11);
        // d sample / d x (pixels) and / d y, then chain through x = lx*W - 0.5
        g_x = aw * (float)Wl * (fy0 * (d01 - d00) + fy1 * (d11 - d10));
        g_y = aw * (float)Hl * (fx0 * (d10 - d00) + fx1 * (d11 - d01));
        if (live) {
          if (yt && xl) atomic_add4(grad_value + o00, aw * fy0 * fx0, go);
          if (yt && xr) atomic_add4(grad_value + o01, aw * fy0 * fx1, go);
          if (yb && xl) atomic_add4(grad_value + o10, aw * fy1 * fx0, go);
          if (yb && xr) atomic_add4(grad_value + o11, aw * fy1 * fx1, go);
        }
      }
      g_w = group_sum<M::LPG>(g_w);
      g_x = group_sum<M::LPG>(g_x);
      g_y = group_sum<M::LPG>(g_y);
      if (live && sub == 0) {
        grad_w[pair * (int64_t)(L * P) + l * P + p] = g_w;
        grad_loc[(pair * (int64_t)(L * P) + l * P + p) * 2 + 0] = g_x;
        grad_loc[(pair * (int64_t)(L * P) + l * P + p) * 2 + 1] = g_y;
      }
    }
  }
}

static int fill_levels(LevelInfo& lv, const int32_t* level_hw, int L, int S, const char* who) {
  if (L < 1 || L > WM2F_MAX_LEVELS) {
    set_error("%s: L=%d outside [1,%d]", who, L, WM2F_MAX_LEVELS);
    return WM2F_EINVAL;
  }
  int64_t start = 0;
  for (int l = 0; l < L; ++l) {
    lv.h[l] = level_hw[2 * l];
    lv.w[l] = level_hw[2 * l + 1];
    lv.start[l] = (int)start;
    if (lv.h[l] <= 0 || lv.w[l] <= 0) {
      set_error("%s: level %d has non-positive shape (%d,%d)", who, l, lv.h[l], lv.w[l]);
      return WM2F_EINVAL;
    }
    start += (int64_t)lv.h[l] * lv.w[l];
  }
  if (start != S) {
    set_error("%s: sum of level sizes %lld != S=%d", who, (long long)start, S);
    return WM2F_EINVAL;
  }
  return WM2F_OK;
}

// msdeform_tiled.hip
template <bool FUSED>
int launch_tiled(const void* value, const void* a, const void* b, void* out, const int32_t* level_hw, int B, int S,
                 int Q, int heads, int L, int P, int margin, int threads, void* stream, const char* who,
                 bool* handled, int mode, int a_qstride, int b_qstride);

// msdeform_quad.hip
template <bool FUSED>
int launch_quad(const void* value, const void* a, const void* b, void* out, const int32_t* level_hw, int B, int S, int Q,
                int heads, int L, int P, void* stream, const char* who, bool* handled, int mode, int a_qstride,
                int b_qstride);

template <bool FUSED>
int launch_stream(const void* value, const void* a, const void* b, void* out, const int32_t* level_hw, int B, int S,
                  int Q, int heads, int L, int P, void* stream, const char* who, bool* handled, int mode, int a_qstride,
                  int b_qstride, int lanes = 0);

// msdeform_tiled_bwd.hip
int launch_tiled_bwd(const void* value, const void* loc, const void* attn_w, const void* grad_out, void* grad_value,
                     void* grad_loc, void* grad_w, const int32_t* level_hw, int B, int S, int Q, int heads, int L, int P,
                     int margin, void* stream, const char* who, bool* handled, void* det_ws, int rows = 0);
int64_t tiled_bwd_det_workspace(const int32_t* level_hw, int B, int S, int heads, int L);

struct LaunchGeom {
  int64_t n_pairs;
  int n_blocks, blocks_per_xcd, grid;
};

static LaunchGeom geom(int B, int Q, int heads, int D) {
  LaunchGeom g;
  g.n_pairs = (int64_t)B * Q * heads;
  const int ppw = kWave / (D / 4);
  const int64_t n_waves = ceil_div64(g.n_pairs, ppw);
  g.n_blocks = (int)ceil_div64(n_waves, 4);
  g.blocks_per_xcd = ceil_div(g.n_blocks, kNumXcd);
  g.grid = g.blocks_per_xcd * kNumXcd;
  return g;
}

// variant (production library): 0 = auto (streaming quad kernel, else LDS-window kernel, else direct gather), 1 = direct
//          gather only, 2 = LDS-window kernel only, 4 = streaming quad kernel only -- the three kernels the library runs,
//          selectable so that the two fall-backs can be held to the golden vectors on shapes `auto` gives to the first.
// Profiling build only (libwm2f_prof.so, include/wm2f_prof.h) -- measured negatives, ablations, stamped builds:
//          3 = phased quad kernel (superseded by the streaming form), 13/23/43 its ablations, 73 stamped;
//          12/22/32/42/52 LDS-window ablations, 62 = LDS windows in slab-major work order;
//          5 = streaming kernel with per-window flags instead of workgroup barriers, 6 = tiles in 2-wide vertical strips,
//          7 = round-1 loader schedule, 8 = half-head form (two 77-KiB workgroups per CU), 44 without LDS reads,
//          74 / 84 stamped full-head / half-head builds
template <bool FUSED>
static int launch_fwd(const void* value, const void* a, const void* b, const void* ref, void* out,
                      const int32_t* level_hw, int B, int S, int Q, int heads, int D, int L, int P, int dtype,
                      void* stream, const char* who, int variant = 0, int margin = 4) {
  WM2F_REQUIRE(dtype == WM2F_F32, "%s: only WM2F_F32 is built", who);
  WM2F_REQUIRE(value && a && b && out && level_hw && (!FUSED || ref), "%s: null pointer", who);
  WM2F_REQUIRE(B > 0 && S > 0 && Q > 0 && heads > 0 && P > 0, "%s: non-positive size", who);
  WM2F_REQUIRE(D == 8 || D == 16 || D == 32 || D == 64, "%s: head_dim %d not in {8,16,32,64}", who, D);
  LevelInfo lv;
  if (int rc = fill_levels(lv, level_hw, L, S, who)) return rc;
#ifndef WM2F_PROFILING
  if (variant != 0 && variant != 1 && variant != 2 && variant != 4) {
    set_error("%s: variant %d is a superseded kernel, a timing ablation or a stamped build: profiling library only "
              "(libwm2f_prof.so, include/wm2f_prof.h)", who, variant);
    return WM2F_EUNSUPPORTED;
  }
#endif
  if (D == 32 && margin == 4 && (variant == 0 || variant % 10 == 4 || (variant >= 5 && variant <= 8))) {
    bool handled = false;
    const int smode = variant == 5 ? 100 : variant == 6 ? 200 : variant == 7 ? 300 : variant == 8 ? 400 : variant == 84 ? 74 : variant / 10;
    if (int rc = launch_stream<FUSED>(value, a, b, out, level_hw, B, S, Q, heads, L, P, stream, who, &handled, smode, 0, 0))
      return rc;
    if (handled) return WM2F_OK;
    if (variant != 0) {
      set_error("%s: the streaming quad kernel needs D=32, P=4, Q==S and 3 levels with sides 1:2:4, coarse first", who);
      return WM2F_EUNSUPPORTED;
    }
  }
#ifdef WM2F_PROFILING
  if (D == 32 && margin == 4 && variant % 10 == 3) {
    bool handled = false;
    if (int rc = launch_quad<FUSED>(value, a, b, out, level_hw, B, S, Q, heads, L, P, stream, who, &handled,
                                    variant / 10, 0, 0))
      return rc;
    if (handled) return WM2F_OK;
    set_error("%s: the phased quad kernel needs D=32, P=4, Q==S and 3 levels with sides 1:2:4, coarse first", who);
    return WM2F_EUNSUPPORTED;
  }
#endif
  if (variant != 1 && D == 32) {
    bool handled = false;
    if (int rc = launch_tiled<FUSED>(value, a, b, out, level_hw, B, S, Q, heads, L, P, margin,
                                     512, stream, who, &handled,
                                     variant >= 12 && variant % 10 == 2 ? variant / 10 : 0, 0, 0))
      return rc;
    if (handled) return WM2F_OK;
  }
  if (variant >= 2) {
    set_error("%s: the LDS-window kernel does not apply to this shape (needs D=32, P=4, Q==S, L<=4)", who);
    return WM2F_EUNSUPPORTED;
  }
  const LaunchGeom g = geom(B, Q, heads, D);
  hipStream_t st = (hipStream_t)stream;
#define WM2F_LAUNCH_FWD(DD, PC)                                                                       \
  hipLaunchKernelGGL((msdeform_fwd_kernel<DD, FUSED, PC>), dim3(g.grid), dim3(256), 0, st,            \
                     (const float*)value, (const float*)a, (const float*)b, (const float*)ref, (float*)out, \
                     lv, S, Q, heads, L, P, g.n_pairs, g.blocks_per_xcd, g.n_blocks)
#define WM2F_DISPATCH_P(DD) \
  if (P == 4) WM2F_LAUNCH_FWD(DD, 4); else WM2F_LAUNCH_FWD(DD, 0)
  switch (D) {
    case 8: WM2F_DISPATCH_P(8); break;
    case 16: WM2F_DISPATCH_P(16); break;
    case 32: WM2F_DISPATCH_P(32); break;
    default: WM2F_DISPATCH_P(64); break;
  }
#undef WM2F_DISPATCH_P
#undef WM2F_LAUNCH_FWD
  WM2F_CHECK_LAUNCH(who);
  return WM2F_OK;
}

}  // namespace wm2f

using namespace wm2f;

extern "C" int wm2f_msdeform_fwd(const void* value, const void* loc, const void* attn_w, void* out,
                                 const int32_t* level_hw, int B, int S, int Q, int heads, int D, int L, int P,
                                 int dtype, void* stream) {
  return launch_fwd<false>(value, loc, attn_w, nullptr, out, level_hw, B, S, Q, heads, D, L, P, dtype, stream,
                           "wm2f_msdeform_fwd");
}

extern "C" int wm2f_msdeform_fused_fwd(const void* value, const void* offsets, const void* logits, const void* ref,
                                       void* out, const int32_t* level_hw, int B, int S, int Q, int heads, int D,
                                       int L, int P, int dtype, void* stream) {
  return launch_fwd<true>(value, offsets, logits, ref, out, level_hw, B, S, Q, heads, D, L, P, dtype, stream,
                          "wm2f_msdeform_fused_fwd");
}

extern "C" int wm2f_msdeform_fwd_v(const void* value, const void* a, const void* b, const void* ref, void* out,
                                   const int32_t* level_hw, int B, int S, int Q, int heads, int D, int L, int P,
                                   int dtype, int fused, int variant, int margin, void* stream) {
  if (fused)
    return launch_fwd<true>(value, a, b, ref, out, level_hw, B, S, Q, heads, D, L, P, dtype, stream,
                            "wm2f_msdeform_fwd_v", variant, margin);
  return launch_fwd<false>(value, a, b, nullptr, out, level_hw, B, S, Q, heads, D, L, P, dtype, stream,
                           "wm2f_msdeform_fwd_v", variant, margin);
}

extern "C" int wm2f_msdeform_fused_packed_fwd(const void* value, const void* packed, void* out,
                                              const int32_t* level_hw, int B, int S, int Q, int heads, int D, int L,
                                              int P, int dtype, int margin, void* stream) {
  const char* who = "wm2f_msdeform_fused_packed_fwd";
  WM2F_REQUIRE(dtype == WM2F_F32, "%s: only WM2F_F32 is built", who);
  WM2F_REQUIRE(value && packed && out && level_hw, "%s: null pointer", who);
  WM2F_REQUIRE(B > 0 && S > 0 && Q > 0 && heads > 0 && P > 0, "%s: non-positive size", who);
  LevelInfo lv;
  if (int rc = fill_levels(lv, level_hw, L, S, who)) return rc;
  const int row = heads * L * P * 3;  // [offsets heads*L*P*2 | logits heads*L*P]
  bool handled = false;
  if (D == 32 && margin == 4) {
    const float* a = (const float*)packed;
    if (int rc = launch_stream<true>(value, a, a + heads * L * P * 2, out, level_hw, B, S, Q, heads, L, P, stream, who,
                                     &handled, 0, row, row))
      return rc;
  }
  if (D == 32 && !handled) {
    const float* a = (const float*)packed;
    if (int rc = launch_tiled<true>(value, a, a + heads * L * P * 2, out, level_hw, B, S, Q, heads, L, P, margin, 512,
                                    stream, who, &handled, 0, row, row))
      return rc;
  }
  if (!handled) {
    set_error("%s: needs D=32, P=4, Q==S, L<=4 and windows that fit LDS; use wm2f_msdeform_fused_fwd", who);
    return WM2F_EUNSUPPORTED;
  }
  return WM2F_OK;
}

extern "C" int wm2f_msdeform_fused_lanes_fwd(const void* value, const void* lanes, void* out, const int32_t* level_hw,
                                             int B, int S, int Q, int heads, int D, int L, int P, int dtype, int head_major,
                                             void* stream) {
  const char* who = "wm2f_msdeform_fused_lanes_fwd";
  WM2F_REQUIRE(dtype == WM2F_F32, "%s: only WM2F_F32 is built", who);
  WM2F_REQUIRE(value && lanes && out && level_hw, "%s: null pointer", who);
  WM2F_REQUIRE(B > 0 && S > 0 && Q > 0 && heads > 0 && P > 0, "%s: non-positive size", who);
  LevelInfo lv;
  if (int rc = fill_levels(lv, level_hw, L, S, who)) return rc;
  bool handled = false;
  if (D == 32 && L == 3 && P == 4) {
    // floats between consecutive tokens, and between the heads of one token: (B, Q, heads, 36), or head-major (heads, B, Q, 36)
    const bool rows_hm = (head_major & 1) != 0, value_hm = (head_major & 2) != 0;
    const int row = rows_hm ? L * P * 3 : heads * L * P * 3;
    const int head_stride = rows_hm ? B * Q * L * P * 3 : L * P * 3;
    int smode = 0;
#ifdef WM2F_PROFILING
    if (const char* e = getenv("WM2F_K1_STAMP")) smode = atoi(e) ? 7 : 0;  // profiling build: the stamped kernel on the lane-major rows
    if (const char* e = getenv("WM2F_K1_MODE")) smode = atoi(e);          // profiling build: 200 strip order, 300 round-1 loader schedule
#endif
    if (int rc = launch_stream<true>(value, lanes, lanes, out, level_hw, B, S, Q, heads, L, P, stream, who, &handled, smode, row,
                                     head_stride, 1 | (value_hm ? 2 : 0) | (head_major & 4)))
      return rc;
  }
  if (!handled) {
    set_error("%s: the lane-major form exists for the streaming kernel only (D = 32, P = 4, Q == S, 3 levels with sides "
              "1:2:4 coarse first): use wm2f_msdeform_fused_packed_fwd with the [offsets | logits] rows", who);
    return WM2F_EUNSUPPORTED;
  }
  return WM2F_OK;
}

// K1 for training, on the merged projection's rows: forward and backward of
//     out = multi_scale_deformable_attention(value, shapes, ref + offsets / (W, H), softmax(logits))        HF:983-1002, :798-837
// as ONE op of (value, rows), rows = (B, Q, heads * L * P * 3) = [offsets | logits] per token, so that the prologue's
// elementwise passes (divide, add, softmax, casts) and their backward (and autograd's zero-fill + add of the two row slices)
// do not exist.  dtype = WM2F_F32: rows, out, grad_out, grad_rows fp32; WM2F_BF16: all four bf16 (what a bf16-autocast Linear
// writes and reads).  value and grad_value are fp32 either way (the arithmetic is fp32, as the dependency's is under autocast:
// grid_sample is on its fp32 list).  Reference points are those of HF:1127-1156 with valid ratios of 1 (pixel centres).
// WM2F_EUNSUPPORTED where the streaming / LDS-window kernels do not apply: the caller composes the op from wm2f_msdeform_fwd.
extern "C" int wm2f_msdeform_rows_fwd(const void* value, const void* rows, void* out, const int32_t* level_hw, int B, int S, int Q,
                                      int heads, int D, int L, int P, int dtype, void* stream) {
  const char* who = "wm2f_msdeform_rows_fwd";
  WM2F_REQUIRE(dtype == WM2F_F32 || dtype == WM2F_BF16, "%s: dtype", who);
  WM2F_REQUIRE(value && rows && out && level_hw, "%s: null pointer", who);
  WM2F_REQUIRE(B > 0 && S > 0 && Q > 0 && heads > 0 && P > 0, "%s: non-positive size", who);
  LevelInfo lv;
  if (int rc = fill_levels(lv, level_hw, L, S, who)) return rc;
  bool handled = false;
  if (D == 32 && L == 3 && P == 4 && heads % 2 == 0) {
    const int row = heads * L * P * 3, n_off = heads * L * P * 2;
    const void* logits = dtype == WM2F_BF16 ? (const void*)((const unsigned short*)rows + n_off) : (const void*)((const float*)rows + n_off);
    if (int rc = launch_stream<true>(value, rows, logits, out, level_hw, B, S, Q, heads, L, P, stream, who, &handled, 0, row, row,
                                     dtype == WM2F_BF16 ? 24 : 0))
      return rc;
  }
  if (!handled) {
    set_error("%s: needs D = 32, P = 4, Q == S, an even head count and 3 levels the streaming kernel takes", who);
    return WM2F_EUNSUPPORTED;
  }
  return WM2F_OK;
}

// grad_value (fp32; need NOT be cleared: the first kernel clears it, the second accumulates), grad_rows (every element written).
extern "C" int wm2f_msdeform_rows_bwd(const void* value, const void* rows, const void* grad_out, void* grad_value, void* grad_rows,
                                      const int32_t* level_hw, int B, int S, int Q, int heads, int D, int L, int P, int dtype,
                                      void* stream) {
  const char* who = "wm2f_msdeform_rows_bwd";
  WM2F_REQUIRE(dtype == WM2F_F32 || dtype == WM2F_BF16, "%s: dtype", who);
  WM2F_REQUIRE(value && rows && grad_out && grad_value && grad_rows && level_hw, "%s: null pointer", who);
  WM2F_REQUIRE(B > 0 && S > 0 && Q > 0 && heads > 0 && P > 0, "%s: non-positive size", who);
  LevelInfo lv;
  if (int rc = fill_levels(lv, level_hw, L, S, who)) return rc;
  bool handled = false;
  if (D == 32)
    if (int rc = launch_tiled_bwd(value, rows, nullptr, grad_out, grad_value, grad_rows, nullptr, level_hw, B, S, Q, heads, L, P, 4,
                                  stream, who, &handled, nullptr, dtype == WM2F_BF16 ? 2 : 1))
      return rc;
  if (!handled) {
    set_error("%s: needs D = 32, P = 4, Q == S, an even head count and 3 levels whose windows fit LDS", who);
    return WM2F_EUNSUPPORTED;
  }
  return WM2F_OK;
}

extern "C" int wm2f_msdeform_bwd(const void* value, const void* loc, const void* attn_w, const void* grad_out,
                                 void* grad_value, void* grad_loc, void* grad_attn_w, const int32_t* level_hw,
                                 int B, int S, int Q, int heads, int D, int L, int P, int dtype, void* stream) {
  const char* who = "wm2f_msdeform_bwd";
  WM2F_REQUIRE(dtype == WM2F_F32, "%s: only WM2F_F32 is built", who);
  WM2F_REQUIRE(value && loc && attn_w && grad_out && grad_value && grad_loc && grad_attn_w && level_hw,
               "%s: null pointer", who);
  WM2F_REQUIRE(B > 0 && S > 0 && Q > 0 && heads > 0 && P > 0, "%s: non-positive size", who);
  WM2F_REQUIRE(D == 8 || D == 16 || D == 32 || D == 64, "%s: head_dim %d not in {8,16,32,64}", who, D);
  LevelInfo lv;
  if (int rc = fill_levels(lv, level_hw, L, S, who)) return rc;
  if (D == 32) {  // LDS-window backward: the scatter is absorbed on chip
    bool handled = false;
    if (int rc = launch_tiled_bwd(value, loc, attn_w, grad_out, grad_value, grad_loc, grad_attn_w, level_hw, B, S, Q,
                                  heads, L, P, 4, stream, who, &handled, nullptr))
      return rc;
    if (handled) return WM2F_OK;
  }
  const LaunchGeom g = geom(B, Q, heads, D);
  hipStream_t st = (hipStream_t)stream;
#define WM2F_LAUNCH_BWD(DD)                                                                                  \
  hipLaunchKernelGGL((msdeform_bwd_kernel<DD>), dim3(g.grid), dim3(256), 0, st, (const float*)value,          \
                     (const float*)loc, (const float*)attn_w, (const float*)grad_out, (float*)grad_value,    \
                     (float*)grad_loc, (float*)grad_attn_w, lv, S, Q, heads, L, P, g.n_pairs, g.blocks_per_xcd, \
                     g.n_blocks)
  switch (D) {
    case 8: WM2F_LAUNCH_BWD(8); break;
    case 16: WM2F_LAUNCH_BWD(16); break;
    case 32: WM2F_LAUNCH_BWD(32); break;
    default: WM2F_LAUNCH_BWD(64); break;
  }
#undef WM2F_LAUNCH_BWD
  WM2F_CHECK_LAUNCH(who);
  return WM2F_OK;
}

extern "C" int64_t wm2f_msdeform_bwd_det_workspace(const int32_t* level_hw, int B, int S, int heads, int D, int L) {
  if (!level_hw || B <= 0 || S <= 0 || heads <= 0 || D != 32) return 0;
  return tiled_bwd_det_workspace(level_hw, B, S, heads, L);
}

extern "C" int wm2f_msdeform_bwd_det(const void* value, const void* loc, const void* attn_w, const void* grad_out,
                                     void* grad_value, void* grad_loc, void* grad_attn_w, void* workspace,
                                     const int32_t* level_hw, int B, int S, int Q, int heads, int D, int L, int P, int dtype,
                                     void* stream) {
  const char* who = "wm2f_msdeform_bwd_det";
  WM2F_REQUIRE(dtype == WM2F_F32, "%s: only WM2F_F32 is built", who);
  WM2F_REQUIRE(value && loc && attn_w && grad_out && grad_value && grad_loc && grad_attn_w && level_hw && workspace,
               "%s: null pointer", who);
  WM2F_REQUIRE(B > 0 && S > 0 && Q > 0 && heads > 0 && P > 0, "%s: non-positive size", who);
  LevelInfo lv;
  if (int rc = fill_levels(lv, level_hw, L, S, who)) return rc;
  bool handled = false;
  if (D == 32 && S < (1 << 17)) {
    if (int rc = launch_tiled_bwd(value, loc, attn_w, grad_out, grad_value, grad_loc, grad_attn_w, level_hw, B, S, Q, heads, L,
                                  P, 4, stream, who, &handled, workspace))
      return rc;
  }
  if (!handled) {
    set_error("%s: only the LDS-window backward has the fixed-point form (head_dim 32, 4 points, self-attention Q == S, "
              "levels that fit its windows, S < 2^17)", who);
    return WM2F_EUNSUPPORTED;
  }
  return WM2F_OK;
}
