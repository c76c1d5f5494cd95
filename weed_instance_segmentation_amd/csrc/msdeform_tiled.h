// Shared geometry of the LDS-window deformable-attention kernels (forward: msdeform_tiled.hip,
// backward: msdeform_tiled_bwd.hip).  See msdeform_tiled.hip for the design.
#pragma once
#include "common.h"

namespace wm2f {


constexpr int kMaxLv = 4;

struct TileGeom {
  int fine;              // index of the finest level (defines the tile grid)
  int F, M;              // tile side in finest-level pixels, margin in pixels (every level)
  int tiles_x, tiles_y;
  int h[kMaxLv], w[kMaxLv], start[kMaxLv];
  int win_w[kMaxLv], win_h[kMaxLv];  // window size (same for every tile)
  int lds_off4[kMaxLv];              // window base in LDS, in float4 units (padded to 8 pixels)
  int lv_tab_off4;                   // 16-int level table behind the windows
  int order;                         // 0 = (image, head)-major ids, 1 = heads innermost
  int a_qstride, b_qstride;          // floats between consecutive queries in the two operand arrays
};

__device__ __forceinline__ int ceil_div_i(int a, int b) {  // b > 0, a may be negative
  return (a >= 0) ? (a + b - 1) / b : -((-a) / b);
}
__device__ __forceinline__ int floor_div_i(int a, int b) {  // b > 0
  return (a >= 0) ? a / b : -((-a + b - 1) / b);
}

// First query index (along one axis) of level size n whose reference point (q + 0.5) / n lies at or
// beyond t * F / nf:  q >= (2 t F n - nf) / (2 nf).
__device__ __forceinline__ int q_lo(int t, int F, int n, int nf) {
  int v = ceil_div_i(2 * t * F * n - nf, 2 * nf);
  return v < 0 ? 0 : (v > n ? n : v);
}

__device__ __forceinline__ float4 ld4g(const float* p) { return *reinterpret_cast<const float4*>(p); }

__device__ __forceinline__ void fma4s(float4& acc, float s, const float4& v) {
  acc.x = fmaf(s, v.x, acc.x);
  acc.y = fmaf(s, v.y, acc.y);
  acc.z = fmaf(s, v.z, acc.z);
  acc.w = fmaf(s, v.w, acc.w);
}

using f32x2 = __attribute__((ext_vector_type(2))) float;

// acc (4 channels as two packed pairs) += s * v   -> two v_pk_fma_f32
__device__ __forceinline__ void pk_fma4(f32x2& lo, f32x2& hi, float s, const float4& v) {
  const f32x2 ss = {s, s};
  lo = __builtin_elementwise_fma((f32x2){v.x, v.y}, ss, lo);
  hi = __builtin_elementwise_fma((f32x2){v.z, v.w}, ss, hi);
}


typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct TiledPlan {
  TileGeom g;
  size_t lds_bytes;
  bool ok;
};

// Host: choose the tile side F (16, 8, 4) so that every level's window fits the CU's 160 KiB of LDS beside the kernels' static arrays.
inline TiledPlan plan_tiled(const int32_t* level_hw, int L, int margin) {
  TiledPlan p{};
  p.ok = false;
  if (L < 1 || L > kMaxLv) return p;
  TileGeom& g = p.g;
  int64_t start = 0, best = -1;
  for (int l = 0; l < L; ++l) {
    g.h[l] = level_hw[2 * l];
    g.w[l] = level_hw[2 * l + 1];
    g.start[l] = (int)start;
    start += (int64_t)g.h[l] * g.w[l];
    if ((int64_t)g.h[l] * g.w[l] > best) {
      best = (int64_t)g.h[l] * g.w[l];
      g.fine = l;
    }
  }
  g.M = margin;
  const int Wf = g.w[g.fine], Hf = g.h[g.fine];
  for (int F : {16, 8, 4}) {
    int off4 = 0;
    for (int l = 0; l < L; ++l) {
      // window side: floor(x1 + M) + 1 - floor(x0 - M) + 1 <= ceil(F * Wl / Wf) + 2M + 2
      g.win_w[l] = (F * g.w[l] + Wf - 1) / Wf + 2 * margin + 2;
      g.win_h[l] = (F * g.h[l] + Hf - 1) / Hf + 2 * margin + 2;
      g.lds_off4[l] = off4;
      off4 += ((g.win_w[l] * g.win_h[l] + 7) / 8) * 64;  // whole 8-pixel (1 KiB) LDS-DMA pieces
    }
    g.lv_tab_off4 = off4;
    off4 += 2 * kMaxLv;  // 8-int-per-level table behind the windows
    if ((size_t)off4 * 16 <= 156 * 1024) {  // 160 KiB minus the kernels' static arrays (the backward's channel maxima: 2.3 KiB)
      g.F = F;
      g.tiles_x = (Wf + F - 1) / F;
      g.tiles_y = (Hf + F - 1) / F;
      p.lds_bytes = (size_t)off4 * 16;
      p.ok = true;
      return p;
    }
  }
  return p;
}

}  // namespace wm2f
