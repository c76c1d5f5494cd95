// K1, LDS-window variant: multi-scale deformable attention for head_dim 32.
// Same arithmetic as msdeform.hip (transformers modeling_mask2former.py:798-837, fused variant
// :983-1002); different memory strategy, chosen from measurement.
//
// Why: the direct-gather kernel is bound by the vector-memory path, not by HBM -- rocprofv3 PMC at
// config 2: TA busy 87 %, TCP->TCC read requests ~= every 128-B line a wave asks for (L1 gives no
// reuse: a (query, head) touches 6 KB, the 32-KB L1 is shared by up to 32 waves), 8.6 GB through a
// ~64 B/clk/CU pipe.  Neighbouring queries sample overlapping pixels, so the reuse has to be held
// explicitly: in LDS.
//
// Decomposition: one workgroup (8 waves) = (image, head, SPATIAL tile of FxF finest-level pixels).
// The tile owns every query of EVERY level whose reference point falls in it (F^2 + (F/2)^2 +
// (F/4)^2 queries for a 3-level pyramid), because they all sample the same three neighbourhoods.
//   1. stage: per level, the window [tile footprint +- margin] of this head's 128-B value slices
//      goes HBM/L2 -> LDS once (pixels outside the image are staged as zeros = zero padding);
//   2. gather: 4 lanes per query (8 channels each = two float4) read the 4 bilinear corners of each
//      sampling point from LDS (ds_read_b128), branch-free, with packed FMAs; a point whose 2x2
//      footprint is not inside the window is flagged and redone on a slow path (global loads, full
//      bounds checks), so ANY offsets stay correct -- the margin only decides how often the fast
//      path is taken.  4 rather than 8 lanes per query halves the redundant coordinate arithmetic,
//      which is what bounds this phase (ablation: 142 us of pure issue out of 244 us).
// LDS: (26^2 + 18^2 + 14^2) pixels x 128 B = 149.5 KiB at F = 16, margin 4 -> one workgroup per CU.
// XCD-aware launch: blocks sharing an XCD take a contiguous range of logical ids; ids run heads
// innermost, so the 8 heads of a tile run side by side on one XCD and share the location / weight
// lines (a head uses 96 + 48 B of each query's 768 + 384 B), and neighbouring tiles follow at once.
//
// Roofline: HBM, same algorithmic bytes as msdeform.hip (550 502 400 B per launch at config 2).
#include "msdeform_tiled.h"

namespace wm2f {

// Slow path for one point: per-corner image-bounds checks, corners from global memory.
__device__ __forceinline__ void point_slow(float4& acc, const float* __restrict__ vlev, int Hl, int Wl, int row_stride,
                                        float x, float y, float aw) {
  if (!(x > -1.f && x < (float)Wl && y > -1.f && y < (float)Hl)) return;
  const float x0f = floorf(x), y0f = floorf(y);
  const int x0 = (int)x0f, y0 = (int)y0f;
  const float fx1 = x - x0f, fy1 = y - y0f, fx0 = 1.f - fx1, fy0 = 1.f - fy1;
  const bool xl = x0 >= 0, xr = x0 + 1 < Wl, yt = y0 >= 0, yb = y0 + 1 < Hl;
  const float* p00 = vlev + (int64_t)(y0 * Wl + x0) * row_stride;
  if (yt && xl) fma4s(acc, aw * fy0 * fx0, ld4g(p00));
  if (yt && xr) fma4s(acc, aw * fy0 * fx1, ld4g(p00 + row_stride));
  if (yb && xl) fma4s(acc, aw * fy1 * fx0, ld4g(p00 + (int64_t)Wl * row_stride));
  if (yb && xr) fma4s(acc, aw * fy1 * fx1, ld4g(p00 + (int64_t)(Wl + 1) * row_stride));
}

__device__ const float4 g_zero_page[1] = {{0.f, 0.f, 0.f, 0.f}};  // LDS-DMA source for out-of-image pixels


// Everything a query needs before it can start sampling; loaded one query ahead of its use.
template <int NL, int P>
struct QueryOps {
  float4 lc[NL * P / 2];  // (x, y) of two points each: locations or raw offsets
  float4 wt[NL * P / 4];  // four weights / logits each
  int qxi, qyi, Wq, Hq;
  int64_t pair;
  bool valid;
};

// FUSED = false: a = loc (B,Q,heads,NL,P,2), b = attn_w (B,Q,heads,NL,P)
// FUSED = true : a = raw offsets, b = raw logits; reference points are recomputed from the query grid.
// MODE 0 = the kernel; 1 = staging only, 2 = gather only: timing ablations (outputs are NOT valid),
// reachable only through wm2f_msdeform_fwd_v variants 12 / 22.
template <int NL, int P, bool FUSED, int kTiledThreads, int MODE = 0>
__global__ __launch_bounds__(kTiledThreads) void msdeform_tiled_fwd_kernel(const float* __restrict__ value,
                                                                           const float* __restrict__ a_in,
                                                                           const float* __restrict__ b_in,
                                                                           float* __restrict__ out, TileGeom g, int S,
                                                                           int Q, int heads, int n_logical,
                                                                           int per_xcd) {
  constexpr int D = 32;
  constexpr int kLQ = 4;                      // lanes per query in the gather: 8 channels each
  constexpr int kSlots = kTiledThreads / kLQ;  // queries handled per pass
  constexpr int kWaves = kTiledThreads / kWave;
  extern __shared__ __attribute__((aligned(16))) float4 win[];  // windows, then a 16-int level table
  const int id = xcd_contiguous_id(blockIdx.x, per_xcd);
  if (id >= n_logical) return;
  // runtime-indexable per-level table: H, W, start, nqx, qx0, qy0, first query index, -
  int* lv_tab = reinterpret_cast<int*>(win + g.lv_tab_off4);
  const int n_tiles = g.tiles_x * g.tiles_y;
  int tile, h, b;
  if (g.order == 0) {  // (image, head)-major: consecutive ids = neighbouring tiles of one value slab
    tile = id % n_tiles;
    const int bh = id / n_tiles;
    h = bh % heads;
    b = bh / heads;
  } else {  // heads innermost: the 8 heads of a tile run together and share the loc / weight lines
    h = id % heads;
    const int bt = id / heads;
    tile = bt % n_tiles;
    b = bt / n_tiles;
  }
  const int ty = tile / g.tiles_x, tx = tile - ty * g.tiles_x;
  const int tid = threadIdx.x, j = tid & (kLQ - 1), slot = tid / kLQ;  // gather: lane j owns chunks j, j+4
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row_stride = heads * D;
  const float* vb = value + ((int64_t)b * S * heads + h) * D;  // head slice of token 0
  const int Wf = g.w[g.fine], Hf = g.h[g.fine];

  // ---- per-level geometry of this tile (wave-uniform)
  int wx0[NL], wy0[NL], qx0[NL], qy0[NL], nqx[NL], qcnt[NL + 1];
  qcnt[0] = 0;
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    const int Wl = g.w[l], Hl = g.h[l];
    // window origin: floor(t*F*Wl/Wf - 0.5 - M) = floor((2 t F Wl - Wf) / (2 Wf)) - M
    wx0[l] = floor_div_i(2 * tx * g.F * Wl - Wf, 2 * Wf) - g.M;
    wy0[l] = floor_div_i(2 * ty * g.F * Hl - Hf, 2 * Hf) - g.M;
    qx0[l] = q_lo(tx, g.F, Wl, Wf);
    qy0[l] = q_lo(ty, g.F, Hl, Hf);
    nqx[l] = q_lo(tx + 1, g.F, Wl, Wf) - qx0[l];
    const int nqy = q_lo(ty + 1, g.F, Hl, Hf) - qy0[l];
    qcnt[l + 1] = qcnt[l] + nqx[l] * nqy;
  }
  const int nq = qcnt[NL];
  if (threadIdx.x == 0) {
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      lv_tab[l * 8 + 0] = g.h[l];
      lv_tab[l * 8 + 1] = g.w[l];
      lv_tab[l * 8 + 2] = g.start[l];
      lv_tab[l * 8 + 3] = nqx[l] < 1 ? 1 : nqx[l];
      lv_tab[l * 8 + 4] = qx0[l];
      lv_tab[l * 8 + 5] = qy0[l];
      lv_tab[l * 8 + 6] = qcnt[l];
      lv_tab[l * 8 + 7] = __float_as_int(1.f / (float)(nqx[l] < 1 ? 1 : nqx[l]));
    }
  }

  auto load_ops = [&](int qi) __attribute__((always_inline)) {
    QueryOps<NL, P> o;
    o.valid = qi < nq;
    if (!o.valid) qi = 0;  // nq >= 1 whenever this is used for real work; keeps the loads in range
    int lq = 0;
#pragma unroll
    for (int l = 1; l < NL; ++l) lq += (qi >= qcnt[l]) ? 1 : 0;
    const int4 ta = *reinterpret_cast<const int4*>(lv_tab + lq * 8), tb = *reinterpret_cast<const int4*>(lv_tab + lq * 8 + 4);
    const int Hq = ta.x, Wq = ta.y, st = ta.z, nx = ta.w, ox = tb.x, oy = tb.y, loc_i = qi - tb.z;
    const int ly_ = (int)(((float)loc_i + 0.5f) * __int_as_float(tb.w));  // exact: small integers
    const int lx_ = loc_i - ly_ * nx;
    o.qxi = ox + lx_;
    o.qyi = oy + ly_;
    o.Wq = Wq;
    o.Hq = Hq;
    int q = st + o.qyi * Wq + o.qxi;
    if (q > Q - 1) q = Q - 1;
    o.pair = ((int64_t)b * Q + q) * heads + h;
    const float* ap = a_in + ((int64_t)b * Q + q) * g.a_qstride + h * (NL * P * 2);
    const float* bp = b_in + ((int64_t)b * Q + q) * g.b_qstride + h * (NL * P);
    if (MODE == 3 || MODE == 5) {  // ablation: synthetic operands, no global loads
#pragma unroll
      for (int i = 0; i < NL * P / 2; ++i) o.lc[i] = make_float4(0.3f * i - 1.f, 0.7f - 0.2f * i, 0.1f * (qi & 15) - 2.f, 1.5f);
#pragma unroll
      for (int i = 0; i < NL * P / 4; ++i) o.wt[i] = make_float4(0.1f, 0.2f * i, 0.3f, 0.05f * j);
      return o;
    }
#pragma unroll
    for (int i = 0; i < NL * P / 2; ++i) o.lc[i] = ld4g(ap + 4 * i);
#pragma unroll
    for (int i = 0; i < NL * P / 4; ++i) o.wt[i] = ld4g(bp + 4 * i);
    return o;
  };

  __syncthreads();  // lv_tab visible
  // operands of this thread's first query: in flight while the windows are staged
  QueryOps<NL, P> nxt = load_ops(slot);

  // ---- 1. stage the windows with LDS-DMA: a wave-instruction moves 8 pixels x 128 B, lane-linear
  // in LDS; every load of the workgroup is in flight before the single wait.
#pragma unroll
  for (int l = 0; l < (MODE >= 2 ? 0 : NL); ++l) {
    const int Wl = g.w[l], Hl = g.h[l], ww = g.win_w[l];
    const int npix = ww * g.win_h[l];
    const int n_chunks = (npix + 7) >> 3;  // 8-pixel pieces; the window allocation is padded to that
    const float inv_ww = 1.f / (float)ww;
    const float* vlev = vb + (int64_t)g.start[l] * row_stride;
    for (int c = wave; c < n_chunks; c += kWaves) {
      const int idx = c * 8 + (lane >> 3);
      const int wy = (int)(((float)idx + 0.5f) * inv_ww);
      const int wx = idx - wy * ww;
      const int x = wx0[l] + wx, y = wy0[l] + wy;
      const bool in = idx < npix && x >= 0 && x < Wl && y >= 0 && y < Hl;
      const float* src = in ? vlev + (int64_t)(y * Wl + x) * row_stride + (lane & 7) * 4
                            : reinterpret_cast<const float*>(g_zero_page);
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(win + g.lds_off4[l] + c * 64), 16, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  if (MODE == 1) {  // ablation: keep the staged data observable, skip the gather
    if (tid == 0) out[(int64_t)id] = win[id & 1023].x;
    return;
  }
  // ---- 2. gather: 8 lanes per query, all levels' queries of the tile
  for (int qi = slot; qi < nq; qi += kSlots) {
    const QueryOps<NL, P> cur = nxt;
    nxt = load_ops(qi + kSlots);  // prefetch the next query's operands under this query's sampling
    const int64_t pair = cur.pair;
    const float* ap = a_in + (pair / heads) * g.a_qstride + h * (NL * P * 2);
    const float* bp = b_in + (pair / heads) * g.b_qstride + h * (NL * P);

    float wts[NL * P];
#pragma unroll
    for (int i = 0; i < NL * P / 4; ++i) {
      wts[4 * i] = cur.wt[i].x; wts[4 * i + 1] = cur.wt[i].y; wts[4 * i + 2] = cur.wt[i].z; wts[4 * i + 3] = cur.wt[i].w;
    }
    float refx = 0.f, refy = 0.f, sm_max = 0.f, sm_inv = 1.f;
    if (FUSED) {  // softmax over the NL*P logits (HF:986-991) and the reference point (HF:1127-1156)
      float mx = wts[0];
#pragma unroll
      for (int i = 1; i < NL * P; ++i) mx = fmaxf(mx, wts[i]);
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < NL * P; ++i) {
        wts[i] = __expf(wts[i] - mx);
        s += wts[i];
      }
      const float inv = __builtin_amdgcn_rcpf(s);
      sm_max = mx;
      sm_inv = inv;
#pragma unroll
      for (int i = 0; i < NL * P; ++i) wts[i] *= inv;
      refx = ((float)cur.qxi + 0.5f) * __builtin_amdgcn_rcpf((float)cur.Wq);
      refy = ((float)cur.qyi + 0.5f) * __builtin_amdgcn_rcpf((float)cur.Hq);
    }

    // Branch-free fast path: every point issues its 4 LDS reads unconditionally (address clamped to
    // the window origin and weight zeroed when the footprint is not inside the window), so the 12
    // points form ONE basic block and the LDS reads pipeline instead of stalling point by point.
    f32x2 acc_lo = {0.f, 0.f}, acc_hi = {0.f, 0.f};    // channels 4j .. 4j+3
    f32x2 acc2_lo = {0.f, 0.f}, acc2_hi = {0.f, 0.f};  // channels 16+4j .. 16+4j+3
    unsigned slow = 0;  // bit (l*P + p): the point's 2x2 footprint is not inside the staged window
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      const int Wl = g.w[l], Hl = g.h[l], ww = g.win_w[l], wh = g.win_h[l];
      const float4* wl = win + g.lds_off4[l] + j;
      const float bx = refx * (float)Wl - 0.5f, by = refy * (float)Hl - 0.5f;
#pragma unroll
      for (int p = 0; p < P; p += 2) {
        const float4 lc = cur.lc[(l * P + p) / 2];  // two points: (x, y, x, y)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float lx = e ? lc.z : lc.x, ly = e ? lc.w : lc.y;
          float x, y;
          if (FUSED) {  // loc = ref + off / (W, H); pixel = loc * (W, H) - 0.5  ==  ref*W - 0.5 + off
            x = bx + lx;
            y = by + ly;
          } else {  // grid_sample's own arithmetic (align_corners = False)
            x = ((2.f * lx - 1.f + 1.f) * (float)Wl - 1.f) * 0.5f;
            y = ((2.f * ly - 1.f + 1.f) * (float)Hl - 1.f) * 0.5f;
          }
          const float x0f = floorf(x), y0f = floorf(y);
          const int xr = (int)x0f - wx0[l], yr = (int)y0f - wy0[l];
          const bool fast = ((unsigned)xr < (unsigned)(ww - 1)) & ((unsigned)yr < (unsigned)(wh - 1));
          const int idx = fast ? yr * ww + xr : 0;
          const float aw = fast ? wts[l * P + p + e] : 0.f;
          slow |= fast ? 0u : (1u << (l * P + p + e));
          const float fx1 = x - x0f, fy1 = y - y0f;
          const float a1 = aw * fy1, a0 = aw - a1;
          const float w01 = a0 * fx1, w00 = a0 - w01, w11 = a1 * fx1, w10 = a1 - w11;
          const float4* c = wl + idx * 8;
          float4 v00, v01, v10, v11, u00, u01, u10, u11;
          if (MODE == 4 || MODE == 5) {  // ablation: no LDS reads
            v00 = make_float4(fx1, fy1, x, y); v01 = make_float4(y, x, fx1, 1.f); v10 = v00; v11 = v01;
            u00 = v01; u01 = v00; u10 = v01; u11 = v00;
            asm volatile("" :: "v"(c));
          } else {
            v00 = c[0]; v01 = c[8]; v10 = c[ww * 8]; v11 = c[ww * 8 + 8];
            u00 = c[4]; u01 = c[12]; u10 = c[ww * 8 + 4]; u11 = c[ww * 8 + 12];
          }
          pk_fma4(acc_lo, acc_hi, w00, v00);
          pk_fma4(acc_lo, acc_hi, w01, v01);
          pk_fma4(acc_lo, acc_hi, w10, v10);
          pk_fma4(acc_lo, acc_hi, w11, v11);
          pk_fma4(acc2_lo, acc2_hi, w00, u00);
          pk_fma4(acc2_lo, acc2_hi, w01, u01);
          pk_fma4(acc2_lo, acc2_hi, w10, u10);
          pk_fma4(acc2_lo, acc2_hi, w11, u11);
        }
      }
    }
    float4 acc = make_float4(acc_lo.x, acc_lo.y, acc_hi.x, acc_hi.y);
    float4 acc2 = make_float4(acc2_lo.x, acc2_lo.y, acc2_hi.x, acc2_hi.y);
    while (slow) {  // rare: ONE copy of the general code, operands re-derived from memory
      const int i = __ffs(slow) - 1;
      slow &= slow - 1;
      const int l = i / P;
      const int Hl = lv_tab[l * 8 + 0], Wl = lv_tab[l * 8 + 1], st_l = lv_tab[l * 8 + 2];
      const float lx = ap[i * 2], ly = ap[i * 2 + 1];
      float aw = bp[i], x, y;
      if (FUSED) {
        aw = __expf(aw - sm_max) * sm_inv;
        x = refx * (float)Wl - 0.5f + lx;
        y = refy * (float)Hl - 0.5f + ly;
      } else {
        x = ((2.f * lx - 1.f + 1.f) * (float)Wl - 1.f) * 0.5f;
        y = ((2.f * ly - 1.f + 1.f) * (float)Hl - 1.f) * 0.5f;
      }
      point_slow(acc, vb + (int64_t)st_l * row_stride + j * 4, Hl, Wl, row_stride, x, y, aw);
      point_slow(acc2, vb + (int64_t)st_l * row_stride + 16 + j * 4, Hl, Wl, row_stride, x, y, aw);
    }
    *reinterpret_cast<float4*>(out + pair * D + j * 4) = acc;
    *reinterpret_cast<float4*>(out + pair * D + 16 + j * 4) = acc2;
  }
}

// ------------------------------------------------------------------------------------ host side
template <bool FUSED>
int launch_tiled(const void* value, const void* a, const void* b, void* out, const int32_t* level_hw, int B, int S,
                 int Q, int heads, int L, int P, int margin, int threads, void* stream, const char* who,
                 bool* handled, int mode, int a_qstride, int b_qstride) {
  *handled = false;
  if (P != 4 || L < 1 || L > kMaxLv || margin < 0) return WM2F_OK;
  if ((int64_t)Q != S) return WM2F_OK;  // the tiling assumes queries == value tokens (encoder self-attention)
  TiledPlan p = plan_tiled(level_hw, L, margin);
  if (!p.ok) return WM2F_OK;
  p.g.order = (mode == 6) ? 0 : 1;
  p.g.a_qstride = a_qstride > 0 ? a_qstride : heads * L * P * 2;
  p.g.b_qstride = b_qstride > 0 ? b_qstride : heads * L * P;  // default: heads innermost (measured 4 % faster); mode 6 = slab-major
  const int64_t n_logical = (int64_t)B * heads * p.g.tiles_x * p.g.tiles_y;
  if (n_logical > (1 << 30)) return WM2F_OK;
  const int per_xcd = (int)ceil_div64(n_logical, kNumXcd);
  hipStream_t st = (hipStream_t)stream;
#ifdef WM2F_PROFILING  /* timing ablations (invalid outputs): profiling build only */
#define WM2F_TL_ABLATIONS(NLv)                                                          \
    if (NLv == 3 && mode == 1) kfn = msdeform_tiled_fwd_kernel<3, 4, FUSED, 512, 1>;    \
    if (NLv == 3 && mode == 2) kfn = msdeform_tiled_fwd_kernel<3, 4, FUSED, 512, 2>;    \
    if (NLv == 3 && mode == 3) kfn = msdeform_tiled_fwd_kernel<3, 4, FUSED, 512, 3>;    \
    if (NLv == 3 && mode == 4) kfn = msdeform_tiled_fwd_kernel<3, 4, FUSED, 512, 4>;    \
    if (NLv == 3 && mode == 5) kfn = msdeform_tiled_fwd_kernel<3, 4, FUSED, 512, 5>;
#else
#define WM2F_TL_ABLATIONS(NLv)
#endif
#define WM2F_TL(NLv)                                                                                              \
  case NLv: {                                                                                                     \
    auto kfn = msdeform_tiled_fwd_kernel<NLv, 4, FUSED, 512>;                                                     \
    WM2F_TL_ABLATIONS(NLv)                                                                                        \
    if (p.lds_bytes > 64 * 1024) {                                                                                \
      hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize,           \
                                         (int)p.lds_bytes);                                                       \
      if (e != hipSuccess) {                                                                                      \
        set_error("%s: cannot raise dynamic LDS to %zu: %s", who, p.lds_bytes, hipGetErrorString(e));            \
        return WM2F_ELAUNCH;                                                                                      \
      }                                                                                                           \
    }                                                                                                             \
    hipLaunchKernelGGL(kfn, dim3(per_xcd* kNumXcd), dim3(512), p.lds_bytes, st, (const float*)value,    \
                       (const float*)a, (const float*)b, (float*)out, p.g, S, Q, heads, (int)n_logical, per_xcd); \
  } break;
  switch (L) {
    WM2F_TL(1) WM2F_TL(2) WM2F_TL(3) WM2F_TL(4)
    default: return WM2F_OK;
  }
#undef WM2F_TL
#undef WM2F_TL_ABLATIONS
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: tiled launch failed: %s", who, hipGetErrorString(e));
    return WM2F_ELAUNCH;
  }
  *handled = true;
  return WM2F_OK;
}

// explicit instantiations used by msdeform.hip
template int launch_tiled<false>(const void*, const void*, const void*, void*, const int32_t*, int, int, int, int,
                                 int, int, int, int, void*, const char*, bool*, int, int, int);
template int launch_tiled<true>(const void*, const void*, const void*, void*, const int32_t*, int, int, int, int, int,
                                int, int, int, void*, const char*, bool*, int, int, int);

}  // namespace wm2f
