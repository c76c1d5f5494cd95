// K1 backward, LDS-window variant (head_dim 32).  Same tiling as msdeform_tiled.hip.
//
// The direct backward (msdeform.hip) scatters grad_value with global float atomics: 8.46 GB of added
// bytes per call against a chip-wide atomic rate of ~1.3 TB/s -- measured 25 ms per call, 150 ms of a
// 400 ms train step.  Here the scatter is absorbed on chip:
//
//   kernel A (grad_loc, grad_attn_w): stage the VALUE windows in LDS (as the forward does), then per
//     sampling point gather the four corners, dot them with the query's grad_out (8 channels per lane,
//     4 lanes per query, quad shuffles), and form the three scalar gradients.  No scatter.
//   kernel B (grad_value): the LDS window is an ACCUMULATOR: every (query, point, corner) adds
//     weight * grad_out with ds_add_f32; once per tile the window is flushed to grad_value with global
//     atomics shaped as whole 128-B rows (lane = channel), i.e. 0.63 GB per call instead of 8.46 GB.
// Points whose footprint leaves the window fall back to global loads / global atomics, so results do
// not depend on the margin.
//
// [round-1 history] With fp32 LDS atomics: kernel A 0.52 ms; kernel B 10.3 ms, ALL of it in the ds_add_f32 stream
// (ablation: without the flush 10.3 ms, without the LDS adds 0.2 ms): an LDS float atomic costs ~176
// cycles per wave-instruction (~2.7 cycles per lane) whether or not lanes conflict -- the wave-per-query
// form below (two contiguous 128-B rows per instruction) runs exactly as long as a scattered form did.
// Total 10.8 ms vs 25 ms for the global-atomic backward.
// Now: the window accumulates in FIXED POINT with integer LDS atomics (~5 cycles per wave-instruction, measured by
// tools/probes/lds_atomic_rate.hip; scaling and its overflow bound are at the accumulation site):
// kernel B 2.31 ms, kernel A 0.51 ms.
#include "msdeform_tiled.h"
#include <stdlib.h>
#include <type_traits>

namespace wm2f {

constexpr int kBwdThreads = 512;
constexpr int kQuadBwdThreads = 1024;  // the quad-form value kernel

__device__ const float4 g_zero_page_bwd[1] = {{0.f, 0.f, 0.f, 0.f}};  // LDS-DMA source for out-of-image pixels

template <int NL>
struct TileCtx {
  int tile, h, b;
  int wx0[NL], wy0[NL], nq;
};

// Decode the logical id, fill the per-level LDS table (H, W, start, nqx, qx0, qy0, first query, 1/nqx).
template <int NL>
__device__ __forceinline__ TileCtx<NL> tile_setup(const TileGeom& g, int id, int heads, int* lv_tab) {
  TileCtx<NL> c;
  const int n_tiles = g.tiles_x * g.tiles_y;
  c.h = id % heads;
  const int bt = id / heads;
  c.tile = bt % n_tiles;
  c.b = bt / n_tiles;
  const int ty = c.tile / g.tiles_x, tx = c.tile - ty * g.tiles_x;
  const int Wf = g.w[g.fine], Hf = g.h[g.fine];
  int qcnt = 0;
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    const int Wl = g.w[l], Hl = g.h[l];
    c.wx0[l] = floor_div_i(2 * tx * g.F * Wl - Wf, 2 * Wf) - g.M;
    c.wy0[l] = floor_div_i(2 * ty * g.F * Hl - Hf, 2 * Hf) - g.M;
    const int qx0 = q_lo(tx, g.F, Wl, Wf), qy0 = q_lo(ty, g.F, Hl, Hf);
    const int nqx = q_lo(tx + 1, g.F, Wl, Wf) - qx0, nqy = q_lo(ty + 1, g.F, Hl, Hf) - qy0;
    if (threadIdx.x == 0) {
      lv_tab[l * 8 + 0] = Hl;
      lv_tab[l * 8 + 1] = Wl;
      lv_tab[l * 8 + 2] = g.start[l];
      lv_tab[l * 8 + 3] = nqx < 1 ? 1 : nqx;
      lv_tab[l * 8 + 4] = qx0;
      lv_tab[l * 8 + 5] = qy0;
      lv_tab[l * 8 + 6] = qcnt;
      lv_tab[l * 8 + 7] = __float_as_int(1.f / (float)(nqx < 1 ? 1 : nqx));
    }
    qcnt += nqx * nqy;
  }
  c.nq = qcnt;
  return c;
}

// query index of slot qi inside the tile -> global query id (level table in LDS)
template <int NL>
__device__ __forceinline__ int tile_query(const int* lv_tab, int qi, int Q) {
  int lq = 0;
#pragma unroll
  for (int l = 1; l < NL; ++l) lq += (qi >= lv_tab[l * 8 + 6]) ? 1 : 0;
  const int4 ta = *reinterpret_cast<const int4*>(lv_tab + lq * 8), tb = *reinterpret_cast<const int4*>(lv_tab + lq * 8 + 4);
  const int loc_i = qi - tb.z;
  const int ly_ = (int)(((float)loc_i + 0.5f) * __int_as_float(tb.w));
  const int lx_ = loc_i - ly_ * ta.w;
  int q = ta.z + (tb.y + ly_) * ta.y + (tb.x + lx_);
  return q > Q - 1 ? Q - 1 : q;
}

// the same, plus where the query sits: its level and its (column, row) there (the reference point of the ROWS forms)
template <int NL>
__device__ __forceinline__ int tile_query_ex(const int* lv_tab, int qi, int Q, int& Wq, int& Hq, int& qx, int& qy) {
  int lq = 0;
#pragma unroll
  for (int l = 1; l < NL; ++l) lq += (qi >= lv_tab[l * 8 + 6]) ? 1 : 0;
  const int4 ta = *reinterpret_cast<const int4*>(lv_tab + lq * 8), tb = *reinterpret_cast<const int4*>(lv_tab + lq * 8 + 4);
  const int loc_i = qi - tb.z;
  const int ly_ = (int)(((float)loc_i + 0.5f) * __int_as_float(tb.w));
  const int lx_ = loc_i - ly_ * ta.w;
  Hq = ta.x;
  Wq = ta.y;
  qx = tb.x + lx_;
  qy = tb.y + ly_;
  int q = ta.z + qy * ta.y + qx;
  return q > Q - 1 ? Q - 1 : q;
}

__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
typedef __bf16 bf16x2b_t __attribute__((ext_vector_type(2)));
typedef float f32x2b_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16_b(float lo, float hi) {  // v_cvt_pk_bf16_f32, round to nearest even
  const f32x2b_t v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2b_t));
}

// ROWS forms (wm2f_msdeform_rows_bwd): the kernels read the merged projection's [offsets | logits] rows (B, Q, heads * NL * P * 3)
// -- fp32 (ROWS = 1) or bf16 (ROWS = 2: grad_out and the row gradients are bf16 as well) -- and redo the prologue of HF:983-1002
// in registers: softmax over the (query, head)'s NL * P logits, sampling pixel = reference pixel + offset (the reference point
// of a token is the centre of its own pixel, HF:1127-1156 with valid ratios of 1; in pixels of level l:
// (column + 0.5) * W_l / W_q - 0.5, exact for the 1 : 2 : 4 pyramids).  The 12 logits of one (query, head):
template <int ROWS, int NP>
__device__ __forceinline__ void load_logits(const void* rows, int64_t elem, float (&lg)[NP]) {
  static_assert(NP % 4 == 0, "whole 4-element pieces");
  if (ROWS == 2) {
    const uint2* p = reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(rows) + elem);
#pragma unroll
    for (int i = 0; i < NP / 4; ++i) {
      const uint2 u = p[i];
      lg[4 * i] = bf_lo(u.x); lg[4 * i + 1] = bf_hi(u.x); lg[4 * i + 2] = bf_lo(u.y); lg[4 * i + 3] = bf_hi(u.y);
    }
  } else {
    const float4* p = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(rows) + elem);
#pragma unroll
    for (int i = 0; i < NP / 4; ++i) {
      const float4 u = p[i];
      lg[4 * i] = u.x; lg[4 * i + 1] = u.y; lg[4 * i + 2] = u.z; lg[4 * i + 3] = u.w;
    }
  }
}

template <int NP>
__device__ __forceinline__ void softmax_regs(float (&lg)[NP]) {  // in place; HF:986-991
  float mx = lg[0];
#pragma unroll
  for (int i = 1; i < NP; ++i) mx = fmaxf(mx, lg[i]);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    lg[i] = __expf(lg[i] - mx);
    s += lg[i];
  }
  const float inv = __builtin_amdgcn_rcpf(s);  // as the forward kernel's prologue
#pragma unroll
  for (int i = 0; i < NP; ++i) lg[i] *= inv;
}

template <int ROWS>
__device__ __forceinline__ float2 load_offset(const void* rows, int64_t elem) {  // (x, y) of one point; elem even
  if (ROWS == 2) {
    const unsigned u = *reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned short*>(rows) + elem);
    return make_float2(bf_lo(u), bf_hi(u));
  }
  return *reinterpret_cast<const float2*>(reinterpret_cast<const float*>(rows) + elem);
}

__device__ __forceinline__ float dot8(const float4& a0, const float4& a1, const float4& b0, const float4& b1) {
  return a0.x * b0.x + a0.y * b0.y + a0.z * b0.z + a0.w * b0.w + a1.x * b1.x + a1.y * b1.y + a1.z * b1.z + a1.w * b1.w;
}

template <int K>
__device__ __forceinline__ int bcast_dpp(int v) {  // value of lane K of this lane's quad
  return __builtin_amdgcn_mov_dpp(v, K * 0x55, 0xF, 0xF, true);
}
__device__ __forceinline__ float quad_sum_dpp(float v) {  // same order of additions as quad_sum below: (v + xor 1) + xor 2
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
  return v + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_max_dpp(float v) {
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true)));
  return fmaxf(v, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true)));
}
// dot8 in the packed form the hardware has (v_pk_fma_f32: two products per instruction on register PAIRS as they were loaded):
// even and odd channels run as two partial sums, added at the end -- 4 packed FMAs + 1 add per corner.  (Left to the compiler,
// dot8's serial chain was packed ACROSS corners, with two v_mov per v_pk_fma to build the pairs: 112 instructions per point.)
__device__ __forceinline__ float dot8p(const f32x2 (&g)[4], const float4& a, const float4& b) {
  f32x2 acc = (f32x2){a.x, a.y} * g[0];
  acc = __builtin_elementwise_fma((f32x2){a.z, a.w}, g[1], acc);
  acc = __builtin_elementwise_fma((f32x2){b.x, b.y}, g[2], acc);
  acc = __builtin_elementwise_fma((f32x2){b.z, b.w}, g[3], acc);
  return acc.x + acc.y;
}
__device__ __forceinline__ float quad_sum(float v) {  // over the 4 lanes of a query
  v += __shfl_xor(v, 1, kWave);
  v += __shfl_xor(v, 2, kWave);
  return v;
}

__device__ __forceinline__ void pixel_coords(float lx, float ly, int Wl, int Hl, float& x, float& y) {
  x = ((2.f * lx - 1.f + 1.f) * (float)Wl - 1.f) * 0.5f;  // grid_sample, align_corners = False
  y = ((2.f * ly - 1.f + 1.f) * (float)Hl - 1.f) * 0.5f;
}

// ---------------------------------------------------------------------------------- kernel A
// ABL (profiling build, OUTPUTS NOT VALID): 1 = no window staging, 2 = staging only
template <int NL, int P, int ROWS = 0, int NT = kBwdThreads, int ABL = 0>
__global__ __launch_bounds__(NT) void msdeform_tiled_bwd_lw_kernel(
    const float* __restrict__ value, const float* __restrict__ loc, const float* __restrict__ attn_w,
    const float* __restrict__ grad_out, float* __restrict__ grad_loc, float* __restrict__ grad_w, TileGeom g, int S,
    int Q, int heads, int n_logical, int per_xcd) {
  constexpr int D = 32, kLQ = 4, kSlots = NT / kLQ, kWaves = NT / kWave;
  extern __shared__ __attribute__((aligned(16))) float4 win[];
  const int id = xcd_contiguous_id(blockIdx.x, per_xcd);
  if (id >= n_logical) return;
  int* lv_tab = reinterpret_cast<int*>(win + g.lv_tab_off4);
  const TileCtx<NL> c = tile_setup<NL>(g, id, heads, lv_tab);
  const int tid = threadIdx.x, j = tid & (kLQ - 1), slot = tid / kLQ;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row_stride = heads * D;
  const float* vb = value + ((int64_t)c.b * S * heads + c.h) * D;

  // stage the value windows (zero outside the image)
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    if (ABL == 1) break;
    const int Wl = g.w[l], Hl = g.h[l], ww = g.win_w[l];
    const int npix = ww * g.win_h[l], n_chunks = (npix + 7) >> 3;
    const float inv_ww = 1.f / (float)ww;
    const float* vlev = vb + (int64_t)g.start[l] * row_stride;
    for (int ch = wave; ch < n_chunks; ch += kWaves) {
      const int idx = ch * 8 + (lane >> 3);
      const int wy = (int)(((float)idx + 0.5f) * inv_ww), wx = idx - wy * ww;
      const int x = c.wx0[l] + wx, y = c.wy0[l] + wy;
      const bool in = idx < npix && x >= 0 && x < Wl && y >= 0 && y < Hl;
      const float* src = in ? vlev + (int64_t)(y * Wl + x) * row_stride + (lane & 7) * 4
                            : reinterpret_cast<const float*>(g_zero_page_bwd);
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(win + g.lds_off4[l] + ch * 64), 16, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (ABL == 2) return;

  // One quad per (query, head); lane j owns the 8 channels {4 j .. 4 j + 3, 16 + 4 j ..} AND, per level, point j: it works out
  // that point's pixel, weights and window address once (not four times), the quad then walks the level's four points --
  // address of point K by DPP broadcast, 8 window reads and 4 eight-channel dot products per lane, DPP sums over the quad,
  // kept by lane K -- and lane j finishes its own point's three gradients.  (The first form computed every point in all four
  // lanes and summed through ds_bpermute: ~200 vector instructions per point and lane, 975 of the kernel's 1015 us at
  // config 2; tools/probes/k1_rows_bench.py with WM2F_K1_LW_THREADS=1 / 2.)
  for (int qi = slot; qi < c.nq; qi += kSlots) {
    int Wq = 1, Hq = 1, qx = 0, qy = 0;
    const int q = ROWS ? tile_query_ex<NL>(lv_tab, qi, Q, Wq, Hq, qx, qy) : tile_query<NL>(lv_tab, qi, Q);
    const int64_t pair = ((int64_t)c.b * Q + q) * heads + c.h;
    float4 go0, go1;
    if (ROWS == 2) {
      const unsigned short* gp = reinterpret_cast<const unsigned short*>(grad_out) + pair * D + j * 4;
      const uint2 a = *reinterpret_cast<const uint2*>(gp), b2 = *reinterpret_cast<const uint2*>(gp + 16);
      go0 = make_float4(bf_lo(a.x), bf_hi(a.x), bf_lo(a.y), bf_hi(a.y));
      go1 = make_float4(bf_lo(b2.x), bf_hi(b2.x), bf_lo(b2.y), bf_hi(b2.y));
    } else {
      go0 = ld4g(grad_out + pair * D + j * 4);
      go1 = ld4g(grad_out + pair * D + 16 + j * 4);
    }
    const f32x2 gp[4] = {{go0.x, go0.y}, {go0.z, go0.w}, {go1.x, go1.y}, {go1.z, go1.w}};
    if (ROWS && grad_w) {
      // ROWS: `grad_w` carries grad_value, which the NEXT kernel accumulates into with atomics: this kernel, whose tiles' queries
      // partition the tokens, clears each (token, head) row on the way -- the 352-MB fill in front of the backward is gone
      *reinterpret_cast<float4*>(grad_w + pair * D + j * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(grad_w + pair * D + 16 + j * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // ROWS: this (query, head)'s piece of the token's row: offsets at off_e, logits at log_e (element indices)
    const int64_t row_e = ((int64_t)c.b * Q + q) * (heads * NL * P * 3);
    const int64_t off_e = row_e + c.h * (NL * P * 2), log_e = row_e + heads * (NL * P * 2) + c.h * (NL * P);
    // this lane's points (level l, point j): attention weight, (ROWS) raw offset
    float awl[NL];
    float2 ofl[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      if (ROWS == 2) {
        awl[l] = __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(loc)[log_e + l * P + j] << 16);
        ofl[l] = load_offset<2>(loc, off_e + (l * P + j) * 2);
      } else if (ROWS == 1) {
        awl[l] = loc[log_e + l * P + j];
        ofl[l] = load_offset<1>(loc, off_e + (l * P + j) * 2);
      } else {
        awl[l] = attn_w[pair * (NL * P) + l * P + j];
        ofl[l] = *reinterpret_cast<const float2*>(loc + (pair * (NL * P) + l * P + j) * 2);
      }
    }
    if (ROWS) {  // softmax over the (query, head)'s NL * P logits: NL per lane, the quad holds them all (HF:986-991)
      float mx = awl[0];
#pragma unroll
      for (int l = 1; l < NL; ++l) mx = fmaxf(mx, awl[l]);
      mx = quad_max_dpp(mx);
      float sm = 0.f;
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        awl[l] = __expf(awl[l] - mx);
        sm += awl[l];
      }
      const float inv = __builtin_amdgcn_rcpf(quad_sum_dpp(sm));  // as the forward kernel's prologue
#pragma unroll
      for (int l = 0; l < NL; ++l) awl[l] *= inv;
    }
    float gwl[NL], gxl[NL], gyl[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      const int Wl = g.w[l], Hl = g.h[l], ww = g.win_w[l], wh = g.win_h[l];
      const float4* wl = win + g.lds_off4[l] + j;
      const float* vlev = vb + (int64_t)g.start[l] * row_stride + j * 4;
      float x, y;
      if (ROWS) {  // the query's reference pixel at this level (see load_logits) + the offset
        x = (((float)qx + 0.5f) * ((float)Wl / (float)Wq) - 0.5f) + ofl[l].x;
        y = (((float)qy + 0.5f) * ((float)Hl / (float)Hq) - 0.5f) + ofl[l].y;
      } else {
        pixel_coords(ofl[l].x, ofl[l].y, Wl, Hl, x, y);
      }
      const float x0f = floorf(x), y0f = floorf(y);
      const int x0 = (int)x0f, y0 = (int)y0f;
      const int xr = x0 - c.wx0[l], yr = y0 - c.wy0[l];
      const bool inside = x > -1.f && x < (float)Wl && y > -1.f && y < (float)Hl;
      const bool inwin = (unsigned)xr < (unsigned)(ww - 1) && (unsigned)yr < (unsigned)(wh - 1);
      // The quad walks the level's four points WITHOUT branches: a point that is not in the window is read at window address 0
      // and its sums are dropped, so that the reads of point K + 1 can be issued under the dot products of point K (with a
      // branch per point every walk paid its own LDS latency).  Points in the image but outside the window -- rare -- are
      // redone from global memory behind a wave-uniform test.
      const int addr = inwin ? (yr * ww + xr) * 8 : 0;
      float m00 = 0.f, m01 = 0.f, m10 = 0.f, m11 = 0.f;  // this lane's point: the four corner dot products
      auto walk = [&](auto kc) __attribute__((always_inline)) {
        constexpr int K = decltype(kc)::value;
        const float4* cc = wl + bcast_dpp<K>(addr);
        const float d00 = quad_sum_dpp(dot8p(gp, cc[0], cc[4]));
        const float d01 = quad_sum_dpp(dot8p(gp, cc[8], cc[12]));
        const float d10 = quad_sum_dpp(dot8p(gp, cc[ww * 8], cc[ww * 8 + 4]));
        const float d11 = quad_sum_dpp(dot8p(gp, cc[ww * 8 + 8], cc[ww * 8 + 12]));
        if (j == K) { m00 = d00; m01 = d01; m10 = d10; m11 = d11; }
      };
      walk(std::integral_constant<int, 0>{});
      walk(std::integral_constant<int, 1>{});
      walk(std::integral_constant<int, 2>{});
      walk(std::integral_constant<int, 3>{});
      if (!inwin) m00 = m01 = m10 = m11 = 0.f;
      const bool far = inside && !inwin;
      if (__builtin_amdgcn_ballot_w64(far) != 0) {  // general path, global loads
        auto walk_far = [&](auto kc) __attribute__((always_inline)) {
          constexpr int K = decltype(kc)::value;
          if (!bcast_dpp<K>((int)far)) return;  // (quad-uniform)
          const int kx0 = bcast_dpp<K>(x0), ky0 = bcast_dpp<K>(y0);
          const bool xl = kx0 >= 0, xr2 = kx0 + 1 < Wl, yt = ky0 >= 0, yb = ky0 + 1 < Hl;
          const float* p00 = vlev + (int64_t)(ky0 * Wl + kx0) * row_stride;
          float d00 = 0.f, d01 = 0.f, d10 = 0.f, d11 = 0.f;
          if (yt && xl) d00 = dot8(go0, go1, ld4g(p00), ld4g(p00 + 16));
          if (yt && xr2) d01 = dot8(go0, go1, ld4g(p00 + row_stride), ld4g(p00 + row_stride + 16));
          if (yb && xl) d10 = dot8(go0, go1, ld4g(p00 + (int64_t)Wl * row_stride), ld4g(p00 + (int64_t)Wl * row_stride + 16));
          if (yb && xr2)
            d11 = dot8(go0, go1, ld4g(p00 + (int64_t)(Wl + 1) * row_stride), ld4g(p00 + (int64_t)(Wl + 1) * row_stride + 16));
          d00 = quad_sum_dpp(d00);
          d01 = quad_sum_dpp(d01);
          d10 = quad_sum_dpp(d10);
          d11 = quad_sum_dpp(d11);
          if (j == K) { m00 = d00; m01 = d01; m10 = d10; m11 = d11; }
        };
        walk_far(std::integral_constant<int, 0>{});
        walk_far(std::integral_constant<int, 1>{});
        walk_far(std::integral_constant<int, 2>{});
        walk_far(std::integral_constant<int, 3>{});
      }
      const float fx1 = x - x0f, fy1 = y - y0f, fx0 = 1.f - fx1, fy0 = 1.f - fy1;
      const float aw = awl[l];
      // d pixel / d location = (W_l, H_l); d pixel / d offset = 1 (ROWS)
      const float sx = ROWS ? aw : aw * (float)Wl, sy = ROWS ? aw : aw * (float)Hl;
      gwl[l] = inside ? fy0 * (fx0 * m00 + fx1 * m01) + fy1 * (fx0 * m10 + fx1 * m11) : 0.f;
      gxl[l] = inside ? sx * (fy0 * (m01 - m00) + fy1 * (m11 - m10)) : 0.f;
      gyl[l] = inside ? sy * (fx0 * (m10 - m00) + fx1 * (m11 - m01)) : 0.f;
      if (!ROWS) {
        grad_w[pair * (NL * P) + l * P + j] = gwl[l];
        *reinterpret_cast<float2*>(grad_loc + (pair * (NL * P) + l * P + j) * 2) = make_float2(gxl[l], gyl[l]);
      }
    }
    if (ROWS) {
      // softmax backward: d logit_i = w_i (g_i - sum_k w_k g_k)
      float sdot = 0.f;
#pragma unroll
      for (int l = 0; l < NL; ++l) sdot += awl[l] * gwl[l];
      sdot = quad_sum_dpp(sdot);
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        const float gl = awl[l] * (gwl[l] - sdot);
        if (ROWS == 2) {
          unsigned short* gr = reinterpret_cast<unsigned short*>(grad_loc);
          *reinterpret_cast<unsigned*>(gr + off_e + (l * P + j) * 2) = pack_bf16_b(gxl[l], gyl[l]);
          gr[log_e + l * P + j] = (unsigned short)(pack_bf16_b(gl, 0.f) & 0xffffu);
        } else {
          *reinterpret_cast<float2*>(grad_loc + off_e + (l * P + j) * 2) = make_float2(gxl[l], gyl[l]);
          grad_loc[log_e + l * P + j] = gl;
        }
      }
    }
  }
}

#ifdef WM2F_PROFILING  // superseded by the quad form below; kept as its measured baseline
// ---------------------------------------------------------------------------------- kernel B
// Wave-per-query.  Lanes 0..NL*P-1 each work out ONE sampling point (pixel, four weights, flags); the
// wave then walks the points with those values as scalars (v_readlane), and every atomic instruction
// adds two contiguous 128-B rows: lanes 0-31 = the 32 channels of the left corner, lanes 32-63 = the
// right corner.  No redundant coordinate arithmetic, at most 2-way LDS bank conflicts.
//
// DET (wm2f_msdeform_bwd_det): run-to-run identical grad_value.  The window sums are already order-independent
// (integer LDS adds); what was not is the order in which overlapping windows (and the rare out-of-window points) reach
// memory with FLOAT atomics.  Here a tile does not add its window to grad_value at all: it STORES the whole window (plain
// coalesced float4 stores, converted with the tile's scale) into its own slab of a staging buffer, and a second kernel
// (units_gather_kernel) sums, for every grad_value element, the windows that cover it in a fixed tile order.  The rare
// points whose footprint leaves the window are added as INTEGERS into an int64 image in one fixed-point unit per
// (image, head): unit = 2^(E - 44), E = exponent of max|grad_out| over that image and head (a pre-pass); any element's
// true sum is below S * 2^(E + 1) (each of the S queries adds at most its grad_out times weights that sum to <= 1),
// i.e. below 2^62 units for S < 2^17 -- no overflow, and integer addition commutes.  (All adds as 64-bit integer
// atomics were tried first: 7.5 ms against 2.8 ms for the float atomics -- the 64-bit atomic rate; the staging form
// moves 0.6 GB each way as plain traffic instead.)
template <int NL, int P, bool DET>
__global__ __launch_bounds__(kBwdThreads) void msdeform_tiled_bwd_value_kernel(
    const float* __restrict__ loc, const float* __restrict__ attn_w, const float* __restrict__ grad_out,
    float* __restrict__ grad_value, float* __restrict__ staging, long long* __restrict__ acc64, const int* __restrict__ emax_bits,
    TileGeom g, int S, int Q, int heads, int n_logical, int per_xcd) {
  constexpr int D = 32, kWaves = kBwdThreads / kWave, NP = NL * P;
  static_assert(NP <= 64, "one lane per sampling point");
  extern __shared__ __attribute__((aligned(16))) float4 win[];
  const int id = xcd_contiguous_id(blockIdx.x, per_xcd);
  if (id >= n_logical) return;
  int* lv_tab = reinterpret_cast<int*>(win + g.lv_tab_off4);
  const TileCtx<NL> c = tile_setup<NL>(g, id, heads, lv_tab);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ch = lane & 31, hs = lane >> 5;  // channel, left / right corner
  const int row_stride = heads * D;
  float* gvb = grad_value + ((int64_t)c.b * S * heads + c.h) * D;

  for (int i = tid; i < g.lv_tab_off4; i += kBwdThreads) win[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  // Fixed-point accumulation.  An LDS float atomic costs ~190 cycles per wave-instruction on gfx950, an integer one
  // ~5 (tools/probes/lds_atomic_rate.hip), so the window holds int32 sums of round(x * scale[channel]).
  // scale = 2^31 / (512 * max|grad_out| of this tile, head and channel): a window element receives from each of the
  // tile's <= 336 queries at most sum_p attention_weight <= 1 times a bilinear weight <= 1 of that query's
  // grad_out, so |sum| <= 336 * max < 2^31 / scale -- the integer sum cannot overflow.  Each addend is rounded to
  // 2^-22 * max (the fp32 sum it replaces rounds each partial sum to 2^-24 of its own magnitude).
  __shared__ float ch_max[kBwdThreads / kWave][32];
  {
    float m = 0.f;
    for (int qi = wave; qi < c.nq; qi += kWaves) {
      const int q = tile_query<NL>(lv_tab, qi, Q);
      m = fmaxf(m, fabsf(grad_out[(((int64_t)c.b * Q + q) * heads + c.h) * D + ch]));
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    if (lane < 32) ch_max[wave][lane] = m;
  }
  __syncthreads();
  float gmax = 0.f;
#pragma unroll
  for (int w = 0; w < kBwdThreads / kWave; ++w) gmax = fmaxf(gmax, ch_max[w][ch]);
  const float fx_scale = gmax > 0.f ? 4194304.f / gmax : 0.f;  // 2^22 / max
  const float fx_inv = gmax > 0.f ? gmax * (1.f / 4194304.f) : 0.f;
  float det_unit_inv = 0.f;  // DET: 2^(44 - E), a float -> units of the spill image
  long long* acc_b = nullptr;
  __shared__ float ch_inv[32];
  if (DET) {
    int E = ((emax_bits[c.b * heads + c.h] >> 23) & 0xff) - 127;
    if (E < -80) E = -80;
    det_unit_inv = __int_as_float((44 - E + 127) << 23);
    acc_b = acc64 + ((int64_t)c.b * S * heads + c.h) * D;
    if (tid < 32) ch_inv[tid] = fx_inv;  // lane = channel for tid < 32
  }
  int* wini = reinterpret_cast<int*>(win);

  // this lane's point (lanes >= NP idle in the per-point phase): level constants
  const int pl = lane < NP ? lane / P : 0;
  int Wl_p = g.w[0], Hl_p = g.h[0], ww_p = g.win_w[0], wh_p = g.win_h[0], wx0_p = c.wx0[0], wy0_p = c.wy0[0];
  int base_p = g.lds_off4[0] * 4, start_p = g.start[0];
#pragma unroll
  for (int l = 1; l < NL; ++l)
    if (pl == l) {
      Wl_p = g.w[l]; Hl_p = g.h[l]; ww_p = g.win_w[l]; wh_p = g.win_h[l]; wx0_p = c.wx0[l]; wy0_p = c.wy0[l];
      base_p = g.lds_off4[l] * 4; start_p = g.start[l];
    }

  for (int qi = wave; qi < c.nq; qi += kWaves) {
    const int q = tile_query<NL>(lv_tab, qi, Q);
    const int64_t pair = ((int64_t)c.b * Q + q) * heads + c.h;
    const float gof = grad_out[pair * D + ch];
    const float gofs = gof * fx_scale;
    // ---- per-point phase (one lane per point)
    float w00 = 0.f, w01 = 0.f, w10 = 0.f, w11 = 0.f;
    int lds_idx = -1, glb_idx = 0, flags = 0;  // flags: bit0..3 corner in image, bit4 footprint in window
    if (lane < NP) {
      const float2 lc = *reinterpret_cast<const float2*>(loc + pair * (NP * 2) + lane * 2);
      const float aw = attn_w[pair * NP + lane];
      float x, y;
      pixel_coords(lc.x, lc.y, Wl_p, Hl_p, x, y);
      if (x > -1.f && x < (float)Wl_p && y > -1.f && y < (float)Hl_p) {
        const float x0f = floorf(x), y0f = floorf(y);
        const int x0 = (int)x0f, y0 = (int)y0f;
        const float fx1 = x - x0f, fy1 = y - y0f;
        const float a1 = aw * fy1, a0 = aw - a1;
        w01 = a0 * fx1; w00 = a0 - w01; w11 = a1 * fx1; w10 = a1 - w11;
        const bool xl = x0 >= 0, xr = x0 + 1 < Wl_p, yt = y0 >= 0, yb = y0 + 1 < Hl_p;
        flags = (yt && xl ? 1 : 0) | (yt && xr ? 2 : 0) | (yb && xl ? 4 : 0) | (yb && xr ? 8 : 0);
        const int xrw = x0 - wx0_p, yrw = y0 - wy0_p;
        if ((unsigned)xrw < (unsigned)(ww_p - 1) && (unsigned)yrw < (unsigned)(wh_p - 1)) {
          flags |= 16;
          lds_idx = base_p + (yrw * ww_p + xrw) * 32;  // float index of the top-left corner's row
        }
        glb_idx = (start_p + y0 * Wl_p + x0);  // token of the top-left corner (may be out of range: flags)
        flags |= (ww_p << 8) | (Wl_p << 16);   // row pitches for the bottom corners (ww < 256, Wl < 65536)
      }
    }
    // ---- per-corner-row phase: point parameters as wave-uniform scalars
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int f = __builtin_amdgcn_readlane(flags, p);
      if ((f & 15) == 0) continue;  // point outside the image (uniform branch)
      const float s00 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w00), p));
      const float s01 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w01), p));
      const float s10 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w10), p));
      const float s11 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w11), p));
      const int li = __builtin_amdgcn_readlane(lds_idx, p), gi = __builtin_amdgcn_readlane(glb_idx, p);
      const int wwp = (f >> 8) & 0xff, Wlp = (f >> 16) & 0xffff;
      const float wt = hs ? s01 : s00, wb = hs ? s11 : s10;
      const bool ok_t = (f >> hs) & 1, ok_b = (f >> (2 + hs)) & 1;
      if (f & 16) {  // whole footprint inside the window: LDS accumulation
        if (ok_t) atomicAdd(wini + li + hs * 32 + ch, __float2int_rn(wt * gofs));
        if (ok_b) atomicAdd(wini + li + (wwp + hs) * 32 + ch, __float2int_rn(wb * gofs));
      } else if (DET) {  // rare: straight to memory, in units
        if (ok_t) atomicAdd(reinterpret_cast<unsigned long long*>(acc_b + (int64_t)(gi + hs) * row_stride + ch),
                            (unsigned long long)__float2ll_rn(wt * gof * det_unit_inv));
        if (ok_b) atomicAdd(reinterpret_cast<unsigned long long*>(acc_b + (int64_t)(gi + Wlp + hs) * row_stride + ch),
                            (unsigned long long)__float2ll_rn(wb * gof * det_unit_inv));
      } else {  // rare: straight to memory
        if (ok_t) atomicAdd(gvb + (int64_t)(gi + hs) * row_stride + ch, wt * gof);
        if (ok_b) atomicAdd(gvb + (int64_t)(gi + Wlp + hs) * row_stride + ch, wb * gof);
      }
    }
  }
  __syncthreads();
  if (DET) {  // the whole window, converted, into this tile's slab (float4 i holds channels 4 (i % 8) ..)
    float4* slab = reinterpret_cast<float4*>(staging) + (int64_t)id * g.lv_tab_off4;
    const int4* wi4 = reinterpret_cast<const int4*>(win);
    for (int i = tid; i < g.lv_tab_off4; i += kBwdThreads) {
      const int4 v = wi4[i];
      const float4 sc = *reinterpret_cast<const float4*>(ch_inv + 4 * (i & 7));
      slab[i] = make_float4((float)v.x * sc.x, (float)v.y * sc.y, (float)v.z * sc.z, (float)v.w * sc.w);
    }
    return;
  }
  // flush: one lane per channel, 32 lanes per pixel -> every atomic wave-instruction is two whole 128-B rows
  const int pslot = tid >> 5;
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    const int Wl = g.w[l], Hl = g.h[l], ww = g.win_w[l];
    const int npix = ww * g.win_h[l];
    const float inv_ww = 1.f / (float)ww;
    const int* wli = wini + g.lds_off4[l] * 4;
    float* glev = gvb + (int64_t)g.start[l] * row_stride;
    for (int idx = pslot; idx < npix; idx += kBwdThreads / 32) {
      const int wy = (int)(((float)idx + 0.5f) * inv_ww), wx = idx - wy * ww;
      const int x = c.wx0[l] + wx, y = c.wy0[l] + wy;
      if (x < 0 || x >= Wl || y < 0 || y >= Hl) continue;
      const int vi = wli[idx * 32 + ch];
      if (vi != 0) atomicAdd(glev + (int64_t)(y * Wl + x) * row_stride + ch, (float)vi * fx_inv);
    }
  }
}

#endif  // WM2F_PROFILING (wave-per-query value kernel)

typedef __attribute__((address_space(3))) int lds_int_t;

// ---------------------------------------------------------------------------------- kernel B, quad form
// Same windows, fixed-point sums, flush / slab and out-of-window rule as the kernel above, but the work is laid out as in
// the forward's quad kernels: a wave pass = 16 queries x 4 lanes, each lane owning the 8 channels {4k + j}.  Coordinates,
// bilinear weights and window tests are then computed once per (query, point) by every lane of the quad in parallel
// for 16 queries, instead of by 12 lanes of a wave whose other 52 wait and which then walks the 12 points serially with
// six v_readlane each: the wave-per-query kernel issued 19.2 k vector + 10 k scalar instructions per wave (PMC: vector
// pipe 51 % busy, the rest waits) -- 400 vector instructions per (query, head) for its 24 LDS atomics; this form needs
// ~100.  LDS banks: at a fixed k all lanes touch dword 4k + j of their pixel row, i.e. the same 4 banks per row parity;
// the row is therefore stored XOR-swizzled, channel c of window pixel r at dword c ^ 4 ((r >> 1) & 7), which spreads 16
// neighbouring pixels over all 64 banks and costs nothing (v_xad_u32 forms (4k ^ 4s) + base in one instruction); the
// flush and the slab store undo it.
// NT threads (16 waves: 4 per SIMD, so that one wave's vector work runs under another's LDS atomics), and a pass's 12
// points split over NSPLIT waves (21 passes per tile do not divide over 16 waves; 42 half-passes nearly do).
template <int NL, int P, bool DET, int NT = kQuadBwdThreads, int NSPLIT = 2, int ROWS = 0>
__global__ __launch_bounds__(NT) void msdeform_tiled_bwd_value_quad_kernel(
    const float* __restrict__ loc, const float* __restrict__ attn_w, const float* __restrict__ grad_out,
    float* __restrict__ grad_value, float* __restrict__ staging, long long* __restrict__ acc64, const int* __restrict__ emax_bits,
    TileGeom g, int S, int Q, int heads, int n_logical, int per_xcd) {
  constexpr int D = 32, kWaves = NT / kWave, NP = NL * P;
  extern __shared__ __attribute__((aligned(16))) float4 win[];
  const int id = xcd_contiguous_id(blockIdx.x, per_xcd);
  if (id >= n_logical) return;
  int* lv_tab = reinterpret_cast<int*>(win + g.lv_tab_off4);
  const TileCtx<NL> c = tile_setup<NL>(g, id, heads, lv_tab);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slot = lane >> 2, j = lane & 3;
  const int row_stride = heads * D;
  float* gvb = grad_value + ((int64_t)c.b * S * heads + c.h) * D;

  for (int i = tid; i < g.lv_tab_off4; i += NT) win[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  // grad_out element (ROWS = 2: bf16)
  auto go_at = [&](int64_t i) __attribute__((always_inline)) {
    if (ROWS == 2) return __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(grad_out)[i] << 16);
    return grad_out[i];
  };
  // per-channel scale of the fixed-point sums: as in the kernel above (2^22 / max |grad_out| of this tile, head, channel)
  __shared__ float ch_max[NT / kWave][32];
  __shared__ float ch_scale[32], ch_inv[32];
  {
    const int ch = lane & 31;
    float m = 0.f;
    for (int qi = wave * 2 + (lane >> 5); qi < c.nq; qi += 2 * kWaves) {
      const int q = tile_query<NL>(lv_tab, qi, Q);
      m = fmaxf(m, fabsf(go_at((((int64_t)c.b * Q + q) * heads + c.h) * D + ch)));
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    if (lane < 32) ch_max[wave][lane] = m;
  }
  __syncthreads();
  if (tid < 32) {
    float gmax = 0.f;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) gmax = fmaxf(gmax, ch_max[w][tid]);
    ch_scale[tid] = gmax > 0.f ? 4194304.f / gmax : 0.f;  // 2^22 / max
    ch_inv[tid] = gmax > 0.f ? gmax * (1.f / 4194304.f) : 0.f;
  }
  __syncthreads();
  float sc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) sc[k] = ch_scale[4 * k + j];
  float det_unit_inv = 0.f;
  long long* acc_b = nullptr;
  if (DET) {
    int E = ((emax_bits[c.b * heads + c.h] >> 23) & 0xff) - 127;
    if (E < -80) E = -80;
    det_unit_inv = __int_as_float((44 - E + 127) << 23);
    acc_b = acc64 + ((int64_t)c.b * S * heads + c.h) * D;
  }
  int* wini = reinterpret_cast<int*>(win);
  const unsigned lds_base = (unsigned)(size_t)(lds_int_t*)wini;  // 32-bit LDS address of the window

  const int n_unit = ((c.nq + 15) >> 4) * NSPLIT;
  for (int unit = wave; unit < n_unit; unit += kWaves) {
    const int pass = unit / NSPLIT, part_pts = unit - pass * NSPLIT;
    const int qi = pass * 16 + slot;
    const bool valid = qi < c.nq;
    int Wq = 1, Hq = 1, qx = 0, qy = 0;
    const int q = ROWS ? tile_query_ex<NL>(lv_tab, valid ? qi : c.nq - 1, Q, Wq, Hq, qx, qy) : tile_query<NL>(lv_tab, valid ? qi : c.nq - 1, Q);
    const int64_t pair = ((int64_t)c.b * Q + q) * heads + c.h;
    float raw[8], gs[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      raw[k] = valid ? go_at(pair * D + 4 * k + j) : 0.f;
      gs[k] = raw[k] * sc[k];
    }
    const float* lp = loc + pair * (NP * 2);
    const float* ap = attn_w + pair * NP;
    // ROWS: softmax over the (query, head)'s logits and the reference pixel, redone here (see load_logits)
    const int64_t row_e = ((int64_t)c.b * Q + q) * (heads * NP * 3);
    const int64_t off_e = row_e + c.h * (NP * 2), log_e = row_e + heads * (NP * 2) + c.h * NP;
    float awr[NP];
    float2 ofr[NP];  // ROWS: the unit's offsets, requested together with the logits (loaded point by point, behind each point's
                     // branches, every point paid its own trip to L2)
    if (ROWS) {
#pragma unroll
      for (int i = 0; i < NP; ++i)
        if (NSPLIT == 1 || i % NSPLIT == part_pts) ofr[i] = load_offset<ROWS>(loc, off_e + i * 2);
      load_logits<ROWS, NP>(loc, log_e, awr);
      softmax_regs<NP>(awr);
    }
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      const int Wl = g.w[l], Hl = g.h[l], ww = g.win_w[l], wh = g.win_h[l];
      const int base = g.lds_off4[l] * 4 + j;
      const float rpx = ((float)qx + 0.5f) * ((float)Wl / (float)Wq) - 0.5f, rpy = ((float)qy + 0.5f) * ((float)Hl / (float)Hq) - 0.5f;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        if (NSPLIT > 1 && (l * P + p) % NSPLIT != part_pts) continue;  // wave-uniform
        float aw, x, y;
        if (ROWS) {
          const float2 of = ofr[l * P + p];
          aw = valid ? awr[l * P + p] : 0.f;
          x = rpx + of.x;
          y = rpy + of.y;
        } else {
          const float2 lc = *reinterpret_cast<const float2*>(lp + (l * P + p) * 2);
          aw = valid ? ap[l * P + p] : 0.f;
          pixel_coords(lc.x, lc.y, Wl, Hl, x, y);
        }
        if (!(x > -1.f && x < (float)Wl && y > -1.f && y < (float)Hl) || aw == 0.f) continue;  // per lane (quad-uniform)
        const float x0f = floorf(x), y0f = floorf(y);
        const int x0 = (int)x0f, y0 = (int)y0f;
        const float fx1 = x - x0f, fy1 = y - y0f;
        const float a1 = aw * fy1, a0 = aw - a1;
        const float w01 = a0 * fx1, w00 = a0 - w01, w11 = a1 * fx1, w10 = a1 - w11;
        const bool xl = x0 >= 0, xr = x0 + 1 < Wl, yt = y0 >= 0, yb = y0 + 1 < Hl;
        const int xrw = x0 - c.wx0[l], yrw = y0 - c.wy0[l];
        const bool inwin = (unsigned)xrw < (unsigned)(ww - 1) && (unsigned)yrw < (unsigned)(wh - 1);
        const float wc[4] = {w00, w01, w10, w11};
        const bool ok[4] = {yt && xl, yt && xr, yb && xl, yb && xr};
        if (inwin) {
          const int r0 = yrw * ww + xrw;  // window pixel of the top-left corner
#pragma unroll
          for (int cn = 0; cn < 4; ++cn) {
            if (!ok[cn]) continue;
            const int r = r0 + (cn & 1) + (cn >> 1) * ww;
            // byte address of (pixel r, channel 4k + j): row base + ((16 k) ^ (16 s)); (x ^ c) + y is one v_xad_u32
            const unsigned sx = (unsigned)((r >> 1) & 7) << 4;
            const unsigned ab = lds_base + (unsigned)(base + r * 32) * 4u;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              // round half up in ONE instruction (v_cvt_rpi_i32_f32 = floor(x + 0.5); |x| < 2^22): the v_rndne + v_cvt pair
              // of __float2int_rn and the two-instruction address were 4 of the 5 vector instructions per LDS atomic
              int v;
              asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(v) : "v"(wc[cn] * gs[k]));
              __hip_atomic_fetch_add(reinterpret_cast<lds_int_t*>((size_t)((sx ^ (16u * k)) + ab)), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
          }
        } else {  // rare: straight to memory
          const int tok0 = g.start[l] + y0 * Wl + x0;
#pragma unroll
          for (int cn = 0; cn < 4; ++cn) {
            if (!ok[cn]) continue;
            const int64_t o = (int64_t)(tok0 + (cn & 1) + (cn >> 1) * Wl) * row_stride + j;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              if (DET) atomicAdd(reinterpret_cast<unsigned long long*>(acc_b + o + 4 * k),
                                 (unsigned long long)__float2ll_rn(wc[cn] * raw[k] * det_unit_inv));
              else atomicAdd(gvb + o + 4 * k, wc[cn] * raw[k]);
            }
          }
        }
      }
    }
  }
  __syncthreads();
  if (DET) {  // the whole window, converted and un-swizzled, into this tile's slab
    float4* slab = reinterpret_cast<float4*>(staging) + (int64_t)id * g.lv_tab_off4;
    const int4* wi4 = reinterpret_cast<const int4*>(win);
    for (int i = tid; i < g.lv_tab_off4; i += NT) {
      int l = 0;
#pragma unroll
      for (int k = 1; k < NL; ++k) l += (i >= g.lds_off4[k]) ? 1 : 0;
      int o4 = g.lds_off4[0];
#pragma unroll
      for (int k = 1; k < NL; ++k)
        if (l == k) o4 = g.lds_off4[k];
      const int r = (i - o4) >> 3, c4 = (i - o4) & 7;  // window pixel, channel group 4 c4 .. 4 c4 + 3
      const int4 v = wi4[o4 + r * 8 + (c4 ^ ((r >> 1) & 7))];
      const float4 si = *reinterpret_cast<const float4*>(ch_inv + 4 * c4);
      slab[i] = make_float4((float)v.x * si.x, (float)v.y * si.y, (float)v.z * si.z, (float)v.w * si.w);
    }
    return;
  }
  // flush: one lane per channel, 32 lanes per pixel -> every atomic wave-instruction is two whole 128-B rows
  const int pslot = tid >> 5, ch = lane & 31;
  const float fx_inv = ch_inv[ch];
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    const int Wl = g.w[l], Hl = g.h[l], ww = g.win_w[l];
    const int npix = ww * g.win_h[l];
    const float inv_ww = 1.f / (float)ww;
    const int* wli = wini + g.lds_off4[l] * 4;
    float* glev = gvb + (int64_t)g.start[l] * row_stride;
    for (int idx = pslot; idx < npix; idx += NT / 32) {
      const int wy = (int)(((float)idx + 0.5f) * inv_ww), wx = idx - wy * ww;
      const int x = c.wx0[l] + wx, y = c.wy0[l] + wy;
      if (x < 0 || x >= Wl || y < 0 || y >= Hl) continue;
      const int vi = wli[idx * 32 + (ch ^ (((idx >> 1) & 7) << 2))];
      if (vi != 0) atomicAdd(glev + (int64_t)(y * Wl + x) * row_stride + ch, (float)vi * fx_inv);
    }
  }
}

// max |grad_out| per (image, head) as float bits (non-negative floats order like their bit patterns): blockDim = heads * D
// threads, one (head, channel) each, striding over the queries of one image chunk.
__global__ void absmax_image_head_kernel(const float* __restrict__ go, int* __restrict__ emax_bits, int Q, int heads, int D, int rows_per_block) {
  const int b = blockIdx.y, t = threadIdx.x, row = heads * D;
  const int q0 = blockIdx.x * rows_per_block, q1 = min(Q, q0 + rows_per_block);
  float m = 0.f;
  for (int q = q0; q < q1; ++q) m = fmaxf(m, fabsf(go[((int64_t)b * Q + q) * row + t]));
  // D = 32: the head's channels are the 32 lanes of a half wave -- one atomic per (workgroup, head); one per thread
  // (all on B * heads words) cost 4.1 ms
#pragma unroll
  for (int o = 16; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((t & 31) == 0) atomicMax(emax_bits + b * heads + t / D, __float_as_int(m));
}

// grad_value element (b, token, head, channel) = sum over the tiles whose window covers the token's pixel, in (ty, tx)
// order, of that window's value, plus the out-of-window spill in units.  One workgroup per token: thread = (head, channel).
// The covering tiles are the same for the whole workgroup: the first wave tests the <= 64 candidate tiles (one per lane,
// the integer divisions of the window origins happen once per token, not once per element) and compacts the hits, in
// candidate order, into LDS.
template <int NL>
__global__ void units_gather_kernel(const float* __restrict__ staging, const long long* __restrict__ acc, const int* __restrict__ emax_bits,
                                    float* __restrict__ out, TileGeom g, int S, int heads, int reach) {
  const int tok = blockIdx.x, b = blockIdx.y, t = threadIdx.x;  // blockDim = heads * 32
  const int h = t >> 5, ch = t & 31;
  __shared__ int n_cov;
  __shared__ int cov_tile[64], cov_woff[64];
  int l = 0;
#pragma unroll
  for (int k = 1; k < NL; ++k) l += (tok >= g.start[k]) ? 1 : 0;
  int off4 = g.lds_off4[0];
#pragma unroll
  for (int k = 1; k < NL; ++k)
    if (l == k) off4 = g.lds_off4[k];
  if (t < 64) {
    int Wl = g.w[0], Hl = g.h[0], ww = g.win_w[0], wh = g.win_h[0], st = g.start[0];
#pragma unroll
    for (int k = 1; k < NL; ++k)
      if (l == k) { Wl = g.w[k]; Hl = g.h[k]; ww = g.win_w[k]; wh = g.win_h[k]; st = g.start[k]; }
    const int y = (tok - st) / Wl, x = (tok - st) - y * Wl;
    const int Wf = g.w[g.fine], Hf = g.h[g.fine];
    // tile whose region holds the pixel centre, then every tile within `reach` of it (host: margin / region + 2)
    const int txc = min(g.tiles_x - 1, (int)(((int64_t)(2 * x + 1) * Wf) / ((int64_t)2 * g.F * Wl)));
    const int tyc = min(g.tiles_y - 1, (int)(((int64_t)(2 * y + 1) * Hf) / ((int64_t)2 * g.F * Hl)));
    const int side = 2 * reach + 1;  // host: side * side <= 64
    const int ty = tyc - reach + t / side, tx = txc - reach + t % side;
    bool hit = false;
    int woff = 0;
    if (t < side * side && ty >= 0 && ty < g.tiles_y && tx >= 0 && tx < g.tiles_x) {
      const int wy = y - (floor_div_i(2 * ty * g.F * Hl - Hf, 2 * Hf) - g.M);
      const int wx = x - (floor_div_i(2 * tx * g.F * Wl - Wf, 2 * Wf) - g.M);
      hit = (unsigned)wy < (unsigned)wh && (unsigned)wx < (unsigned)ww;
      woff = (wy * ww + wx) * 32;
    }
    const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
    if (hit) {
      const int pos = __builtin_popcountll(m & ((1ull << t) - 1ull));
      cov_tile[pos] = ty * g.tiles_x + tx;
      cov_woff[pos] = woff;
    }
    if (t == 0) n_cov = __builtin_popcountll(m);
  }
  __syncthreads();
  const int n_tiles = g.tiles_x * g.tiles_y;
  float sum = 0.f;
  for (int k = 0; k < n_cov; ++k) {
    const int64_t id = ((int64_t)b * n_tiles + cov_tile[k]) * heads + h;
    sum += staging[(id * g.lv_tab_off4 + off4) * 4 + cov_woff[k] + ch];
  }
  const int64_t i = (((int64_t)b * S + tok) * heads + h) * 32 + ch;
  const long long sp = acc[i];
  if (sp != 0) {
    int E = ((emax_bits[b * heads + h] >> 23) & 0xff) - 127;
    if (E < -80) E = -80;
    sum += (float)ldexp((double)sp, E - 44);
  }
  out[i] = sum;
}

static inline int64_t tiled_bwd_det_pad(int B, int heads) { return ((int64_t)B * heads * 4 + 255) / 256 * 256; }

// bytes of the deterministic form's workspace for these levels (0: the LDS-window backward does not take them)
int64_t tiled_bwd_det_workspace(const int32_t* level_hw, int B, int S, int heads, int L) {
  if (L < 1 || L > kMaxLv) return 0;
  TiledPlan p = plan_tiled(level_hw, L, 4);
  if (!p.ok) return 0;
  const int64_t n_logical = (int64_t)B * heads * p.g.tiles_x * p.g.tiles_y;
  return (int64_t)B * S * heads * 32 * 8 + tiled_bwd_det_pad(B, heads) + n_logical * p.g.lv_tab_off4 * 16;
}

int launch_tiled_bwd(const void* value, const void* loc, const void* attn_w, const void* grad_out, void* grad_value,
                     void* grad_loc, void* grad_w, const int32_t* level_hw, int B, int S, int Q, int heads, int L, int P,
                     int margin, void* stream, const char* who, bool* handled, void* det_ws, int rows) {
  // rows: 0 = loc / attn_w operands; 1 / 2 = the ROWS forms (fp32 / bf16 [offsets | logits] rows in `loc`, their gradient in
  // `grad_loc`; attn_w and grad_w unused): 3 levels, an even head count (16-byte row pieces), no deterministic form
  *handled = false;
  if (P != 4 || L < 1 || L > kMaxLv || margin < 0 || (int64_t)Q != S) return WM2F_OK;
  if (rows && (L != 3 || det_ws || (heads & 1) || rows > 2)) return WM2F_OK;
  TiledPlan p = plan_tiled(level_hw, L, margin);
  if (!p.ok) return WM2F_OK;
  p.g.order = 1;
  p.g.a_qstride = heads * L * P * 2;
  p.g.b_qstride = heads * L * P;
  const int64_t n_logical = (int64_t)B * heads * p.g.tiles_x * p.g.tiles_y;
  if (n_logical > (1 << 30)) return WM2F_OK;
  const int per_xcd = (int)ceil_div64(n_logical, kNumXcd);
  hipStream_t st = (hipStream_t)stream;
#ifdef WM2F_PROFILING
  const char* e_old = getenv("WM2F_K1_BWD_OLD");  // profiling build: the wave-per-query value kernel, for A/B
  const bool old_form = e_old && *e_old == '1';
#else
  const bool old_form = false;
#endif
  long long* acc64 = nullptr;
  int* emax_bits = nullptr;
  float* staging = nullptr;
  const int64_t n_elem = (int64_t)B * S * heads * 32;
  int reach = 0;
  if (det_ws) {  // [int64 spill image of grad_value][B * heads exponent words, padded][one window slab per tile]
    if (heads * 32 > 1024 || p.g.order != 1) return WM2F_OK;
    acc64 = (long long*)det_ws;
    emax_bits = (int*)((char*)det_ws + n_elem * 8);
    staging = (float*)((char*)det_ws + n_elem * 8 + tiled_bwd_det_pad(B, heads));
    hipError_t em = hipMemsetAsync(det_ws, 0, (size_t)n_elem * 8 + (size_t)B * heads * 4, st);
    if (em != hipSuccess) {
      set_error("%s: clearing the workspace failed: %s", who, hipGetErrorString(em));
      return WM2F_ELAUNCH;
    }
    const int rows = 64;
    hipLaunchKernelGGL(absmax_image_head_kernel, dim3(ceil_div(Q, rows), B), dim3(heads * 32), 0, st, (const float*)grad_out, emax_bits,
                       Q, heads, 32, rows);
    for (int l = 0; l < L; ++l) {  // how many tiles away a window can still cover a pixel of level l
      const int region_x = p.g.F * p.g.w[l] / p.g.w[p.g.fine], region_y = p.g.F * p.g.h[l] / p.g.h[p.g.fine];
      const int r = p.g.M / (region_x < 1 ? 1 : region_x) + 2, r2 = p.g.M / (region_y < 1 ? 1 : region_y) + 2;
      reach = reach > r ? reach : r;
      reach = reach > r2 ? reach : r2;
    }
    if ((2 * reach + 1) * (2 * reach + 1) > 64 || heads * 32 < 64) return WM2F_OK;  // the gather tests its candidates one per lane
  }
  if (rows) {
    // 16 waves per workgroup (110 registers per lane): 2-3 % faster than 8 on the same box (2450 / 2467 / 2485 against 2503 / 2538 /
    // 2552 us for the whole backward at config 2), 12 waves in between
    auto ka = rows == 2 ? msdeform_tiled_bwd_lw_kernel<3, 4, 2, 1024> : msdeform_tiled_bwd_lw_kernel<3, 4, 1, 1024>;
    int ka_threads = 1024;
#ifdef WM2F_PROFILING
    const char* e_lw = getenv("WM2F_K1_LW_THREADS");  // profiling build: 8 / 12 waves per workgroup, for A/B
    if (e_lw && atoi(e_lw) == 512) {
      ka = rows == 2 ? msdeform_tiled_bwd_lw_kernel<3, 4, 2, 512> : msdeform_tiled_bwd_lw_kernel<3, 4, 1, 512>;
      ka_threads = 512;
    }
    if (e_lw && atoi(e_lw) == 768) {
      ka = rows == 2 ? msdeform_tiled_bwd_lw_kernel<3, 4, 2, 768> : msdeform_tiled_bwd_lw_kernel<3, 4, 1, 768>;
      ka_threads = 768;
    }
    if (e_lw && atoi(e_lw) == 1 && rows == 2) ka = msdeform_tiled_bwd_lw_kernel<3, 4, 2, 1024, 1>;  // timing ablations
    if (e_lw && atoi(e_lw) == 2 && rows == 2) ka = msdeform_tiled_bwd_lw_kernel<3, 4, 2, 1024, 2>;
    const bool skip_kb = e_lw && atoi(e_lw) == 3;  // the lw kernel alone
#endif
    auto kb = rows == 2 ? msdeform_tiled_bwd_value_quad_kernel<3, 4, false, kQuadBwdThreads, 2, 2>
                        : msdeform_tiled_bwd_value_quad_kernel<3, 4, false, kQuadBwdThreads, 2, 1>;
    if (p.lds_bytes > 64 * 1024) {
      hipError_t e1 = hipFuncSetAttribute((const void*)ka, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
      hipError_t e2 = hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
      if (e1 != hipSuccess || e2 != hipSuccess) {
        set_error("%s: cannot raise dynamic LDS to %zu", who, p.lds_bytes);
        return WM2F_ELAUNCH;
      }
    }
    hipLaunchKernelGGL(ka, dim3(per_xcd * kNumXcd), dim3(ka_threads), p.lds_bytes, st, (const float*)value, (const float*)loc,
                       (const float*)nullptr, (const float*)grad_out, (float*)grad_loc, (float*)grad_value, p.g, S, Q, heads, (int)n_logical, per_xcd);
#ifdef WM2F_PROFILING
    if (!(skip_kb || (e_lw && (atoi(e_lw) == 1 || atoi(e_lw) == 2))))
#endif
    hipLaunchKernelGGL(kb, dim3(per_xcd * kNumXcd), dim3(kQuadBwdThreads), p.lds_bytes, st, (const float*)loc, (const float*)nullptr,
                       (const float*)grad_out, (float*)grad_value, (float*)nullptr, (long long*)nullptr, (const int*)nullptr, p.g, S, Q, heads,
                       (int)n_logical, per_xcd);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
      set_error("%s: tiled backward launch failed: %s", who, hipGetErrorString(e));
      return WM2F_ELAUNCH;
    }
    *handled = true;
    return WM2F_OK;
  }
#ifdef WM2F_PROFILING  // the wave-per-query value kernel exists in the profiling build only (A/B: WM2F_K1_BWD_OLD=1)
#define WM2F_TB_OLD(NLv) \
  if (old_form) kb = det_ws ? msdeform_tiled_bwd_value_kernel<NLv, 4, true> : msdeform_tiled_bwd_value_kernel<NLv, 4, false>;
#else
#define WM2F_TB_OLD(NLv)
#endif
#define WM2F_TB(NLv)                                                                                              \
  case NLv: {                                                                                                     \
    auto ka = msdeform_tiled_bwd_lw_kernel<NLv, 4>;                                                               \
    auto kb = det_ws ? msdeform_tiled_bwd_value_quad_kernel<NLv, 4, true> : msdeform_tiled_bwd_value_quad_kernel<NLv, 4, false>; \
    WM2F_TB_OLD(NLv)                                                                                              \
    if (p.lds_bytes > 64 * 1024) {                                                                                \
      hipError_t e1 = hipFuncSetAttribute((const void*)ka, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes); \
      hipError_t e2 = hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes); \
      if (e1 != hipSuccess || e2 != hipSuccess) {                                                                 \
        set_error("%s: cannot raise dynamic LDS to %zu", who, p.lds_bytes);                                       \
        return WM2F_ELAUNCH;                                                                                      \
      }                                                                                                           \
    }                                                                                                             \
    hipLaunchKernelGGL(ka, dim3(per_xcd* kNumXcd), dim3(kBwdThreads), p.lds_bytes, st, (const float*)value,       \
                       (const float*)loc, (const float*)attn_w, (const float*)grad_out, (float*)grad_loc,         \
                       (float*)grad_w, p.g, S, Q, heads, (int)n_logical, per_xcd);                                \
    hipLaunchKernelGGL(kb, dim3(per_xcd* kNumXcd), dim3(old_form ? kBwdThreads : kQuadBwdThreads), p.lds_bytes, st, (const float*)loc, \
                       (const float*)attn_w, (const float*)grad_out, (float*)grad_value, staging, acc64, emax_bits,  \
                       p.g, S, Q, heads, (int)n_logical, per_xcd);                                                \
    if (det_ws)                                                                                                   \
      hipLaunchKernelGGL(units_gather_kernel<NLv>, dim3(S, B), dim3(heads * 32), 0, st, (const float*)staging,    \
                         (const long long*)acc64, (const int*)emax_bits, (float*)grad_value, p.g, S, heads, reach); \
  } break;
  switch (L) {
    WM2F_TB(1) WM2F_TB(2) WM2F_TB(3) WM2F_TB(4)
    default: return WM2F_OK;
  }
#undef WM2F_TB
#undef WM2F_TB_OLD
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: tiled backward launch failed: %s", who, hipGetErrorString(e));
    return WM2F_ELAUNCH;
  }
  *handled = true;
  return WM2F_OK;
}

}  // namespace wm2f
