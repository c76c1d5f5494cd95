// K2: masked cross-attention of the transformer decoder (queries x feature-map keys).
// Replaces the attention arithmetic of nn.MultiheadAttention as called at transformers
// modeling_mask2former.py:1644-1650 (torch/nn/functional.py:6578-6600):
//     softmax(bias + (q / sqrt(D)) k^T) v,   bias = -inf where the predicted mask blocks a key,
// with the rule of :1912-1914 (a query whose row is fully blocked attends everywhere) applied
// through row_open.  The x num_heads replicated bool mask (:2052) is never materialised: every
// head reads the same (B, Q, N) byte mask.
//
//   q (B, Q, heads*D) pre-scaled;  k, v (B, N, heads*D);  mask (B, Q, N) u8;  row_open (B, Q) i32
//   out (B, Q, heads*D);  lse (B, heads, Q)
//
// Structure: flash-style, split over keys.  grid = (n_splits * q_chunks, heads, B); each of the 4
// waves of a workgroup walks 16-key tiles of its split with an online softmax, all in fp32 on
// v_mfma_f32_16x16x4_f32:
//   S^T = K Q^T   (A = K tile [key][d], B = Q^T [d][query])  -> C layout: column = query (lane&15),
//                  rows = keys 4g+r.  Computing S TRANSPOSED makes the softmax output directly the
//                  B operand of the next product, no lane movement:
//   O^T += V^T P^T (A = V^T [d][key], B = P^T [key][query]).
// Partial (m, l, O) of the 4 waves are merged through LDS, one partial per (split) goes to the
// workspace, and a second kernel merges the splits and writes out / lse.
//
// Roofline: at fp32 the op is MFMA-bound, not HBM-bound: 4*B*heads*Q*N*D flop
// (13.4 GFLOP at N = 16384, config 2) vs 2*B*N*heads*D*4 B of K/V (268 MB).
#include "common.h"
#include <stdlib.h>

namespace wm2f {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kXWaves = 4;

// Experiment knobs of tools/kbench.py (WM2F_K2_*): environment variables in the PROFILING build (libwm2f_prof.so,
// -DWM2F_PROFILING) only; the production library is compiled with the defaults and reads no environment.
#ifdef WM2F_PROFILING
static inline int tune_env(const char* name, int dflt) {
  const char* e = getenv(name);
  return e && *e ? atoi(e) : dflt;
}
#else
static inline constexpr int tune_env(const char*, int dflt) { return dflt; }
#endif

// Key splits.  Forward: measured at config 2 (tools/kbench.py --only k2, knobs WM2F_K2_QTILES / WM2F_K2_WG_TARGET):
// 2 query tiles per workgroup (4 query chunks, ~150 registers -> 3 waves per SIMD instead of 1 at 7 tiles) and 512
// (image, head, split) groups: 33 / 76 / 256 us at N = 1024 / 4096 / 16384 against 57 / 110 / 375 us before.
static inline int xattn_splits(int B, int heads, int N, bool fwd = true) {
  const int n_tiles = ceil_div(N, 16);
  // forward target re-swept with the full-tile kernel (192 ... 2048): 256 groups below 1024 key tiles (25 / 58 us at
  // N = 1024 / 4096 against 31 / 61 with 512), 512 from there on (201 us at N = 16384 against 204)
  int s = ceil_div(fwd ? tune_env("WM2F_K2_WG_TARGET", n_tiles >= 1024 ? 512 : 256) : 1024, B * heads);
  const int max_s = ceil_div(n_tiles, kXWaves);
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  return s;
}

// Epilogue of the key-split kernels: the 4 waves' partial (O, m, l) of a workgroup meet in LDS, are combined and go to the
// workspace as ONE partial per (image, head, query, split).  `m_scale` converts the running max to natural-log units
// (1 for the general kernel, ln 2 for the log2-domain one; p and l are the same numbers in either base).
template <int NQT, int D>
__device__ __forceinline__ void xattn_merge_waves(float (&part)[kXWaves][NQT * 16][D + 4], const f32x4 (&o)[D / 16][NQT],
                                                  const float (&m)[NQT], const float (&l)[NQT], float m_scale, int wave,
                                                  int g, int n, float* __restrict__ ws, int b, int h, int heads, int Q,
                                                  int q0, int n_splits, int split) {
  constexpr int DT = D / 16, QL = NQT * 16, RS = D + 4;
  const float NEG_INF = -INFINITY;
#pragma unroll
  for (int j = 0; j < NQT; ++j) {
    float lt = l[j];
    lt += __shfl_xor(lt, 16, kWave);
    lt += __shfl_xor(lt, 32, kWave);
    float* row = &part[wave][16 * j + n][0];
#pragma unroll
    for (int i = 0; i < DT; ++i) *reinterpret_cast<f32x4*>(row + 16 * i + 4 * g) = o[i][j];
    if (g == 0) {
      row[D] = m[j] * m_scale;  // -inf stays -inf
      row[D + 1] = lt;
    }
  }
  __syncthreads();
  // thread -> (query row, float4 chunk of the RS-wide row); chunk D/4 carries (m, l)
  constexpr int CH = D / 4 + 1;
  for (int idx = threadIdx.x; idx < QL * CH; idx += kXWaves * kWave) {
    const int ql = idx / CH, c = idx - ql * CH;
    const int qi = q0 + ql;
    if (qi >= Q) continue;
    float mw[kXWaves], M = NEG_INF;
#pragma unroll
    for (int w = 0; w < kXWaves; ++w) {
      mw[w] = part[w][ql][D];
      M = fmaxf(M, mw[w]);
    }
    const float Ms = (M == NEG_INF) ? 0.f : M;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float L = 0.f;
#pragma unroll
    for (int w = 0; w < kXWaves; ++w) {
      const float sc = __expf(mw[w] - Ms);
      if (c < D / 4) {
        const float4 x = *reinterpret_cast<const float4*>(&part[w][ql][4 * c]);
        acc.x += sc * x.x; acc.y += sc * x.y; acc.z += sc * x.z; acc.w += sc * x.w;
      } else {
        L += sc * part[w][ql][D + 1];
      }
    }
    float* wrow = ws + ((((int64_t)b * heads + h) * Q + qi) * n_splits + split) * RS;
    if (c < D / 4)
      *reinterpret_cast<float4*>(wrow + 4 * c) = acc;
    else
      *reinterpret_cast<float4*>(wrow + D) = make_float4(M, L, 0.f, 0.f);
  }
}

template <int NQT, int D>
__global__ __launch_bounds__(kXWaves* kWave) void masked_xattn_fwd_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    const uint8_t* __restrict__ mask, const int* __restrict__ row_open, float* __restrict__ ws, int Q, int N,
    int heads, int n_splits, int tiles_per_split) {
  constexpr int DK = D / 4;   // k-steps of S^T: lane group g owns d = DK*g .. DK*g+DK-1
  constexpr int DT = D / 16;  // 16-row tiles of O^T
  constexpr int QL = NQT * 16;
  constexpr int RS = D + 4;  // LDS / workspace row: O[D], m, l, pad, pad (keeps float4 alignment)
  __shared__ __attribute__((aligned(16))) float part[kXWaves][QL][RS];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, n = lane & 15;
  const int split = blockIdx.x % n_splits, qc = blockIdx.x / n_splits;
  const int h = blockIdx.y, b = blockIdx.z;
  const int q0 = qc * QL;
  const int E = heads * D;
  const float NEG_INF = -INFINITY;

  // ---- Q^T fragments (B operand), kept for the whole kernel
  float qf[NQT][DK];
  int qrow[NQT];
  bool use_mask[NQT];
#pragma unroll
  for (int j = 0; j < NQT; ++j) {
    int qi = q0 + 16 * j + n;
    if (qi > Q - 1) qi = Q - 1;  // padding rows replay the last query; never stored
    qrow[j] = qi;
    const float* qp = q + ((int64_t)b * Q + qi) * E + h * D + DK * g;
#pragma unroll
    for (int t = 0; t < DK; t += 4) {
      const float4 x = *reinterpret_cast<const float4*>(qp + t);
      qf[j][t] = x.x; qf[j][t + 1] = x.y; qf[j][t + 2] = x.z; qf[j][t + 3] = x.w;
    }
    use_mask[j] = mask != nullptr && (row_open == nullptr || row_open[(int64_t)b * Q + qi] != 0);
  }

  f32x4 o[DT][NQT];
  float m[NQT], l[NQT];
#pragma unroll
  for (int j = 0; j < NQT; ++j) {
    m[j] = NEG_INF;
    l[j] = 0.f;
#pragma unroll
    for (int i = 0; i < DT; ++i) o[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  const int n_tiles = ceil_div(N, 16);
  int t_end = (split + 1) * tiles_per_split;
  if (t_end > n_tiles) t_end = n_tiles;
  const bool n_al4 = (N & 3) == 0;

  // K / V^T fragments of one 16-key tile.  K (A operand of S^T): lane (key = key0+n, d = DK*g + t);
  // V^T (A operand of O^T): lane (d = 16i+n, key = key0+4g+t).  Loaded ONE TILE AHEAD of their use: with two waves
  // per SIMD the global-load latency of a tile was exposed once per tile (no other work to hide it).
  auto load_tile = [&](int tile, float (&kf_)[DK], float (&vf_)[DT][4]) __attribute__((always_inline)) {
    const int key0 = tile * 16;
    int kk = key0 + n;
    if (kk > N - 1) kk = N - 1;
    const float* kp = k + ((int64_t)b * N + kk) * E + h * D + DK * g;
#pragma unroll
    for (int t = 0; t < DK; t += 4) {
      const float4 x = *reinterpret_cast<const float4*>(kp + t);
      kf_[t] = x.x; kf_[t + 1] = x.y; kf_[t + 2] = x.z; kf_[t + 3] = x.w;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      int vk = key0 + 4 * g + t;
      if (vk > N - 1) vk = N - 1;
      const float* vp = v + ((int64_t)b * N + vk) * E + h * D + n;
#pragma unroll
      for (int i = 0; i < DT; ++i) vf_[i][t] = vp[16 * i];
    }
  };
  float kf_next[DK], vf_next[DT][4];
  {
    int t_first = split * tiles_per_split + wave;
    if (t_first > n_tiles - 1) t_first = n_tiles - 1;  // n_tiles >= 1; keeps the loads in range when this wave has no tile
    load_tile(t_first, kf_next, vf_next);
  }
  for (int tile = split * tiles_per_split + wave; tile < t_end; tile += kXWaves) {
    const int key0 = tile * 16;
    float kf[DK], vf[DT][4];
#pragma unroll
    for (int t = 0; t < DK; ++t) kf[t] = kf_next[t];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
      for (int t = 0; t < 4; ++t) vf[i][t] = vf_next[i][t];
    {
      int t_n = tile + kXWaves;
      if (t_n > n_tiles - 1) t_n = n_tiles - 1;  // the last prefetch re-reads a valid tile and is discarded
      load_tile(t_n, kf_next, vf_next);
    }

    // ---- S^T = K Q^T
    f32x4 s[NQT];
#pragma unroll
    for (int j = 0; j < NQT; ++j) {
      s[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < DK; ++t) s[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[t], qf[j][t], s[j], 0, 0, 0);
    }

    // ---- mask, online softmax (per query = per C column), rescale O
#pragma unroll
    for (int j = 0; j < NQT; ++j) {
      uint32_t mb = 0;
      if (use_mask[j]) {
        const uint8_t* mp = mask + ((int64_t)b * Q + qrow[j]) * N + key0 + 4 * g;
        if (n_al4 && key0 + 4 * g + 3 < N) {
          mb = *reinterpret_cast<const uint32_t*>(mp);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (key0 + 4 * g + r < N) mb |= (uint32_t)(mp[r] != 0) << (8 * r);
        }
      }
      float tmax = NEG_INF;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool dead = ((mb >> (8 * r)) & 0xffu) != 0 || (key0 + 4 * g + r >= N);
        s[j][r] = dead ? NEG_INF : s[j][r];
        tmax = fmaxf(tmax, s[j][r]);
      }
      tmax = fmaxf(tmax, __shfl_xor(tmax, 16, kWave));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, kWave));
      const float m_new = fmaxf(m[j], tmax);
      const float m_safe = (m_new == NEG_INF) ? 0.f : m_new;
      const float alpha = __expf(m[j] - m_safe);
      m[j] = m_new;
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __expf(s[j][r] - m_safe);
        s[j][r] = p;
        psum += p;
      }
      l[j] = l[j] * alpha + psum;  // lane-local partial sum; lane groups are merged at the end
#pragma unroll
      for (int i = 0; i < DT; ++i) o[i][j] *= alpha;
    }

    // ---- O^T += V^T P^T
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
      for (int j = 0; j < NQT; ++j)
#pragma unroll
        for (int t = 0; t < 4; ++t) o[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[i][t], s[j][t], o[i][j], 0, 0, 0);
  }

  xattn_merge_waves<NQT, D>(part, o, m, l, 1.f, wave, g, n, ws, b, h, heads, Q, q0, n_splits, split);
}

// Full-tile form of the kernel above for N % 16 == 0 (every feature level of a 32-divisible input): same split,
// same tiles, same merge, but the per-tile VALU work is cut to what the arithmetic needs -- PMC at config 2 showed 7
// VALU instructions per MFMA and 58 % of the wave cycles in s_waitcnt (profiles/r01_pmc_k2.txt):
//   * K, V and the mask are read through buffer descriptors with a per-lane offset that never changes and the tile
//     as a SCALAR offset (all accesses are in range by construction here; the scalar offset is not range-checked);
//   * K, V AND the mask words of tile t+4 are issued before the arithmetic of tile t into a second register set
//     (2x unrolled ping-pong, no copies), so no load is waited for at its use;
//   * the softmax runs in the log2 domain (q is scaled by log2 e once): exp2 without a multiply per element;
//   * the max over the 4 lane groups uses v_permlane16_swap / v_permlane32_swap (VALU) instead of two LDS permutes;
//   * the file is compiled with -amdgpu-mfma-vgpr-form, which keeps S in VGPRs (40 accvgpr moves per tile gone).
template <int NQT, int D>
__global__ __launch_bounds__(kXWaves* kWave) void masked_xattn_fwd_full_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    const uint8_t* __restrict__ mask, const int* __restrict__ row_open, float* __restrict__ ws, int Q, int N,
    int heads, int n_splits, int tiles_per_split) {
  constexpr int DK = D / 4;
  constexpr int DT = D / 16;
  constexpr int QL = NQT * 16;
  constexpr int RS = D + 4;
  constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;
  constexpr float kTau = 8.f;  // lazy-rescale slack, log2 units
  __shared__ __attribute__((aligned(16))) float part[kXWaves][QL][RS];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: tile offsets stay in SGPRs
  const int g = lane >> 4, n = lane & 15;
  const int split = blockIdx.x % n_splits, qc = blockIdx.x / n_splits;
  const int h = blockIdx.y, b = blockIdx.z;
  const int q0 = qc * QL;
  const int E = heads * D;
  const float NEG_INF = -INFINITY;

  float qf[NQT][DK];
  int qrow[NQT];
  bool use_mask[NQT];
#pragma unroll
  for (int j = 0; j < NQT; ++j) {
    int qi = q0 + 16 * j + n;
    if (qi > Q - 1) qi = Q - 1;
    qrow[j] = qi;
    const float* qp = q + ((int64_t)b * Q + qi) * E + h * D + DK * g;
#pragma unroll
    for (int t = 0; t < DK; t += 4) {
      const float4 x = *reinterpret_cast<const float4*>(qp + t);
      qf[j][t] = x.x * kLog2e; qf[j][t + 1] = x.y * kLog2e; qf[j][t + 2] = x.z * kLog2e; qf[j][t + 3] = x.w * kLog2e;
    }
    use_mask[j] = mask != nullptr && (row_open == nullptr || row_open[(int64_t)b * Q + qi] != 0);
  }

  f32x4 o[DT][NQT];
  float m[NQT], l[NQT];
#pragma unroll
  for (int j = 0; j < NQT; ++j) {
    m[j] = NEG_INF;
    l[j] = 0.f;
#pragma unroll
    for (int i = 0; i < DT; ++i) o[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  const int n_tiles = N / 16;
  int t_end = (split + 1) * tiles_per_split;
  if (t_end > n_tiles) t_end = n_tiles;

  const __amdgpu_buffer_rsrc_t k_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)(k + (int64_t)b * N * E), 0, N * E * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t v_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)(v + (int64_t)b * N * E), 0, N * E * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t m_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(mask != nullptr ? mask + (int64_t)b * Q * N : nullptr), 0, mask != nullptr ? Q * N : 0, 0x00020000);
  const uint32_t k_voff = (uint32_t)((n * E + h * D + DK * g) * 4);      // K: lane (key0 + n, d = DK g + t)
  const uint32_t v_voff = (uint32_t)((4 * g * E + h * D + n) * 4);       // V^T: lane (d = 16 i + n, key0 + 4 g + t)
  uint32_t m_voff[NQT];                                                   // mask: 4 bytes (keys key0 + 4 g ..) of row qrow
#pragma unroll
  // a row that does not use the mask (row_open = 0, or no mask at all) reads beyond the descriptor's range: zeros = open
  for (int j = 0; j < NQT; ++j) m_voff[j] = use_mask[j] ? (uint32_t)(qrow[j] * N + 4 * g) : 0x80000000u;
  const int row_bytes = E * 4;

  struct Frag {
    float k[DK];
    float v[DT][4];
    uint32_t mb[NQT];
  };
  auto load_tile = [&](int tile, Frag& f) __attribute__((always_inline)) {
    if (tile > n_tiles - 1) tile = n_tiles - 1;  // the prefetch past the end re-reads a valid tile and is discarded
    const int soff = tile * 16 * row_bytes;
#pragma unroll
    for (int t = 0; t < DK; t += 4) {
      const f32x4 x = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(k_rsrc, k_voff + 4 * t, soff, 0));
      f.k[t] = x[0]; f.k[t + 1] = x[1]; f.k[t + 2] = x[2]; f.k[t + 3] = x[3];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < DT; ++i)
        f.v[i][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(v_rsrc, v_voff + 64 * i, soff + t * row_bytes, 0));
#pragma unroll
    for (int j = 0; j < NQT; ++j) f.mb[j] = __builtin_amdgcn_raw_buffer_load_b32(m_rsrc, m_voff[j], tile * 16, 0);
  };
  // max over the 4 lane groups (rows of 16 lanes) with the gfx950 row swaps: v_permlane16_swap exchanges the odd rows
  // of its first operand with the even rows of its second, v_permlane32_swap the upper half with the lower half.
  // Both results pass through an empty asm: this clang folds fmaxf(r[0], r[1]) of the builtin's pair into r[0]
  // (the v_max disappears from the ISA), which the opaque copies prevent.
  // v_max_f32 without the canonicalising v_max x, x the compiler puts in front of fmaxf on values that come out of a bit
  // cast (4 of the 6 vector instructions of a group maximum were those)
  auto vmax = [](float x, float y) __attribute__((always_inline)) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
  };
  auto group_max = [&](float x) __attribute__((always_inline)) {
    const uint32_t u = __builtin_bit_cast(uint32_t, x);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float y = vmax(__builtin_bit_cast(float, (uint32_t)a[0]), __builtin_bit_cast(float, (uint32_t)a[1]));
    const uint32_t w = __builtin_bit_cast(uint32_t, y);
    const auto c = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return vmax(__builtin_bit_cast(float, (uint32_t)c[0]), __builtin_bit_cast(float, (uint32_t)c[1]));
  };
  auto compute = [&](const Frag& f, bool live) __attribute__((always_inline)) {
    f32x4 s[NQT];
#pragma unroll
    for (int j = 0; j < NQT; ++j) {
      s[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < DK; ++t) s[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.k[t], qf[j][t], s[j], 0, 0, 0);
    }
    float tmax[NQT];
    bool grow = false;
#pragma unroll
    for (int j = 0; j < NQT; ++j) {
      uint32_t mb = f.mb[j];
      asm volatile("" : "+v"(mb));  // ordered after the fences: free, the mask tests of fb were placed in compute(fa) -- behind a vmcnt(0)
      mb = live ? mb : 0xffffffffu;
#pragma unroll
      for (int r = 0; r < 4; ++r) s[j][r] = ((mb >> (8 * r)) & 0xffu) != 0 ? NEG_INF : s[j][r];
      tmax[j] = group_max(fmaxf(fmaxf(s[j][0], s[j][1]), fmaxf(s[j][2], s[j][3])));
      grow |= tmax[j] > m[j] + kTau;
    }
    // Lazy rescale: m[j] is the reference the running (l, O) are scaled by, and it only has to stay within kTau
    // (log2 units) of the true running maximum -- p <= 2^kTau, the quotient O / l is the same number.  The
    // rescale of l and O (1 exp + 9 multiplies per query tile) runs only when some query's tile maximum exceeds
    // its reference by more than that: the first live tile, and rarely afterwards.  ONE wave-uniform branch per
    // key tile (one per query tile serialised the tiles' mask / max chains: 10 s_nop each).  fp32 MFMAs and vector
    // instructions do not execute together on a SIMD (SQ_VALU_MFMA_COEXEC_CYCLES = 0 on this kernel,
    // profiles/r02_pmc_k2.txt), so every vector instruction removed is time removed.
    if (__builtin_amdgcn_ballot_w64(grow) != 0) {
#pragma unroll
      for (int j = 0; j < NQT; ++j) {
        const float m_new = fmaxf(m[j], tmax[j]);
        const float m_s = (m_new == NEG_INF) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m[j] - m_s);
        m[j] = m_new;
        l[j] *= alpha;
#pragma unroll
        for (int i = 0; i < DT; ++i) o[i][j] *= alpha;
      }
    }
#pragma unroll
    for (int j = 0; j < NQT; ++j) {
      const float m_safe = (m[j] == NEG_INF) ? 0.f : m[j];
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f(s[j][r] - m_safe);
        s[j][r] = p;
        psum += p;
      }
      l[j] += psum;
    }
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
      for (int j = 0; j < NQT; ++j)
#pragma unroll
        for (int t = 0; t < 4; ++t) o[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.v[i][t], s[j][t], o[i][j], 0, 0, 0);
    // pin O here: otherwise these MFMAs drift below the next prefetch, the V fragments stay live across it, the
    // prefetch lands in other registers and is copied back at the loop end -- behind a wait for the loads
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
      for (int j = 0; j < NQT; ++j) asm volatile("" : "+v"(o[i][j]));
  };

  Frag fa, fb;
  int tile = split * tiles_per_split + wave;
  load_tile(tile, fa);
  // Two tiles per iteration in ONE basic block, with scheduling fences that keep each prefetch above the arithmetic
  // it overlaps.  With a loop exit between the halves the compiler sank the loads of fb into the block of
  // compute(fb) and waited for them at once (half of the tiles were not prefetched); a wave with an odd tile count
  // now runs its last half on a tile marked all-dead instead (p = 0, alpha = 1: no contribution).
  while (tile < t_end) {
    load_tile(tile + kXWaves, fb);
    __builtin_amdgcn_sched_barrier(0);
    compute(fa, true);
    __builtin_amdgcn_sched_barrier(0);
    load_tile(tile + 2 * kXWaves, fa);
    __builtin_amdgcn_sched_barrier(0);
    compute(fb, tile + kXWaves < t_end);
    __builtin_amdgcn_sched_barrier(0);
    tile += 2 * kXWaves;
  }

  xattn_merge_waves<NQT, D>(part, o, m, l, kLn2, wave, g, n, ws, b, h, heads, Q, q0, n_splits, split);
}

// Query-split form of the kernel above (selectable with WM2F_K2_QSPLIT=1, not the default): the 4 waves of a workgroup walk the SAME
// key tiles of a split and each owns NQT query tiles of its own, so K / V are fetched from L2 once per (image, head,
// split) -- the waves' identical fragment loads meet in the CU's L1 -- while a wave still carries only NQT = 2 tiles
// of state (3 waves per SIMD).  No cross-wave merge: every wave writes its queries' partial (O, m, l) directly.
template <int NQT, int D>
__global__ __launch_bounds__(kXWaves* kWave) void masked_xattn_fwd_qsplit_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    const uint8_t* __restrict__ mask, const int* __restrict__ row_open, float* __restrict__ ws, int Q, int N,
    int heads, int n_splits, int tiles_per_split) {
  constexpr int DK = D / 4;   // k-steps of S^T: lane group g owns d = DK*g .. DK*g+DK-1
  constexpr int DT = D / 16;  // 16-row tiles of O^T
  constexpr int QL = NQT * 16;  // queries of ONE WAVE; a workgroup covers kXWaves * QL
  constexpr int RS = D + 4;     // workspace row: O[D], m, l, pad, pad (keeps float4 alignment)

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, n = lane & 15;
  const int split = blockIdx.x % n_splits, qc = blockIdx.x / n_splits;
  const int h = blockIdx.y, b = blockIdx.z;
  const int q0 = (qc * kXWaves + wave) * QL;
  if (q0 >= Q) return;  // no barrier in this kernel: a wave without queries just leaves
  const int E = heads * D;
  const float NEG_INF = -INFINITY;

  // ---- Q^T fragments (B operand), kept for the whole kernel
  float qf[NQT][DK];
  int qrow[NQT];
  bool use_mask[NQT];
#pragma unroll
  for (int j = 0; j < NQT; ++j) {
    int qi = q0 + 16 * j + n;
    if (qi > Q - 1) qi = Q - 1;  // padding rows replay the last query; never stored
    qrow[j] = qi;
    const float* qp = q + ((int64_t)b * Q + qi) * E + h * D + DK * g;
#pragma unroll
    for (int t = 0; t < DK; t += 4) {
      const float4 x = *reinterpret_cast<const float4*>(qp + t);
      qf[j][t] = x.x; qf[j][t + 1] = x.y; qf[j][t + 2] = x.z; qf[j][t + 3] = x.w;
    }
    use_mask[j] = mask != nullptr && (row_open == nullptr || row_open[(int64_t)b * Q + qi] != 0);
  }

  f32x4 o[DT][NQT];
  float m[NQT], l[NQT];
#pragma unroll
  for (int j = 0; j < NQT; ++j) {
    m[j] = NEG_INF;
    l[j] = 0.f;
#pragma unroll
    for (int i = 0; i < DT; ++i) o[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  const int n_tiles = ceil_div(N, 16);
  int t_end = (split + 1) * tiles_per_split;
  if (t_end > n_tiles) t_end = n_tiles;
  const bool n_al4 = (N & 3) == 0;

  // K / V^T fragments of one 16-key tile.  K (A operand of S^T): lane (key = key0+n, d = DK*g + t);
  // V^T (A operand of O^T): lane (d = 16i+n, key = key0+4g+t).  Loaded ONE TILE AHEAD of their use: with two waves
  // per SIMD the global-load latency of a tile was exposed once per tile (no other work to hide it).
  auto load_tile = [&](int tile, float (&kf_)[DK], float (&vf_)[DT][4]) __attribute__((always_inline)) {
    const int key0 = tile * 16;
    int kk = key0 + n;
    if (kk > N - 1) kk = N - 1;
    const float* kp = k + ((int64_t)b * N + kk) * E + h * D + DK * g;
#pragma unroll
    for (int t = 0; t < DK; t += 4) {
      const float4 x = *reinterpret_cast<const float4*>(kp + t);
      kf_[t] = x.x; kf_[t + 1] = x.y; kf_[t + 2] = x.z; kf_[t + 3] = x.w;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      int vk = key0 + 4 * g + t;
      if (vk > N - 1) vk = N - 1;
      const float* vp = v + ((int64_t)b * N + vk) * E + h * D + n;
#pragma unroll
      for (int i = 0; i < DT; ++i) vf_[i][t] = vp[16 * i];
    }
  };
  float kf_next[DK], vf_next[DT][4];
  {
    int t_first = split * tiles_per_split;
    if (t_first > n_tiles - 1) t_first = n_tiles - 1;  // n_tiles >= 1; keeps the loads in range when this wave has no tile
    load_tile(t_first, kf_next, vf_next);
  }
  for (int tile = split * tiles_per_split; tile < t_end; ++tile) {
    const int key0 = tile * 16;
    float kf[DK], vf[DT][4];
#pragma unroll
    for (int t = 0; t < DK; ++t) kf[t] = kf_next[t];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
      for (int t = 0; t < 4; ++t) vf[i][t] = vf_next[i][t];
    {
      int t_n = tile + 1;
      if (t_n > n_tiles - 1) t_n = n_tiles - 1;  // the last prefetch re-reads a valid tile and is discarded
      load_tile(t_n, kf_next, vf_next);
    }

    // ---- S^T = K Q^T
    f32x4 s[NQT];
#pragma unroll
    for (int j = 0; j < NQT; ++j) {
      s[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < DK; ++t) s[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[t], qf[j][t], s[j], 0, 0, 0);
    }

    // ---- mask, online softmax (per query = per C column), rescale O
#pragma unroll
    for (int j = 0; j < NQT; ++j) {
      uint32_t mb = 0;
      if (use_mask[j]) {
        const uint8_t* mp = mask + ((int64_t)b * Q + qrow[j]) * N + key0 + 4 * g;
        if (n_al4 && key0 + 4 * g + 3 < N) {
          mb = *reinterpret_cast<const uint32_t*>(mp);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (key0 + 4 * g + r < N) mb |= (uint32_t)(mp[r] != 0) << (8 * r);
        }
      }
      float tmax = NEG_INF;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool dead = ((mb >> (8 * r)) & 0xffu) != 0 || (key0 + 4 * g + r >= N);
        s[j][r] = dead ? NEG_INF : s[j][r];
        tmax = fmaxf(tmax, s[j][r]);
      }
      tmax = fmaxf(tmax, __shfl_xor(tmax, 16, kWave));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, kWave));
      const float m_new = fmaxf(m[j], tmax);
      const float m_safe = (m_new == NEG_INF) ? 0.f : m_new;
      const float alpha = __expf(m[j] - m_safe);
      m[j] = m_new;
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __expf(s[j][r] - m_safe);
        s[j][r] = p;
        psum += p;
      }
      l[j] = l[j] * alpha + psum;  // lane-local partial sum; lane groups are merged at the end
#pragma unroll
      for (int i = 0; i < DT; ++i) o[i][j] *= alpha;
    }

    // ---- O^T += V^T P^T
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
      for (int j = 0; j < NQT; ++j)
#pragma unroll
        for (int t = 0; t < 4; ++t) o[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[i][t], s[j][t], o[i][j], 0, 0, 0);
  }

  // ---- each wave owns its queries: straight to the workspace (one partial per key split)
#pragma unroll
  for (int j = 0; j < NQT; ++j) {
    float lt = l[j];
    lt += __shfl_xor(lt, 16, kWave);
    lt += __shfl_xor(lt, 32, kWave);
    const int qi = q0 + 16 * j + n;
    if (qi >= Q) continue;
    float* wrow = ws + ((((int64_t)b * heads + h) * Q + qi) * n_splits + split) * RS;
#pragma unroll
    for (int i = 0; i < DT; ++i) *reinterpret_cast<f32x4*>(wrow + 16 * i + 4 * g) = o[i][j];
    if (g == 0) *reinterpret_cast<float4*>(wrow + D) = make_float4(m[j], lt, 0.f, 0.f);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// bf16 form of the full-tile kernel (BASELINE configs 2-4 run under bf16 autocast, where the dependency's attention products
// are bf16 matrix products with an fp32 softmax between them: baddbmm / softmax / bmm at TORCHF:6578-6600).
// q, k, v arrive in bf16 AS THE in_proj LINEARS EMIT THEM -- (B, Q | N, heads * 32), no cast pass, half the K / V bytes;
// both products run on v_mfma_f32_16x16x16_bf16 with fp32 accumulation; S, the online softmax, (m, l) and O stay fp32 (the
// dependency rounds S to bf16 before its softmax; this kernel does not); P is rounded to bf16 for the second product as the
// dependency's bmm does.  Same split over keys, same (O, m, l) partials, same merge kernels as the fp32 form.
//   S^T = K Q^T: A = K tile, lane (key n, group g) holds K[key][8 g .. 8 g + 7] -- ONE 16-byte load; its elements 0..3 feed MFMA
//        step 0 and 4..7 step 1, i.e. the contraction index of (step s, group g, element i) is d = 8 g + 4 s + i, and Q^T is held
//        in the same order.  C layout: column = query (lane & 15), rows = keys 4 g + r -- directly the B operand of
//   O^T += V^T P^T: the A operand wants V TRANSPOSED (4 consecutive keys per lane at a fixed d).  V stays in its natural layout in
//        HBM: lane (key n, group g) fetches V[key][8 g .. + 7] (16 bytes), the wave writes the 16 x 32 tile row-major into 1 KiB
//        of its own LDS and reads it back with ds_read_b64_tr_b16 (4 keys x 16 d per 16-lane group, delivered column-major).
template <int NQT>
__global__ __launch_bounds__(kXWaves* kWave) void masked_xattn_bf16_fwd_full_kernel(
    const uint16_t* __restrict__ q, const uint16_t* __restrict__ k, const uint16_t* __restrict__ v,
    const uint8_t* __restrict__ mask, const int* __restrict__ row_open, float* __restrict__ ws, int Q, int N,
    int heads, int n_splits, int tiles_per_split) {
  constexpr int D = 32, DT = 2, QL = NQT * 16, RS = D + 4;
  constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;
  constexpr float kTau = 8.f;
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4_t;
  __shared__ __attribute__((aligned(16))) float part[kXWaves][QL][RS];
  __shared__ __attribute__((aligned(16))) uint16_t vtile[kXWaves][16 * D];  // a wave's V tile, [key][d] row-major

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, n = lane & 15;
  const int split = blockIdx.x % n_splits, qc = blockIdx.x / n_splits;
  const int h = blockIdx.y, b = blockIdx.z;
  const int q0 = qc * QL;
  const int E = heads * D;
  const float NEG_INF = -INFINITY;

  s16x4 qf[NQT][2];
  int qrow[NQT];
  bool use_mask[NQT];
#pragma unroll
  for (int j = 0; j < NQT; ++j) {
    int qi = q0 + 16 * j + n;
    if (qi > Q - 1) qi = Q - 1;
    qrow[j] = qi;
    const s16x8 x = *reinterpret_cast<const s16x8*>(q + ((int64_t)b * Q + qi) * E + h * D + 8 * g);
    qf[j][0] = __builtin_shufflevector(x, x, 0, 1, 2, 3);
    qf[j][1] = __builtin_shufflevector(x, x, 4, 5, 6, 7);
    use_mask[j] = mask != nullptr && (row_open == nullptr || row_open[(int64_t)b * Q + qi] != 0);
  }

  f32x4 o[DT][NQT];
  float m[NQT], l[NQT];  // m: the reference maximum in LOG2 units (of s * log2 e)
#pragma unroll
  for (int j = 0; j < NQT; ++j) {
    m[j] = NEG_INF;
    l[j] = 0.f;
#pragma unroll
    for (int i = 0; i < DT; ++i) o[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  const int n_tiles = N / 16;
  int t_end = (split + 1) * tiles_per_split;
  if (t_end > n_tiles) t_end = n_tiles;

  const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(k + (int64_t)b * N * E), 0, N * E * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(v + (int64_t)b * N * E), 0, N * E * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t m_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(mask != nullptr ? mask + (int64_t)b * Q * N : nullptr), 0, mask != nullptr ? Q * N : 0, 0x00020000);
  const uint32_t kv_voff = (uint32_t)((n * E + h * D + 8 * g) * 2);  // K and V alike: lane (key0 + n, d = 8 g ..)
  uint32_t m_voff[NQT];
#pragma unroll
  for (int j = 0; j < NQT; ++j) m_voff[j] = use_mask[j] ? (uint32_t)(qrow[j] * N + 4 * g) : 0x80000000u;
  const int row_bytes = E * 2;

  struct Frag {
    s16x8 k, v;
    uint32_t mb[NQT];
  };
  auto load_tile = [&](int tile, Frag& f) __attribute__((always_inline)) {
    if (tile > n_tiles - 1) tile = n_tiles - 1;
    const int soff = tile * 16 * row_bytes;
    f.k = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(k_rsrc, kv_voff, soff, 0));
    f.v = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(v_rsrc, kv_voff, soff, 0));
#pragma unroll
    for (int j = 0; j < NQT; ++j) f.mb[j] = __builtin_amdgcn_raw_buffer_load_b32(m_rsrc, m_voff[j], tile * 16, 0);
  };
  auto vmax = [](float x, float y) __attribute__((always_inline)) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
  };
  auto group_max = [&](float x) __attribute__((always_inline)) {
    const uint32_t u = __builtin_bit_cast(uint32_t, x);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float y = vmax(__builtin_bit_cast(float, (uint32_t)a[0]), __builtin_bit_cast(float, (uint32_t)a[1]));
    const uint32_t w = __builtin_bit_cast(uint32_t, y);
    const auto c = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return vmax(__builtin_bit_cast(float, (uint32_t)c[0]), __builtin_bit_cast(float, (uint32_t)c[1]));
  };
  // this wave's V tile in LDS: written row-major (lane (n, g): 16 bytes at row n, byte 16 g), read transposed
  uint16_t* vt = &vtile[wave][0];
  const int vt_wr = n * (D * 2) + g * 16;
  const int tq = (lane >> 2) & 3, tp = lane & 3;
  const int vt_rd = (4 * g + tq) * (D * 2) + tp * 8;  // + 32 i for d tile i
  auto compute = [&](const Frag& f, bool live) __attribute__((always_inline)) {
    asm volatile("" ::: "memory");
    *reinterpret_cast<s16x8*>(reinterpret_cast<unsigned char*>(vt) + vt_wr) = f.v;
    asm volatile("" ::: "memory");
    const s16x4 k0 = __builtin_shufflevector(f.k, f.k, 0, 1, 2, 3), k1 = __builtin_shufflevector(f.k, f.k, 4, 5, 6, 7);
    f32x4 s[NQT];
#pragma unroll
    for (int j = 0; j < NQT; ++j) {
      s[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      s[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(k0, qf[j][0], s[j], 0, 0, 0);
      s[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(k1, qf[j][1], s[j], 0, 0, 0);
    }
    s16x4 va[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i)
      va[i] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(reinterpret_cast<unsigned char*>(vt) + vt_rd + 32 * i));
    asm volatile("" ::: "memory");
    float tmax[NQT];
    bool grow = false;
#pragma unroll
    for (int j = 0; j < NQT; ++j) {
      uint32_t mb = f.mb[j];
      asm volatile("" : "+v"(mb));
      mb = live ? mb : 0xffffffffu;
#pragma unroll
      for (int r = 0; r < 4; ++r) s[j][r] = ((mb >> (8 * r)) & 0xffu) != 0 ? NEG_INF : s[j][r] * kLog2e;
      tmax[j] = group_max(fmaxf(fmaxf(s[j][0], s[j][1]), fmaxf(s[j][2], s[j][3])));
      grow |= tmax[j] > m[j] + kTau;
    }
    if (__builtin_amdgcn_ballot_w64(grow) != 0) {
#pragma unroll
      for (int j = 0; j < NQT; ++j) {
        const float m_new = fmaxf(m[j], tmax[j]);
        const float m_s = (m_new == NEG_INF) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m[j] - m_s);
        m[j] = m_new;
        l[j] *= alpha;
#pragma unroll
        for (int i = 0; i < DT; ++i) o[i][j] *= alpha;
      }
    }
    s16x4 pb[NQT];
#pragma unroll
    for (int j = 0; j < NQT; ++j) {
      const float m_safe = (m[j] == NEG_INF) ? 0.f : m[j];
      float psum = 0.f;
      bf16x4_t pk;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f(s[j][r] - m_safe);
        pk[r] = (__bf16)p;
        psum += (float)pk[r];  // l sums what the second product really multiplies by: the rounded p
      }
      l[j] += psum;
      pb[j] = __builtin_bit_cast(s16x4, pk);
    }
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
      for (int j = 0; j < NQT; ++j) o[i][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(va[i], pb[j], o[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
      for (int j = 0; j < NQT; ++j) asm volatile("" : "+v"(o[i][j]));
  };

  Frag fa, fb;
  int tile = split * tiles_per_split + wave;
  load_tile(tile, fa);
  while (tile < t_end) {
    load_tile(tile + kXWaves, fb);
    __builtin_amdgcn_sched_barrier(0);
    compute(fa, true);
    __builtin_amdgcn_sched_barrier(0);
    load_tile(tile + 2 * kXWaves, fa);
    __builtin_amdgcn_sched_barrier(0);
    compute(fb, tile + kXWaves < t_end);
    __builtin_amdgcn_sched_barrier(0);
    tile += 2 * kXWaves;
  }

  xattn_merge_waves<NQT, D>(part, o, m, l, kLn2, wave, g, n, ws, b, h, heads, Q, q0, n_splits, split);
}

// Merge the per-split partials: one thread per (b, h, q, float4 chunk of D).
template <int D>
__global__ __launch_bounds__(256) void masked_xattn_merge_kernel(const float* __restrict__ ws,
                                                                 float* __restrict__ out, float* __restrict__ lse,
                                                                 int B, int heads, int Q, int n_splits) {
  constexpr int RS = D + 4, CH = D / 4;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)B * heads * Q * CH;
  if (idx >= total) return;
  const int c = (int)(idx % CH);
  const int64_t row = idx / CH;  // (b*heads + h)*Q + q
  const int qi = (int)(row % Q);
  const int64_t bh = row / Q;
  const int h = (int)(bh % heads), b = (int)(bh / heads);
  const float* base = ws + row * n_splits * RS;
  float M = -INFINITY;
  for (int s = 0; s < n_splits; ++s) M = fmaxf(M, base[s * RS + D]);
  const float Ms = (M == -INFINITY) ? 0.f : M;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float L = 0.f;
  for (int s = 0; s < n_splits; ++s) {
    const float sc = __expf(base[s * RS + D] - Ms);
    const float4 x = *reinterpret_cast<const float4*>(base + s * RS + 4 * c);
    acc.x += sc * x.x; acc.y += sc * x.y; acc.z += sc * x.z; acc.w += sc * x.w;
    L += sc * base[s * RS + D + 1];
  }
  const float inv = 1.f / L;
  *reinterpret_cast<float4*>(out + ((int64_t)b * Q + qi) * heads * D + h * D + 4 * c) =
      make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
  if (lse != nullptr && c == 0) lse[row] = M + logf(L);
}

}  // namespace wm2f

using namespace wm2f;

extern "C" int64_t wm2f_masked_xattn_workspace(int B, int heads, int Q, int N, int D) {
  if (B <= 0 || heads <= 0 || Q <= 0 || N <= 0 || D <= 0) return 0;
  return (int64_t)B * heads * Q * xattn_splits(B, heads, N) * (D + 4) * 4;
}

extern "C" int wm2f_masked_xattn_fwd(const void* q, const void* k, const void* v, const void* mask,
                                     const void* row_open, void* out, void* lse, void* workspace, int B, int heads,
                                     int Q, int N, int D, int dtype, void* stream) {
  const char* who = "wm2f_masked_xattn_fwd";
  WM2F_REQUIRE(dtype == WM2F_F32, "%s: only WM2F_F32 is built", who);
  WM2F_REQUIRE(q && k && v && out && workspace, "%s: null pointer", who);
  WM2F_REQUIRE(B > 0 && heads > 0 && Q > 0 && N > 0, "%s: non-positive size", who);
  WM2F_REQUIRE(D == 16 || D == 32 || D == 64, "%s: head_dim %d not in {16,32,64}", who, D);
  WM2F_REQUIRE(heads <= 65535 && B <= 65535, "%s: heads / B exceed the grid limits", who);
  const int n_splits = xattn_splits(B, heads, N);
  const int n_tiles = ceil_div(N, 16);
  const int tps = ceil_div(n_tiles, n_splits);
  // query row tiles per workgroup: 2 (see xattn_splits; D = 64: at most 3 by register budget)
  const int cap = (D == 64) ? 2 : tune_env("WM2F_K2_QTILES", 7);
  const int q_tiles = ceil_div(Q, 16);
  const int q_chunks = ceil_div(q_tiles, cap);
  int nqt = ceil_div(q_tiles, q_chunks);
  hipStream_t st = (hipStream_t)stream;
  dim3 block(kXWaves * kWave);
  // full-tile kernel: every key tile whole, 32-bit byte offsets, dword-aligned mask rows
  const bool full = (N % 16) == 0 && (int64_t)N * heads * D * 4 < (int64_t(1) << 31) && (int64_t)Q * N < (int64_t(1) << 31) &&
                    (reinterpret_cast<uintptr_t>(mask) & 3) == 0 && tune_env("WM2F_K2_FULL", 1) != 0;
#define WM2F_XL(NQTv, Dv)                                                                                      \
  {                                                                                                            \
    dim3 grid(n_splits* ceil_div(Q, NQTv * 16), heads, B);                                                     \
    if (full)                                                                                                  \
      hipLaunchKernelGGL((masked_xattn_fwd_full_kernel<NQTv, Dv>), grid, block, 0, st, (const float*)q,        \
                         (const float*)k, (const float*)v, (const uint8_t*)mask, (const int*)row_open,         \
                         (float*)workspace, Q, N, heads, n_splits, tps);                                       \
    else                                                                                                       \
      hipLaunchKernelGGL((masked_xattn_fwd_kernel<NQTv, Dv>), grid, block, 0, st, (const float*)q,             \
                         (const float*)k, (const float*)v, (const uint8_t*)mask, (const int*)row_open,         \
                         (float*)workspace, Q, N, heads, n_splits, tps);                                       \
  }
#define WM2F_XD(Dv)                                  \
  if (nqt <= 1) WM2F_XL(1, Dv)                       \
  else if (nqt <= 2) WM2F_XL(2, Dv)                  \
  else if (nqt <= 4) WM2F_XL(4, Dv)                  \
  else WM2F_XL(7, Dv)
  // query-split kernel: a workgroup covers kXWaves * NQT * 16 queries
#define WM2F_XQ(NQTv, Dv)                                                                                      \
  {                                                                                                            \
    dim3 grid(n_splits* ceil_div(Q, kXWaves * NQTv * 16), heads, B);                                           \
    hipLaunchKernelGGL((masked_xattn_fwd_qsplit_kernel<NQTv, Dv>), grid, block, 0, st, (const float*)q,        \
                       (const float*)k, (const float*)v, (const uint8_t*)mask, (const int*)row_open,           \
                       (float*)workspace, Q, N, heads, n_splits, tps);                                         \
  }
  // measured at config 2: 30 / 77 / 263 us, i.e. no better than the key-split form with 2 query tiles (33 / 75 / 258):
  // the L2 re-read was not the limit, the serial MFMA -> softmax -> MFMA chain inside a wave is.  Kept selectable.
  const bool qsplit = D <= 32 && tune_env("WM2F_K2_QSPLIT", 0) != 0;
  if (qsplit) {
    const int per_wave = ceil_div(q_tiles, kXWaves);  // query tiles a wave needs to cover Q with one workgroup
    if (D == 16) {
      if (per_wave <= 1) WM2F_XQ(1, 16) else WM2F_XQ(2, 16)
    } else {
      if (per_wave <= 1) WM2F_XQ(1, 32) else WM2F_XQ(2, 32)
    }
  } else if (D == 16) {
    WM2F_XD(16)
  } else if (D == 32) {
    WM2F_XD(32)
  } else {
    if (nqt <= 1) WM2F_XL(1, 64)
    else if (nqt <= 2) WM2F_XL(2, 64)
    else WM2F_XL(3, 64)
  }
#undef WM2F_XQ
#undef WM2F_XD
#undef WM2F_XL
  WM2F_CHECK_LAUNCH(who);
  const int64_t total = (int64_t)B * heads * Q * (D / 4);
  dim3 mgrid((unsigned)ceil_div64(total, 256));
  if (D == 16)
    hipLaunchKernelGGL((masked_xattn_merge_kernel<16>), mgrid, dim3(256), 0, st, (const float*)workspace,
                       (float*)out, (float*)lse, B, heads, Q, n_splits);
  else if (D == 32)
    hipLaunchKernelGGL((masked_xattn_merge_kernel<32>), mgrid, dim3(256), 0, st, (const float*)workspace,
                       (float*)out, (float*)lse, B, heads, Q, n_splits);
  else
    hipLaunchKernelGGL((masked_xattn_merge_kernel<64>), mgrid, dim3(256), 0, st, (const float*)workspace,
                       (float*)out, (float*)lse, B, heads, Q, n_splits);
  WM2F_CHECK_LAUNCH(who);
  return WM2F_OK;
}

/* bf16 operands (include/wm2f.h): q, k, v bf16, out / lse fp32.  Full-tile form only (N % 16 == 0, D = 32). */
extern "C" int wm2f_masked_xattn_bf16_fwd(const void* q, const void* k, const void* v, const void* mask, const void* row_open,
                                          void* out, void* lse, void* workspace, int B, int heads, int Q, int N, int D,
                                          void* stream) {
  const char* who = "wm2f_masked_xattn_bf16_fwd";
  WM2F_REQUIRE(q && k && v && out && workspace, "%s: null pointer", who);
  WM2F_REQUIRE(B > 0 && heads > 0 && Q > 0 && N > 0, "%s: non-positive size", who);
  WM2F_REQUIRE(heads <= 65535 && B <= 65535, "%s: heads / B exceed the grid limits", who);
  if (D != 32 || (N % 16) != 0 || (int64_t)N * heads * D * 2 >= (int64_t(1) << 31) || (int64_t)Q * N >= (int64_t(1) << 31) ||
      (reinterpret_cast<uintptr_t>(mask) & 3) != 0 || ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) |
                                                        reinterpret_cast<uintptr_t>(v)) & 15) != 0) {
    set_error("%s: needs head_dim 32, N %% 16 == 0, 16-byte aligned operands and 32-bit offsets (cast to fp32 and use "
              "wm2f_masked_xattn_fwd)", who);
    return WM2F_EUNSUPPORTED;
  }
  const int n_splits = xattn_splits(B, heads, N);
  const int n_tiles = N / 16;
  const int tps = ceil_div(n_tiles, n_splits);
  const int q_tiles = ceil_div(Q, 16);
  const int q_chunks = ceil_div(q_tiles, 7);
  const int nqt = ceil_div(q_tiles, q_chunks);
  hipStream_t st = (hipStream_t)stream;
  dim3 block(kXWaves * kWave);
#define WM2F_XB(NQTv)                                                                                                  \
  hipLaunchKernelGGL((masked_xattn_bf16_fwd_full_kernel<NQTv>), dim3(n_splits* ceil_div(Q, NQTv * 16), heads, B), block, 0, \
                     st, (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v, (const uint8_t*)mask,              \
                     (const int*)row_open, (float*)workspace, Q, N, heads, n_splits, tps)
  if (nqt <= 1) WM2F_XB(1);
  else if (nqt <= 2) WM2F_XB(2);
  else if (nqt <= 4) WM2F_XB(4);
  else WM2F_XB(7);
#undef WM2F_XB
  WM2F_CHECK_LAUNCH(who);
  const int64_t total = (int64_t)B * heads * Q * (D / 4);
  hipLaunchKernelGGL((masked_xattn_merge_kernel<32>), dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, st,
                     (const float*)workspace, (float*)out, (float*)lse, B, heads, Q, n_splits);
  WM2F_CHECK_LAUNCH(who);
  return WM2F_OK;
}

// =====================================================================================================
// Backward.  dV = P^T dO;  dP = dO V^T;  dS = P o (dP - delta), delta[q] = sum_d dO[q][d] O[q][d];
// dQ = dS K;  dK = dS^T Q.   P is recomputed from q, k and the forward's log-sum-exp.
//
// Orientation (the opposite of the forward): S and dP are computed with the KEY on the lane column
// (S = Q K^T, C layout: column = key, rows = queries 4g+r), so that P and dS are directly the B operands
// of   dV^T[d][key] += dO^T[d][q] P[q][key]   and   dK^T[d][key] += Q^T[d][q] dS[q][key],
// whose accumulators (column = key, rows = d) store straight into dV / dK rows as float4.  Only dQ needs
// dS with the query on the lane: one 16x16 tile per step crosses a wave-private LDS tile.
// A wave owns whole 16-key tiles -> dK, dV need no reduction; dQ partials are merged over the 4 waves
// through LDS, over the key splits by a small second kernel.
namespace wm2f {

template <int D>
__global__ __launch_bounds__(256) void xattn_delta_kernel(const float* __restrict__ out, const float* __restrict__ go,
                                                          float* __restrict__ delta, int B, int heads, int Q) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (b, h, q)
  if (idx >= (int64_t)B * heads * Q) return;
  const int qi = (int)(idx % Q);
  const int64_t bh = idx / Q;
  const int h = (int)(bh % heads), b = (int)(bh / heads);
  const float* o = out + ((int64_t)b * Q + qi) * heads * D + h * D;
  const float* g = go + ((int64_t)b * Q + qi) * heads * D + h * D;
  float s = 0.f;
#pragma unroll
  for (int d = 0; d < D; d += 4) {
    const float4 a = *reinterpret_cast<const float4*>(o + d), c = *reinterpret_cast<const float4*>(g + d);
    s += a.x * c.x + a.y * c.y + a.z * c.z + a.w * c.w;
  }
  delta[idx] = s;
}

template <int NQT, int D>
__global__ __launch_bounds__(kXWaves* kWave) void masked_xattn_bwd_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    const uint8_t* __restrict__ mask, const int* __restrict__ row_open, const float* __restrict__ lse,
    const float* __restrict__ delta, const float* __restrict__ go, float* __restrict__ dq_ws,
    float* __restrict__ dk, float* __restrict__ dv, int Q, int N, int heads, int n_splits, int tiles_per_split,
    int accumulate) {
  constexpr int DK = D / 4;   // k-steps of S / dP: lane group g owns d = DK*g .. DK*g+DK-1
  constexpr int DT = D / 16;  // 16-wide d tiles
  constexpr int QL = NQT * 16;
  constexpr int TS = 20;  // padded row of the dS transpose tile (keeps float4 alignment)
  __shared__ __attribute__((aligned(16))) float tr[kXWaves][16][TS];
  __shared__ __attribute__((aligned(16))) float dq_part[kXWaves][QL][D];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, n = lane & 15;
  const int split = blockIdx.x % n_splits, qc = blockIdx.x / n_splits;
  const int h = blockIdx.y, b = blockIdx.z;
  const int q0 = qc * QL;
  const int E = heads * D;
  const float* qb = q + (int64_t)b * Q * E + h * D;
  const float* gob = go + (int64_t)b * Q * E + h * D;
  const float* kb = k + (int64_t)b * N * E + h * D;
  const float* vb = v + (int64_t)b * N * E + h * D;
  const float* lseb = lse + ((int64_t)b * heads + h) * Q;
  const float* delb = delta + ((int64_t)b * heads + h) * Q;

  f32x4 dqa[NQT][DT];
#pragma unroll
  for (int jq = 0; jq < NQT; ++jq)
#pragma unroll
    for (int i = 0; i < DT; ++i) dqa[jq][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int n_tiles = ceil_div(N, 16);
  int t_end = (split + 1) * tiles_per_split;
  if (t_end > n_tiles) t_end = n_tiles;

  for (int tile = split * tiles_per_split + wave; tile < t_end; tile += kXWaves) {
    const int key0 = tile * 16;
    int kk = key0 + n;
    const bool key_ok = kk < N;
    if (!key_ok) kk = N - 1;
    // B operands of S and dP: lane (key = key0+n, d = DK*g + t)
    float kf[DK], vf[DK];
#pragma unroll
    for (int t = 0; t < DK; t += 4) {
      const float4 x = *reinterpret_cast<const float4*>(kb + (int64_t)kk * E + DK * g + t);
      const float4 y = *reinterpret_cast<const float4*>(vb + (int64_t)kk * E + DK * g + t);
      kf[t] = x.x; kf[t + 1] = x.y; kf[t + 2] = x.z; kf[t + 3] = x.w;
      vf[t] = y.x; vf[t + 1] = y.y; vf[t + 2] = y.z; vf[t + 3] = y.w;
    }
    // B operand of dQ: lane (key = key0 + 4g + t, d = 16i + n)
    float kr[DT][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      int k2 = key0 + 4 * g + t;
      if (k2 > N - 1) k2 = N - 1;
#pragma unroll
      for (int i = 0; i < DT; ++i) kr[i][t] = kb[(int64_t)k2 * E + 16 * i + n];
    }
    f32x4 dvt[DT], dkt[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i) dvt[i] = dkt[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int jq = 0; jq < NQT; ++jq) {
      // A operands of S and dP: lane (q = 16jq + n, d = DK*g + t)
      int qa = q0 + 16 * jq + n;
      if (qa > Q - 1) qa = Q - 1;
      float qf[DK], gf[DK];
#pragma unroll
      for (int t = 0; t < DK; t += 4) {
        const float4 x = *reinterpret_cast<const float4*>(qb + (int64_t)qa * E + DK * g + t);
        const float4 y = *reinterpret_cast<const float4*>(gob + (int64_t)qa * E + DK * g + t);
        qf[t] = x.x; qf[t + 1] = x.y; qf[t + 2] = x.z; qf[t + 3] = x.w;
        gf[t] = y.x; gf[t + 1] = y.y; gf[t + 2] = y.z; gf[t + 3] = y.w;
      }
      f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < DK; ++t) {
        s = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[t], kf[t], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_16x16x4f32(gf[t], vf[t], dp, 0, 0, 0);
      }
      // rows of this lane: q = q0 + 16jq + 4g + r
      f32x4 p, ds;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qi = q0 + 16 * jq + 4 * g + r;
        const bool q_ok = qi < Q;
        const int qc2 = q_ok ? qi : Q - 1;
        bool dead = !q_ok || !key_ok;
        if (mask != nullptr && (row_open == nullptr || row_open[(int64_t)b * Q + qc2] != 0))
          dead = dead || (mask[((int64_t)b * Q + qc2) * N + kk] != 0);
        const float pv = dead ? 0.f : __expf(s[r] - lseb[qc2]);
        p[r] = pv;
        ds[r] = pv * (dp[r] - delb[qc2]);
      }
      // A operands of dV^T / dK^T: lane (d = 16i + n, q = 16jq + 4g + t)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        int q2 = q0 + 16 * jq + 4 * g + t;
        if (q2 > Q - 1) q2 = Q - 1;  // p, ds are zero there
#pragma unroll
        for (int i = 0; i < DT; ++i) {
          const float got = gob[(int64_t)q2 * E + 16 * i + n];
          const float qt = qb[(int64_t)q2 * E + 16 * i + n];
          dvt[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(got, p[t], dvt[i], 0, 0, 0);
          dkt[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(qt, ds[t], dkt[i], 0, 0, 0);
        }
      }
      // dS to the query-on-lane layout through the wave's LDS tile, then dQ += dS K
#pragma unroll
      for (int r = 0; r < 4; ++r) tr[wave][4 * g + r][n] = ds[r];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const float4 dst4 = *reinterpret_cast<const float4*>(&tr[wave][n][4 * g]);  // lane (q = n, keys 4g..4g+3)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const float dsa[4] = {dst4.x, dst4.y, dst4.z, dst4.w};
      // jq is wave-uniform but a runtime value: select the accumulator with static indices so that
      // dqa stays in registers (a runtime-indexed vector array would go to scratch)
#pragma unroll
      for (int J = 0; J < NQT; ++J)
        if (jq == J) {
#pragma unroll
          for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t)
              dqa[J][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(dsa[t], kr[i][t], dqa[J][i], 0, 0, 0);
        }
    }
    // dV^T / dK^T tiles: column = key n, rows d = 16i + 4g + r  ->  float4 along d
    if (key_ok) {
#pragma unroll
      for (int i = 0; i < DT; ++i) {
        float* pv = dv + ((int64_t)b * N + kk) * E + h * D + 16 * i + 4 * g;
        float* pk = dk + ((int64_t)b * N + kk) * E + h * D + 16 * i + 4 * g;
        if (!accumulate) {  // one query chunk covers Q: this workgroup is the only writer of its keys
          *reinterpret_cast<f32x4*>(pv) = dvt[i];
          *reinterpret_cast<f32x4*>(pk) = dkt[i];
        } else {  // several query chunks (Q > NQT * 16) add into the zero-initialised gradients
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            atomicAdd(pv + r, dvt[i][r]);
            atomicAdd(pk + r, dkt[i][r]);
          }
        }
      }
    }
  }

  // ---- merge dQ over the 4 waves: C layout of dqa is (column = d n, rows q = 4g + r)
#pragma unroll
  for (int jq = 0; jq < NQT; ++jq)
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) dq_part[wave][16 * jq + 4 * g + r][16 * i + n] = dqa[jq][i][r];
  __syncthreads();
  for (int idx = threadIdx.x; idx < QL * (D / 4); idx += kXWaves * kWave) {
    const int ql = idx / (D / 4), c = idx - ql * (D / 4);
    const int qi = q0 + ql;
    if (qi >= Q) continue;
    float4 a = *reinterpret_cast<const float4*>(&dq_part[0][ql][4 * c]);
#pragma unroll
    for (int w = 1; w < kXWaves; ++w) {
      const float4 x = *reinterpret_cast<const float4*>(&dq_part[w][ql][4 * c]);
      a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
    }
    *reinterpret_cast<float4*>(dq_ws + ((((int64_t)b * heads + h) * Q + qi) * n_splits + split) * D + 4 * c) = a;
  }
}

// Full-tile form of the backward (N % 16 == 0, one query chunk covers Q, D = 32): what the general kernel above fetched
// from global memory inside its (key tile, query tile) loop -- the q / grad_out rows as A operands in both orientations
// (~24 loads), lse, delta and row_open of the lane's four rows (12 loads), four mask bytes -- ~40 memory instructions beside
// 40 MFMAs, none of them prefetched (the loop was kept rolled) -- comes from LDS here: q, grad_out, (lse, delta) of the
// workgroup's <= 112 queries are staged ONCE; K, V (both orientations) and the mask bytes of key tile t + 4 are requested
// through buffer descriptors before the arithmetic of tile t, into a second register set; the query-tile loop is unrolled.
// Rows beyond Q carry lse = +inf (p = 0 without a branch); rows that do not use the mask read it through an out-of-range
// offset (zeros).  dQ partials of the four waves meet in LDS that re-uses the staging area.
template <int NQT>
__global__ __launch_bounds__(kXWaves* kWave) void masked_xattn_bwd_full_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    const uint8_t* __restrict__ mask, const int* __restrict__ row_open, const float* __restrict__ lse,
    const float* __restrict__ delta, const float* __restrict__ go, float* __restrict__ dq_ws,
    float* __restrict__ dk, float* __restrict__ dv, int Q, int N, int heads, int n_splits, int tiles_per_split) {
  constexpr int D = 32, DK = 8, DT = 2, QL = NQT * 16, RS = D + 4, TS = 20;
  constexpr int kStage = 2 * QL * RS + 3 * QL, kPart = kXWaves * QL * D;
  __shared__ __attribute__((aligned(16))) float lds[(kStage > kPart ? kStage : kPart)];
  __shared__ __attribute__((aligned(16))) float tr[kXWaves][16][TS];
  float* qs = lds;                  // [QL][RS]
  float* gs = lds + QL * RS;        // [QL][RS]
  float* ld2 = lds + 2 * QL * RS;   // [QL][2] = (lse, delta)
  uint32_t* moff = reinterpret_cast<uint32_t*>(lds + 2 * QL * RS + 2 * QL);  // [QL] byte offset of the row in the mask, or out of range

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, n = lane & 15;
  const int split = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int E = heads * D;
  const float* qb = q + (int64_t)b * Q * E + h * D;
  const float* gob = go + (int64_t)b * Q * E + h * D;
  for (int i = tid; i < QL * (D / 4); i += kXWaves * kWave) {
    const int r = i / (D / 4), c = i - r * (D / 4);
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f), y = x;
    if (r < Q) {
      x = *reinterpret_cast<const float4*>(qb + (int64_t)r * E + 4 * c);
      y = *reinterpret_cast<const float4*>(gob + (int64_t)r * E + 4 * c);
    }
    *reinterpret_cast<float4*>(qs + r * RS + 4 * c) = x;
    *reinterpret_cast<float4*>(gs + r * RS + 4 * c) = y;
  }
  for (int r = tid; r < QL; r += kXWaves * kWave) {
    const bool ok = r < Q;
    ld2[2 * r] = ok ? lse[((int64_t)b * heads + h) * Q + r] : INFINITY;  // p = exp(s - inf) = 0 for padding rows
    ld2[2 * r + 1] = ok ? delta[((int64_t)b * heads + h) * Q + r] : 0.f;
    // a row that does not use the mask (or a padding row) reads it out of range = 0 = open
    const bool use = mask != nullptr && ok && (row_open == nullptr || row_open[(int64_t)b * Q + r] != 0);
    moff[r] = use ? (uint32_t)(r * N) : 0x80000000u;
  }
  __syncthreads();

  const int n_tiles = N / 16;
  int t_end = (split + 1) * tiles_per_split;
  if (t_end > n_tiles) t_end = n_tiles;
  const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(k + (int64_t)b * N * E), 0, N * E * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(v + (int64_t)b * N * E), 0, N * E * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t m_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(mask != nullptr ? mask + (int64_t)b * Q * N : nullptr), 0, mask != nullptr ? Q * N : 0, 0x00020000);
  const uint32_t kv_voff = (uint32_t)((n * E + h * D + DK * g) * 4);   // lane (key0 + n, d = 8 g + t)
  const uint32_t kr_voff = (uint32_t)((4 * g * E + h * D + n) * 4);    // lane (key0 + 4 g + t, d = 16 i + n)
  const int row_bytes = E * 4;
  struct Frag {
    f32x4 kf[2], vf[2];  // B operands of S / dP
    float kr[DT][4];     // B operand of dQ
    uint32_t mb[4];      // mask bytes of query tile 0 (the later query tiles' bytes are requested one tile ahead inside compute)
  };
  auto load_tile = [&](int tile, Frag& f) __attribute__((always_inline)) {
    if (tile > n_tiles - 1) tile = n_tiles - 1;  // the prefetch past the end re-reads a valid tile and is discarded
    const int soff = tile * 16 * row_bytes;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      f.kf[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(k_rsrc, kv_voff + 16 * c, soff, 0));
      f.vf[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(v_rsrc, kv_voff + 16 * c, soff, 0));
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < DT; ++i)
        f.kr[i][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(k_rsrc, kr_voff + 64 * i, soff + t * row_bytes, 0));
#pragma unroll
    for (int r = 0; r < 4; ++r) f.mb[r] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(m_rsrc, moff[4 * g + r] + (uint32_t)n, tile * 16, 0);
  };

  f32x4 dqa[NQT][DT];
#pragma unroll
  for (int jq = 0; jq < NQT; ++jq)
#pragma unroll
    for (int i = 0; i < DT; ++i) dqa[jq][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float* dvb = dv + (int64_t)b * N * E + h * D;
  float* dkb = dk + (int64_t)b * N * E + h * D;

  auto compute = [&](const Frag& f, int tile, bool live) __attribute__((always_inline)) {
    f32x4 dvt[DT], dkt[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i) dvt[i] = dkt[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    uint32_t mcur[4] = {f.mb[0], f.mb[1], f.mb[2], f.mb[3]};
#pragma unroll
    for (int jq = 0; jq < NQT; ++jq) {
      uint32_t mnxt[4] = {0u, 0u, 0u, 0u};
      if (jq + 1 < NQT) {  // the next query tile's mask bytes: (query 16 (jq + 1) + 4 g + r, key key0 + n)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          mnxt[r] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(m_rsrc, moff[16 * (jq + 1) + 4 * g + r] + (uint32_t)n, tile * 16, 0);
      }
      // A operands of S and dP from LDS: lane (q = 16 jq + n, d = 8 g + t)
      const f32x4 qa0 = *reinterpret_cast<const f32x4*>(qs + (16 * jq + n) * RS + DK * g);
      const f32x4 qa1 = *reinterpret_cast<const f32x4*>(qs + (16 * jq + n) * RS + DK * g + 4);
      const f32x4 ga0 = *reinterpret_cast<const f32x4*>(gs + (16 * jq + n) * RS + DK * g);
      const f32x4 ga1 = *reinterpret_cast<const f32x4*>(gs + (16 * jq + n) * RS + DK * g + 4);
      f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa0[t], f.kf[0][t], sacc, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_16x16x4f32(ga0[t], f.vf[0][t], dp, 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa1[t], f.kf[1][t], sacc, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_16x16x4f32(ga1[t], f.vf[1][t], dp, 0, 0, 0);
      }
      f32x4 p, ds;
#pragma unroll
      for (int r = 0; r < 4; ++r) {  // rows of this lane: q = 16 jq + 4 g + r
        const float2 ld = *reinterpret_cast<const float2*>(ld2 + 2 * (16 * jq + 4 * g + r));
        float pv = __expf(sacc[r] - ld.x);
        pv = (mcur[r] != 0u || !live) ? 0.f : pv;
        p[r] = pv;
        ds[r] = pv * (dp[r] - ld.y);
      }
      // A operands of dV^T / dK^T from LDS: lane (d = 16 i + n, q = 16 jq + 4 g + t)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int i = 0; i < DT; ++i) {
          const float got = gs[(16 * jq + 4 * g + t) * RS + 16 * i + n];
          const float qt = qs[(16 * jq + 4 * g + t) * RS + 16 * i + n];
          dvt[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(got, p[t], dvt[i], 0, 0, 0);
          dkt[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(qt, ds[t], dkt[i], 0, 0, 0);
        }
      }
      // dS to the query-on-lane layout through the wave's LDS tile, then dQ += dS K
#pragma unroll
      for (int r = 0; r < 4; ++r) tr[wave][4 * g + r][n] = ds[r];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const f32x4 dsa = *reinterpret_cast<const f32x4*>(&tr[wave][n][4 * g]);  // lane (q = n, keys 4 g .. 4 g + 3)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t) dqa[jq][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(dsa[t], f.kr[i][t], dqa[jq][i], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) mcur[r] = mnxt[r];
    }
    if (live) {  // dV^T / dK^T tiles: column = key n, rows d = 16 i + 4 g + r  ->  float4 along d; this wave is the keys' only writer
      const int64_t ko = (int64_t)(tile * 16 + n) * E;
#pragma unroll
      for (int i = 0; i < DT; ++i) {
        *reinterpret_cast<f32x4*>(dvb + ko + 16 * i + 4 * g) = dvt[i];
        *reinterpret_cast<f32x4*>(dkb + ko + 16 * i + 4 * g) = dkt[i];
      }
    }
  };

  Frag fa, fb;
  int tile = split * tiles_per_split + wave;
  load_tile(tile, fa);
  while (tile < t_end) {
    load_tile(tile + kXWaves, fb);
    __builtin_amdgcn_sched_barrier(0);
    compute(fa, tile, true);
    __builtin_amdgcn_sched_barrier(0);
    load_tile(tile + 2 * kXWaves, fa);
    __builtin_amdgcn_sched_barrier(0);
    compute(fb, tile + kXWaves, tile + kXWaves < t_end);
    __builtin_amdgcn_sched_barrier(0);
    tile += 2 * kXWaves;
  }

  // ---- merge dQ over the 4 waves through LDS (re-using the staging area): C layout of dqa is (column = d n, rows q = 4 g + r)
  __syncthreads();
  float* part = lds;  // [kXWaves][QL][D]
#pragma unroll
  for (int jq = 0; jq < NQT; ++jq)
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) part[(wave * QL + 16 * jq + 4 * g + r) * D + 16 * i + n] = dqa[jq][i][r];
  __syncthreads();
  for (int idx = tid; idx < QL * (D / 4); idx += kXWaves * kWave) {
    const int ql = idx / (D / 4), c = idx - ql * (D / 4);
    if (ql >= Q) continue;
    float4 a = *reinterpret_cast<const float4*>(part + ql * D + 4 * c);
#pragma unroll
    for (int w = 1; w < kXWaves; ++w) {
      const float4 x = *reinterpret_cast<const float4*>(part + (w * QL + ql) * D + 4 * c);
      a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
    }
    *reinterpret_cast<float4*>(dq_ws + ((((int64_t)b * heads + h) * Q + ql) * n_splits + split) * D + 4 * c) = a;
  }
}

// bf16 form of the full-tile backward (bf16 autocast: q, k, v bf16 as saved by wm2f_masked_xattn_bf16_fwd; grad_out, lse, delta
// fp32; grad_k / grad_v bf16, dQ partials fp32).  Same orientation and the same five products per (key tile, query tile) as the
// fp32 kernel above, each as 2 v_mfma_f32_16x16x16_bf16 instead of 8 v_mfma_f32_16x16x4_f32; p and dS are computed in fp32 from
// the fp32 accumulators and rounded to bf16 only as matrix operands.  Operand sources:
//   S = Q K^T, dP = dO V^T      A = q / grad_out rows from the LDS staging (bf16), B = the lane's 16-byte K / V row piece
//                               (contraction order d = 8 g + 4 s + i on both sides, as in the forward)
//   dV^T += dO^T P, dK^T += Q^T dS   A = the staged rows read TRANSPOSED (ds_read_b64_tr_b16), B = p / dS as they stand
//   dQ += dS K                  A = dS through a wave-private LDS tile (the one transpose the orientation leaves), B = the K
//                               tile written to the wave's LDS image and read transposed
template <int NQT>
__global__ __launch_bounds__(kXWaves* kWave) void masked_xattn_bf16_bwd_full_kernel(
    const uint16_t* __restrict__ q, const uint16_t* __restrict__ k, const uint16_t* __restrict__ v,
    const uint8_t* __restrict__ mask, const int* __restrict__ row_open, const float* __restrict__ lse,
    const float* __restrict__ delta, const float* __restrict__ go, float* __restrict__ dq_ws,
    uint16_t* __restrict__ dk, uint16_t* __restrict__ dv, int Q, int N, int heads, int n_splits, int tiles_per_split) {
  constexpr int D = 32, DT = 2, QL = NQT * 16;
  constexpr int PB = 80;  // bytes per staged row: 32 bf16 + 16 bytes of padding (16-byte aligned rows, spread over the banks)
  constexpr int kStageBytes = 2 * QL * PB + QL * 12, kPartBytes = kXWaves * QL * D * 4;
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4_t;
  __shared__ __attribute__((aligned(16))) unsigned char lds[(kStageBytes > kPartBytes ? kStageBytes : kPartBytes)];
  __shared__ __attribute__((aligned(16))) uint16_t ktile[kXWaves][16 * D];   // a wave's K tile, [key][d] row-major
  __shared__ __attribute__((aligned(16))) uint16_t dst[kXWaves][16 * 20];    // a wave's dS tile, [q][key], 40-byte rows
  unsigned char* qs = lds;                // [QL] rows of PB bytes
  unsigned char* gs = lds + QL * PB;
  float* ld2 = reinterpret_cast<float*>(lds + 2 * QL * PB);                  // [QL][2] = (lse, delta)
  uint32_t* moff = reinterpret_cast<uint32_t*>(lds + 2 * QL * PB + QL * 8);  // [QL]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, n = lane & 15;
  const int split = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int E = heads * D;
  const uint16_t* qb = q + (int64_t)b * Q * E + h * D;
  const float* gob = go + (int64_t)b * Q * E + h * D;
  for (int i = tid; i < QL * (D / 4); i += kXWaves * kWave) {  // 4 elements per thread: 8 bytes of q, 16 bytes of grad_out
    const int r = i / (D / 4), c = i - r * (D / 4);
    s16x4 x = {0, 0, 0, 0};
    bf16x4_t y = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
    if (r < Q) {
      x = *reinterpret_cast<const s16x4*>(qb + (int64_t)r * E + 4 * c);
      const float4 gv = *reinterpret_cast<const float4*>(gob + (int64_t)r * E + 4 * c);
      y[0] = (__bf16)gv.x; y[1] = (__bf16)gv.y; y[2] = (__bf16)gv.z; y[3] = (__bf16)gv.w;
    }
    *reinterpret_cast<s16x4*>(qs + r * PB + 8 * c) = x;
    *reinterpret_cast<bf16x4_t*>(gs + r * PB + 8 * c) = y;
  }
  for (int r = tid; r < QL; r += kXWaves * kWave) {
    const bool ok = r < Q;
    ld2[2 * r] = ok ? lse[((int64_t)b * heads + h) * Q + r] : INFINITY;
    ld2[2 * r + 1] = ok ? delta[((int64_t)b * heads + h) * Q + r] : 0.f;
    const bool use = mask != nullptr && ok && (row_open == nullptr || row_open[(int64_t)b * Q + r] != 0);
    moff[r] = use ? (uint32_t)(r * N) : 0x80000000u;
  }
  __syncthreads();

  const int n_tiles = N / 16;
  int t_end = (split + 1) * tiles_per_split;
  if (t_end > n_tiles) t_end = n_tiles;
  const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(k + (int64_t)b * N * E), 0, N * E * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(v + (int64_t)b * N * E), 0, N * E * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t m_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(mask != nullptr ? mask + (int64_t)b * Q * N : nullptr), 0, mask != nullptr ? Q * N : 0, 0x00020000);
  const uint32_t kv_voff = (uint32_t)((n * E + h * D + 8 * g) * 2);  // lane (key0 + n, d = 8 g ..)
  const int row_bytes = E * 2;
  struct Frag {
    s16x8 kf, vf;
    uint32_t mb[4];
  };
  auto load_tile = [&](int tile, Frag& f) __attribute__((always_inline)) {
    if (tile > n_tiles - 1) tile = n_tiles - 1;
    const int soff = tile * 16 * row_bytes;
    f.kf = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(k_rsrc, kv_voff, soff, 0));
    f.vf = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(v_rsrc, kv_voff, soff, 0));
#pragma unroll
    for (int r = 0; r < 4; ++r) f.mb[r] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(m_rsrc, moff[4 * g + r] + (uint32_t)n, tile * 16, 0);
  };

  f32x4 dqa[NQT][DT];
#pragma unroll
  for (int jq = 0; jq < NQT; ++jq)
#pragma unroll
    for (int i = 0; i < DT; ++i) dqa[jq][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  uint16_t* dvb = dv + (int64_t)b * N * E + h * D;
  uint16_t* dkb = dk + (int64_t)b * N * E + h * D;
  unsigned char* kt = reinterpret_cast<unsigned char*>(&ktile[wave][0]);
  unsigned char* dt = reinterpret_cast<unsigned char*>(&dst[wave][0]);
  const int tq = (lane >> 2) & 3, tp = lane & 3;
  const int tr_row = 4 * g + tq;          // row of a transposed read's 4 x 16 block this lane addresses
  const int kt_rd = tr_row * (D * 2) + tp * 8;  // + 32 i
  auto pack4 = [](const f32x4& x) __attribute__((always_inline)) {
    bf16x4_t r;
    r[0] = (__bf16)x[0]; r[1] = (__bf16)x[1]; r[2] = (__bf16)x[2]; r[3] = (__bf16)x[3];
    return __builtin_bit_cast(s16x4, r);
  };

  auto compute = [&](const Frag& f, int tile, bool live) __attribute__((always_inline)) {
    asm volatile("" ::: "memory");
    *reinterpret_cast<s16x8*>(kt + n * (D * 2) + g * 16) = f.kf;  // the K tile, row-major, for the transposed read of dQ's B operand
    asm volatile("" ::: "memory");
    const s16x4 k0 = __builtin_shufflevector(f.kf, f.kf, 0, 1, 2, 3), k1 = __builtin_shufflevector(f.kf, f.kf, 4, 5, 6, 7);
    const s16x4 v0 = __builtin_shufflevector(f.vf, f.vf, 0, 1, 2, 3), v1 = __builtin_shufflevector(f.vf, f.vf, 4, 5, 6, 7);
    s16x4 kb[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i) kb[i] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(kt + kt_rd + 32 * i));
    f32x4 dvt[DT], dkt[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i) dvt[i] = dkt[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    uint32_t mcur[4] = {f.mb[0], f.mb[1], f.mb[2], f.mb[3]};
#pragma unroll
    for (int jq = 0; jq < NQT; ++jq) {
      uint32_t mnxt[4] = {0u, 0u, 0u, 0u};
      if (jq + 1 < NQT) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          mnxt[r] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(m_rsrc, moff[16 * (jq + 1) + 4 * g + r] + (uint32_t)n, tile * 16, 0);
      }
      // A operands of S and dP: lane (q = 16 jq + n, d = 8 g .. 8 g + 7)
      const s16x8 qa = *reinterpret_cast<const s16x8*>(qs + (16 * jq + n) * PB + 16 * g);
      const s16x8 ga = *reinterpret_cast<const s16x8*>(gs + (16 * jq + n) * PB + 16 * g);
      f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
      sacc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_shufflevector(qa, qa, 0, 1, 2, 3), k0, sacc, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_shufflevector(ga, ga, 0, 1, 2, 3), v0, dp, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_shufflevector(qa, qa, 4, 5, 6, 7), k1, sacc, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_shufflevector(ga, ga, 4, 5, 6, 7), v1, dp, 0, 0, 0);
      f32x4 p, ds;
#pragma unroll
      for (int r = 0; r < 4; ++r) {  // rows of this lane: q = 16 jq + 4 g + r, column: key n
        const float2 ld = *reinterpret_cast<const float2*>(ld2 + 2 * (16 * jq + 4 * g + r));
        float pv = __expf(sacc[r] - ld.x);
        pv = (mcur[r] != 0u || !live) ? 0.f : pv;
        p[r] = pv;
        ds[r] = pv * (dp[r] - ld.y);
      }
      const s16x4 pb = pack4(p), dsb = pack4(ds);
      // A operands of dV^T / dK^T: the staged rows transposed -- block rows q = 16 jq + 4 g .. + 3, columns d = 16 i ..
#pragma unroll
      for (int i = 0; i < DT; ++i) {
        const s16x4 got = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(gs + (16 * jq + tr_row) * PB + 32 * i + 8 * tp));
        const s16x4 qt = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(qs + (16 * jq + tr_row) * PB + 32 * i + 8 * tp));
        dvt[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(got, pb, dvt[i], 0, 0, 0);
        dkt[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(qt, dsb, dkt[i], 0, 0, 0);
      }
      // dS to the query-on-lane layout through the wave's LDS tile: element (q = 4 g + r, key n) -> row q, then 4 keys per lane
      asm volatile("" ::: "memory");
#pragma unroll
      for (int r = 0; r < 4; ++r) *reinterpret_cast<short*>(dt + (4 * g + r) * 40 + 2 * n) = dsb[r];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const s16x4 dsa = *reinterpret_cast<const s16x4*>(dt + n * 40 + 8 * g);  // lane (q = n, keys 4 g .. 4 g + 3)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < DT; ++i) dqa[jq][i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(dsa, kb[i], dqa[jq][i], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) mcur[r] = mnxt[r];
    }
    if (live) {  // dV^T / dK^T tiles: column = key n, rows d = 16 i + 4 g + r -> 4 consecutive d of one key: 8 bytes of bf16
      const int64_t ko = (int64_t)(tile * 16 + n) * E;
#pragma unroll
      for (int i = 0; i < DT; ++i) {
        *reinterpret_cast<s16x4*>(dvb + ko + 16 * i + 4 * g) = pack4(dvt[i]);
        *reinterpret_cast<s16x4*>(dkb + ko + 16 * i + 4 * g) = pack4(dkt[i]);
      }
    }
  };

  Frag fa, fb;
  int tile = split * tiles_per_split + wave;
  load_tile(tile, fa);
  while (tile < t_end) {
    load_tile(tile + kXWaves, fb);
    __builtin_amdgcn_sched_barrier(0);
    compute(fa, tile, true);
    __builtin_amdgcn_sched_barrier(0);
    load_tile(tile + 2 * kXWaves, fa);
    __builtin_amdgcn_sched_barrier(0);
    compute(fb, tile + kXWaves, tile + kXWaves < t_end);
    __builtin_amdgcn_sched_barrier(0);
    tile += 2 * kXWaves;
  }

  // ---- merge dQ over the 4 waves through LDS (re-using the staging area): C layout of dqa is (column = d n, rows q = 4 g + r)
  __syncthreads();
  float* part = reinterpret_cast<float*>(lds);  // [kXWaves][QL][D]
#pragma unroll
  for (int jq = 0; jq < NQT; ++jq)
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) part[(wave * QL + 16 * jq + 4 * g + r) * D + 16 * i + n] = dqa[jq][i][r];
  __syncthreads();
  for (int idx = tid; idx < QL * (D / 4); idx += kXWaves * kWave) {
    const int ql = idx / (D / 4), c = idx - ql * (D / 4);
    if (ql >= Q) continue;
    float4 a = *reinterpret_cast<const float4*>(part + ql * D + 4 * c);
#pragma unroll
    for (int w = 1; w < kXWaves; ++w) {
      const float4 x = *reinterpret_cast<const float4*>(part + (w * QL + ql) * D + 4 * c);
      a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
    }
    *reinterpret_cast<float4*>(dq_ws + ((((int64_t)b * heads + h) * Q + ql) * n_splits + split) * D + 4 * c) = a;
  }
}

template <int D>
__global__ __launch_bounds__(256) void xattn_dq_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dq,
                                                              int B, int heads, int Q, int n_splits) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (b, h, q, float4 chunk)
  constexpr int CH = D / 4;
  if (idx >= (int64_t)B * heads * Q * CH) return;
  const int c = (int)(idx % CH);
  const int64_t row = idx / CH;
  const int qi = (int)(row % Q);
  const int64_t bh = row / Q;
  const int h = (int)(bh % heads), b = (int)(bh / heads);
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s = 0; s < n_splits; ++s) {
    const float4 x = *reinterpret_cast<const float4*>(ws + (row * n_splits + s) * D + 4 * c);
    a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
  }
  *reinterpret_cast<float4*>(dq + ((int64_t)b * Q + qi) * heads * D + h * D + 4 * c) = a;
}

}  // namespace wm2f

extern "C" int64_t wm2f_masked_xattn_bwd_workspace(int B, int heads, int Q, int N, int D) {
  if (B <= 0 || heads <= 0 || Q <= 0 || N <= 0 || D <= 0) return 0;
  // dQ partials per split + delta
  return ((int64_t)B * heads * Q * xattn_splits(B, heads, N, false) * D + (int64_t)B * heads * Q) * 4;
}

extern "C" int wm2f_masked_xattn_bwd(const void* q, const void* k, const void* v, const void* mask,
                                     const void* row_open, const void* out, const void* lse, const void* grad_out,
                                     void* grad_q, void* grad_k, void* grad_v, void* workspace, int B, int heads,
                                     int Q, int N, int D, int dtype, void* stream) {
  const char* who = "wm2f_masked_xattn_bwd";
  WM2F_REQUIRE(dtype == WM2F_F32, "%s: only WM2F_F32 is built", who);
  WM2F_REQUIRE(q && k && v && out && lse && grad_out && grad_q && grad_k && grad_v && workspace, "%s: null pointer", who);
  WM2F_REQUIRE(B > 0 && heads > 0 && Q > 0 && N > 0, "%s: non-positive size", who);
  WM2F_REQUIRE(D == 16 || D == 32 || D == 64, "%s: head_dim %d not in {16,32,64}", who, D);
  WM2F_REQUIRE(heads <= 65535 && B <= 65535, "%s: heads / B exceed the grid limits", who);
  const int n_splits = xattn_splits(B, heads, N, false);
  const int tps = ceil_div(ceil_div(N, 16), n_splits);
  hipStream_t st = (hipStream_t)stream;
  float* dq_ws = (float*)workspace;
  float* delta = dq_ws + (int64_t)B * heads * Q * n_splits * D;
  const int64_t nrow = (int64_t)B * heads * Q;
#define WM2F_BD(Dv, NQTv)                                                                                          \
  {                                                                                                                \
    hipLaunchKernelGGL((xattn_delta_kernel<Dv>), dim3((unsigned)ceil_div64(nrow, 256)), dim3(256), 0, st,          \
                       (const float*)out, (const float*)grad_out, delta, B, heads, Q);                             \
    const int q_chunks = ceil_div(Q, NQTv * 16);                                                                   \
    if (q_chunks > 1) { /* dK / dV are sums over the query chunks: zero them, the chunks add atomically */         \
      const size_t kv_bytes = (size_t)B * N * heads * Dv * 4;                                                       \
      if (hipMemsetAsync(grad_k, 0, kv_bytes, st) != hipSuccess || hipMemsetAsync(grad_v, 0, kv_bytes, st) != hipSuccess) { \
        set_error("%s: hipMemsetAsync failed", who);                                                               \
        return WM2F_ELAUNCH;                                                                                       \
      }                                                                                                            \
    }                                                                                                              \
    dim3 grid(n_splits* q_chunks, heads, B);                                                                       \
    if (full_bwd && Dv == 32 && q_chunks == 1)                                                                     \
      hipLaunchKernelGGL((masked_xattn_bwd_full_kernel<NQTv>), dim3(n_splits, heads, B), dim3(kXWaves* kWave), 0, st, \
                         (const float*)q, (const float*)k, (const float*)v, (const uint8_t*)mask, (const int*)row_open, \
                         (const float*)lse, (const float*)delta, (const float*)grad_out, dq_ws, (float*)grad_k,    \
                         (float*)grad_v, Q, N, heads, n_splits, tps);                                              \
    else                                                                                                           \
    hipLaunchKernelGGL((masked_xattn_bwd_kernel<NQTv, Dv>), grid, dim3(kXWaves* kWave), 0, st, (const float*)q,    \
                       (const float*)k, (const float*)v, (const uint8_t*)mask, (const int*)row_open,               \
                       (const float*)lse, (const float*)delta, (const float*)grad_out, dq_ws, (float*)grad_k,      \
                       (float*)grad_v, Q, N, heads, n_splits, tps, q_chunks > 1 ? 1 : 0);                          \
    hipLaunchKernelGGL((xattn_dq_reduce_kernel<Dv>), dim3((unsigned)ceil_div64(nrow*(Dv / 4), 256)), dim3(256), 0, \
                       st, (const float*)dq_ws, (float*)grad_q, B, heads, Q, n_splits);                            \
  }
  const int q_tiles = ceil_div(Q, 16);
  // full-tile backward: whole key tiles, 32-bit byte offsets (the general kernel takes every other shape)
  const bool full_bwd = D == 32 && (N % 16) == 0 && (int64_t)N * heads * D * 4 < (int64_t(1) << 31) && (int64_t)Q * N < (int64_t(1) << 31) &&
                        tps >= 8 &&  // each workgroup stages its 112 query rows once: with 4 tiles per split (N = 1024) that costs more than it saves (237 against 189 us)
                        tune_env("WM2F_K2_BWD_FULL", 1) != 0;
  if (D == 16) {
    if (q_tiles <= 4) WM2F_BD(16, 4) else WM2F_BD(16, 7)
  } else if (D == 32) {
    if (q_tiles <= 4) WM2F_BD(32, 4) else WM2F_BD(32, 7)
  } else {
    WM2F_BD(64, 3)
  }
#undef WM2F_BD
  WM2F_CHECK_LAUNCH(who);
  return WM2F_OK;
}

/* bf16 operands (include/wm2f.h): q, k, v and grad_k / grad_v bf16; out, lse, grad_out, grad_q fp32.  Full-tile form only. */
extern "C" int wm2f_masked_xattn_bf16_bwd(const void* q, const void* k, const void* v, const void* mask, const void* row_open,
                                          const void* out, const void* lse, const void* grad_out, void* grad_q, void* grad_k,
                                          void* grad_v, void* workspace, int B, int heads, int Q, int N, int D, void* stream) {
  const char* who = "wm2f_masked_xattn_bf16_bwd";
  WM2F_REQUIRE(q && k && v && out && lse && grad_out && grad_q && grad_k && grad_v && workspace, "%s: null pointer", who);
  WM2F_REQUIRE(B > 0 && heads > 0 && Q > 0 && N > 0, "%s: non-positive size", who);
  WM2F_REQUIRE(heads <= 65535 && B <= 65535, "%s: heads / B exceed the grid limits", who);
  if (D != 32 || (N % 16) != 0 || Q > 112 || (int64_t)N * heads * D * 2 >= (int64_t(1) << 31) || (int64_t)Q * N >= (int64_t(1) << 31) ||
      ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v) |
        reinterpret_cast<uintptr_t>(grad_k) | reinterpret_cast<uintptr_t>(grad_v) | reinterpret_cast<uintptr_t>(grad_out)) & 15) != 0) {
    set_error("%s: needs head_dim 32, N %% 16 == 0, Q <= 112, 16-byte aligned operands and 32-bit offsets (cast to fp32 and use "
              "wm2f_masked_xattn_bwd)", who);
    return WM2F_EUNSUPPORTED;
  }
  const int n_splits = xattn_splits(B, heads, N, false);
  const int tps = ceil_div(N / 16, n_splits);
  hipStream_t st = (hipStream_t)stream;
  float* dq_ws = (float*)workspace;
  float* delta = dq_ws + (int64_t)B * heads * Q * n_splits * D;
  const int64_t nrow = (int64_t)B * heads * Q;
  hipLaunchKernelGGL((xattn_delta_kernel<32>), dim3((unsigned)ceil_div64(nrow, 256)), dim3(256), 0, st, (const float*)out,
                     (const float*)grad_out, delta, B, heads, Q);
#define WM2F_BB(NQTv)                                                                                                       \
  hipLaunchKernelGGL((masked_xattn_bf16_bwd_full_kernel<NQTv>), dim3(n_splits, heads, B), dim3(kXWaves* kWave), 0, st,          \
                     (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v, (const uint8_t*)mask, (const int*)row_open, \
                     (const float*)lse, (const float*)delta, (const float*)grad_out, dq_ws, (uint16_t*)grad_k,              \
                     (uint16_t*)grad_v, Q, N, heads, n_splits, tps)
  if (ceil_div(Q, 16) <= 4) WM2F_BB(4);
  else WM2F_BB(7);
#undef WM2F_BB
  WM2F_CHECK_LAUNCH(who);
  hipLaunchKernelGGL((xattn_dq_reduce_kernel<32>), dim3((unsigned)ceil_div64(nrow * 8, 256)), dim3(256), 0, st, (const float*)dq_ws,
                     (float*)grad_q, B, heads, Q, n_splits);
  WM2F_CHECK_LAUNCH(who);
  return WM2F_OK;
}
