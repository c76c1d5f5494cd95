// K3 in bf16 (BASELINE configs 3-5 run under bf16 autocast): logits = einsum("bqc,bchw->bqhw") with bf16 operands,
// fp32 accumulation on v_mfma_f32_16x16x16_bf16 and fp32 output (the attention-mask build and the loss read fp32).
// Replaces the same line as mask_einsum.hip: transformers modeling_mask2former.py:2046.
//
// Roofline: HBM.  Per call at config 2: pixel features 8 x 65536 x 256 x 2 B = 268 MB read + logits 8 x 100 x 65536 x
// 4 B = 210 MB written = 478 MB -> 60 us at 8 TB/s, against 26.8 GFLOP = 11 us of dense bf16 MFMA.  (The fp32 kernel
// is MFMA-bound at 231 us.)
//
// Layout: the MFMA operands want the contraction index (channel) contiguous per lane, so the pixel features are
// consumed PIXEL-MAJOR, (B, HW, C) -- one tiled transpose per forward (wm2f_nchw_to_pixel_major_bf16), shared by
// the ten mask-predictor calls.  The channel -> (k-step, lane group, element) assignment is free as long as both
// operands use the same one; it is chosen so that a lane fetches 16 contiguous bytes (8 channels = two k-steps) of
// its pixel per load: channel = 32 * s2 + 8 * kb + 4 * half + e.
//   workgroup = 4 waves x 64 pixels; the query embeddings (<= 112 x C bf16) sit in LDS, padded rows are zero.
//   wave: 7 query tiles x 4 pixel blocks of accumulators (112 VGPRs), B fragments double-buffered in registers.
#include "common.h"

namespace wm2f {
namespace {

using bf16x4 = __attribute__((ext_vector_type(4))) short;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

constexpr int kEPad = 8;  // bf16 elements of row padding in LDS (16 B): ds_read_b64 of 16 rows x 4 lane groups conflict-free

template <int NQT>
__global__ __launch_bounds__(256) void mask_einsum_bf16_kernel(const uint16_t* __restrict__ emb,
                                                              const uint16_t* __restrict__ pixt, float* __restrict__ out,
                                                              int Q, int C, int HW) {
  extern __shared__ __attribute__((aligned(16))) uint16_t e_lds[];  // [NQT * 16][C + kEPad]
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, kb = lane >> 4;
  const int ldc = C + kEPad;
  // ---- query embeddings -> LDS (16-byte pieces; rows >= Q are zero)
  const int pieces = C / 8;
  for (int i = tid; i < NQT * 16 * pieces; i += 256) {
    const int r = i / pieces, c8 = i - r * pieces;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (r < Q) v = *reinterpret_cast<const u32x4*>(emb + ((int64_t)b * Q + r) * C + c8 * 8);
    *reinterpret_cast<u32x4*>(e_lds + r * ldc + c8 * 8) = v;
  }
  __syncthreads();

  const int p0 = blockIdx.x * 256 + wave * 64;
  if (p0 >= HW) return;
  const uint16_t* prow[4];
  bool pok[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    int p = p0 + nb * 16 + n;
    pok[nb] = p < HW;
    if (!pok[nb]) p = HW - 1;
    prow[nb] = pixt + ((int64_t)b * HW + p) * C + kb * 8;
  }
  f32x4 acc[NQT][4];
#pragma unroll
  for (int j = 0; j < NQT; ++j)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) acc[j][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int steps = C / 32;
  u32x4 bn[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) bn[nb] = *reinterpret_cast<const u32x4*>(prow[nb]);
  for (int s2 = 0; s2 < steps; ++s2) {
    u32x4 bc[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) bc[nb] = bn[nb];
    if (s2 + 1 < steps) {
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) bn[nb] = *reinterpret_cast<const u32x4*>(prow[nb] + (s2 + 1) * 32);
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      bf16x4 bf[4];
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const unsigned lo = half ? bc[nb].z : bc[nb].x, hi = half ? bc[nb].w : bc[nb].y;
        bf[nb] = __builtin_bit_cast(bf16x4, (unsigned long long)lo | ((unsigned long long)hi << 32));
      }
#pragma unroll
      for (int j = 0; j < NQT; ++j) {
        // A fragment: lane (query 16 j + n, lane group kb): channels 32 s2 + 8 kb + 4 half + 0..3
        const bf16x4 af = *reinterpret_cast<const bf16x4*>(e_lds + (16 * j + n) * ldc + s2 * 32 + kb * 8 + half * 4);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
          acc[j][nb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(af, bf[nb], acc[j][nb], 0, 0, 0);
      }
    }
  }
  // ---- C layout: lane holds rows (queries) 16 j + 4 kb + r, column (pixel) p0 + 16 nb + n
#pragma unroll
  for (int j = 0; j < NQT; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = 16 * j + 4 * kb + r;
      if (q >= Q) continue;
      float* orow = out + ((int64_t)b * Q + q) * HW + p0 + n;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
        if (pok[nb]) orow[nb * 16] = acc[j][nb][r];
    }
}

// (B, C, HW) bf16 -> (B, HW, C) bf16, LDS tile transpose, both sides coalesced.
// 64 x 64 tiles moved as 32-bit pairs (C and HW even): a lane group reads 128 contiguous bytes of a channel row and writes 128
// contiguous bytes of a pixel row.  (The first form moved 32 x 32 tiles one bf16 at a time -- 64-byte runs on both sides: 2.3 TB/s.)
__global__ __launch_bounds__(256) void nchw_to_pixel_major_bf16_pairs_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst,
                                                                            int C, int HW) {
  __shared__ uint16_t tile[64][66];
  const int b = blockIdx.z, p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int r = 0; r < 64; r += 8) {
    const int c = c0 + ty + r, p = p0 + 2 * tx;
    unsigned v = 0u;
    if (c < C && p < HW) v = *reinterpret_cast<const unsigned*>(src + ((int64_t)b * C + c) * HW + p);
    *reinterpret_cast<unsigned*>(&tile[ty + r][2 * tx]) = v;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 64; r += 8) {
    const int p = p0 + ty + r, c = c0 + 2 * tx;
    if (p < HW && c < C)
      *reinterpret_cast<unsigned*>(dst + ((int64_t)b * HW + p) * C + c) = (unsigned)tile[2 * tx][ty + r] | ((unsigned)tile[2 * tx + 1][ty + r] << 16);
  }
}

__global__ __launch_bounds__(256) void nchw_to_pixel_major_bf16_kernel(const uint16_t* __restrict__ src,
                                                                      uint16_t* __restrict__ dst, int C, int HW) {
  __shared__ uint16_t tile[32][34];
  const int b = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int r = 0; r < 32; r += 8) {
    const int c = c0 + ty + r, p = p0 + tx;
    tile[ty + r][tx] = (c < C && p < HW) ? src[((int64_t)b * C + c) * HW + p] : (uint16_t)0;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 32; r += 8) {
    const int p = p0 + ty + r, c = c0 + tx;
    if (p < HW && c < C) dst[((int64_t)b * HW + p) * C + c] = tile[tx][ty + r];
  }
}


// =====================================================================================================
// Backward under bf16 autocast (replaces grad.to(bf16) + two batched library GEMMs): grad_out is fp32 (the logits are),
// emb / pix and both gradients are bf16, accumulation fp32.
//   g_pix[b][c][p] = sum_q emb[b][q][c] * grad[b][q][p]: one wave = 64 channels x 64-pixel strips; A = emb^T fragments
//       (from a small (B, C, Qp) transposed copy) stay in registers for the wave's lifetime, B = grad rows loaded as
//       fp32 float4 (4 consecutive pixels = the 4 column tiles), rounded to bf16 in registers (v_cvt_pk_bf16_f32, RNE
//       as torch's .to(bfloat16)); stores 8 B per lane and row = 128 contiguous bytes per channel row.
//   g_emb[b][q][c] = sum_p grad[b][q][p] * pix[b][c][p]: both operands contiguous along the contraction (pix NCHW);
//       split over pixel ranges, fp32 partial tiles to the workspace, summed in range order (deterministic).
// HBM-bound: grad (B Q HW 4 B) + g_pix (B C HW 2 B) for the first, grad + pix for the second.

// v_cvt_pk_bf16_f32 through the vector conversion, NOT inline asm: the result feeds MFMA operands, and the wait states
// between a vector write and a matrix read are inserted by the compiler only for instructions it can see (with the asm
// form a quarter of g_emb came out as inf / garbage).
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ bf16x4 frag4(unsigned a, unsigned b) {
  return __builtin_bit_cast(bf16x4, (unsigned long long)a | ((unsigned long long)b << 32));
}

__global__ void emb_transpose_pad_bf16_kernel(const uint16_t* __restrict__ emb, uint16_t* __restrict__ out, int Q, int C, int Qp,
                                              int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // out (B, C, Qp)
  if (i >= total) return;
  const int q = (int)(i % Qp);
  const int64_t bc = i / Qp;
  const int c = (int)(bc % C);
  const int64_t b = bc / C;
  out[i] = q < Q ? emb[(b * Q + q) * C + c] : (uint16_t)0;
}

template <int KS>
__global__ __launch_bounds__(256) void mask_einsum_bf16_bwd_pix_kernel(const uint16_t* __restrict__ embt, const float* __restrict__ go,
                                                                       uint16_t* __restrict__ g_pix, int Q, int C, int HW,
                                                                       int strips_per_wg) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 15, kb = lane >> 4;
  const int b = blockIdx.y;
  const int c0 = (blockIdx.z * 4 + wave) * 64;
  if (c0 >= C) return;  // whole wave; no barrier in this kernel
  constexpr int Qp = KS * 16;
  bf16x4 af[KS][4];  // A[m = channel c0 + 16 mt + n][k = query 16 ks + 4 kb + e]
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
      af[ks][mt] = *reinterpret_cast<const bf16x4*>(embt + ((int64_t)b * C + c0 + 16 * mt + n) * Qp + 16 * ks + 4 * kb);
  const uint32_t kOob = 0x80000000u;
  const __amdgpu_buffer_rsrc_t go_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(go + (int64_t)b * Q * HW), 0, Q * HW * 4, 0x00020000);
  const bool last_ok = 16 * (KS - 1) + 4 * kb < Q;  // Q % 4 == 0: a lane group's 4 query rows exist together
  const int row_bytes = HW * 4;
  uint16_t* out_b = g_pix + (int64_t)b * C * HW;
  for (int i = 0; i < strips_per_wg; ++i) {
    const int p0 = (blockIdx.x * strips_per_wg + i) * 64 + 4 * n;  // this lane's 4 pixels = column tiles 0..3
    if (p0 - 4 * n >= HW) break;
    const bool pvalid = p0 < HW;  // HW % 4 == 0
    const uint32_t voff = pvalid ? (uint32_t)((4 * kb * HW + p0) * 4) : kOob;
    const uint32_t voff_last = last_ok ? voff : kOob;
    f32x4 acc[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 x[KS][4];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        x[ks][e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(go_rsrc, ks == KS - 1 ? voff_last : voff,
                                                                                     (16 * ks + e) * row_bytes, 0));
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x4 bf[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[j] = frag4(pack_bf16(x[ks][0][j], x[ks][1][j]), pack_bf16(x[ks][2][j], x[ks][3][j]));
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(af[ks][mt], bf[j], acc[mt][j], 0, 0, 0);
    }
    if (pvalid) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = c0 + 16 * mt + 4 * kb + r;
          uint2 v;
          v.x = pack_bf16(acc[mt][0][r], acc[mt][1][r]);
          v.y = pack_bf16(acc[mt][2][r], acc[mt][3][r]);
          *reinterpret_cast<uint2*>(out_b + (int64_t)c * HW + p0) = v;
        }
    }
  }
}

template <int MT>
__global__ __launch_bounds__(256) void mask_einsum_bf16_bwd_emb_kernel(const float* __restrict__ go, const uint16_t* __restrict__ pix,
                                                                       float* __restrict__ ws, int Q, int C, int HW, int n_split,
                                                                       int px_per_split, int q_chunks) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 15, kb = lane >> 4;
  const int split = blockIdx.x, b = blockIdx.y / q_chunks, chunk = blockIdx.y % q_chunks;
  const int q0 = chunk * MT * 16;
  const int n_ct = C / 16;
  const int ct0 = (blockIdx.z * 4 + wave) * 4;
  if (ct0 >= n_ct) return;
  const int p_begin = split * px_per_split;
  const int p_end = min(HW, p_begin + px_per_split);
  if (p_begin >= p_end) return;
  const uint32_t kOob = 0x80000000u;
  const __amdgpu_buffer_rsrc_t go_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(go + (int64_t)b * Q * HW), 0, Q * HW * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t pix_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(pix + (int64_t)b * C * HW), 0, C * HW * 2, 0x00020000);
  // lane (m, kb): row m, pixels p + 8 kb .. 8 kb + 7 of the 32-pixel step; element e of half h is the k-value 4 h + e
  uint32_t a_voff[MT], b_voff[4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = q0 + 16 * mt + m;
    a_voff[mt] = row < Q ? (uint32_t)((row * HW + 8 * kb) * 4) : kOob;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = (ct0 + j) * 16 + m;
    b_voff[j] = (ct0 + j < n_ct) ? (uint32_t)((c * HW + 8 * kb) * 2) : kOob;
  }
  struct Frag {
    f32x4 a[MT][2];
    u32x4 bq[4];
  };
  f32x4 acc[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int n_it = ceil_div(p_end - p_begin, 32);
  auto load = [&](Frag& f, int it) __attribute__((always_inline)) {
    const int p = p_begin + 32 * it;
    const bool ok = p + 8 * kb < p_end;  // HW % 8 == 0 and ranges of whole 32-pixel steps: a lane's 8 pixels are in or out together
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        f.a[mt][h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(go_rsrc, (ok ? a_voff[mt] : kOob) + 16 * h, p * 4, 0));
#pragma unroll
    for (int j = 0; j < 4; ++j)
      f.bq[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(pix_rsrc, ok ? b_voff[j] : kOob, p * 2, 0));
  };
  auto compute = [&](const Frag& f) __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      bf16x4 bf[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[j] = h ? frag4(f.bq[j].z, f.bq[j].w) : frag4(f.bq[j].x, f.bq[j].y);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const bf16x4 af = frag4(pack_bf16(f.a[mt][h][0], f.a[mt][h][1]), pack_bf16(f.a[mt][h][2], f.a[mt][h][3]));
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(af, bf[j], acc[mt][j], 0, 0, 0);
      }
    }
  };
  Frag f0, f1;
  load(f0, 0);
  for (int it = 0; it < n_it; it += 2) {
    load(f1, it + 1);
    __builtin_amdgcn_sched_barrier(0);
    compute(f0);
    __builtin_amdgcn_sched_barrier(0);
    load(f0, it + 2);
    __builtin_amdgcn_sched_barrier(0);
    compute(f1);
    __builtin_amdgcn_sched_barrier(0);
  }
  float* w = ws + (((int64_t)blockIdx.y * n_split + split) * (MT * 16)) * C;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (ct0 + j >= n_ct) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) w[(int64_t)(16 * mt + 4 * kb + r) * C + (ct0 + j) * 16 + m] = acc[mt][j][r];
    }
}

__global__ void mask_einsum_bf16_bwd_emb_reduce_kernel(const float* __restrict__ ws, uint16_t* __restrict__ g_emb, int Q, int C,
                                                       int n_split, int q_chunks, int rows_per_chunk, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // g_emb (B, Q, C) bf16, 4 elements per thread
  if (i >= total) return;
  const int C4 = C / 4;
  const int c4 = (int)(i % C4);
  const int64_t bq = i / C4;
  const int q = (int)(bq % Q);
  const int64_t b = bq / Q;
  const int chunk = q / rows_per_chunk, r = q - chunk * rows_per_chunk;
  const float* w = ws + (((b * q_chunks + chunk) * n_split) * rows_per_chunk + r) * C + 4 * c4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k = 0; k < n_split; ++k) {
    const float4 x = *reinterpret_cast<const float4*>(w + (int64_t)k * rows_per_chunk * C);
    s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
  }
  uint2 v;
  v.x = pack_bf16(s.x, s.y);
  v.y = pack_bf16(s.z, s.w);
  *reinterpret_cast<uint2*>(g_emb + bq * C + 4 * c4) = v;
}

}  // namespace
}  // namespace wm2f

using namespace wm2f;

extern "C" int wm2f_mask_einsum_bf16_fwd(const void* emb, const void* pix_pixel_major, void* out, int B, int Q, int C,
                                         int HW, void* stream) {
  const char* who = "wm2f_mask_einsum_bf16_fwd";
  WM2F_REQUIRE(emb && pix_pixel_major && out, "%s: null pointer", who);
  WM2F_REQUIRE(B > 0 && B < 65536 && Q > 0 && HW > 0, "%s: bad size", who);
  WM2F_REQUIRE(C >= 32 && C % 32 == 0 && C <= 512, "%s: C must be a multiple of 32 in [32, 512]", who);
  WM2F_REQUIRE(Q <= 112, "%s: at most 112 queries per call (7 query tiles); split the queries", who);
  const int nqt = ceil_div(Q, 16);
  const size_t lds = (size_t)nqt * 16 * (C + kEPad) * 2;
  dim3 grid(ceil_div(HW, 256), B);
#define WM2F_BF(NQTv)                                                                                                  \
  case NQTv: {                                                                                                         \
    auto kfn = mask_einsum_bf16_kernel<NQTv>;                                                                          \
    if (lds > 64 * 1024 &&                                                                                             \
        hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {   \
      set_error("%s: cannot raise dynamic LDS to %zu", who, lds);                                                      \
      return WM2F_ELAUNCH;                                                                                             \
    }                                                                                                                  \
    hipLaunchKernelGGL(kfn, grid, dim3(256), lds, (hipStream_t)stream, (const uint16_t*)emb,                           \
                       (const uint16_t*)pix_pixel_major, (float*)out, Q, C, HW);                                       \
  } break;
  switch (nqt) {
    WM2F_BF(1) WM2F_BF(2) WM2F_BF(3) WM2F_BF(4) WM2F_BF(5) WM2F_BF(6) WM2F_BF(7)
    default: set_error("%s: unsupported query count", who); return WM2F_EUNSUPPORTED;
  }
#undef WM2F_BF
  WM2F_CHECK_LAUNCH(who);
  return WM2F_OK;
}

extern "C" int wm2f_nchw_to_pixel_major_bf16(const void* src, void* dst, int B, int C, int HW, void* stream) {
  const char* who = "wm2f_nchw_to_pixel_major_bf16";
  WM2F_REQUIRE(src && dst, "%s: null pointer", who);
  WM2F_REQUIRE(B > 0 && B < 65536 && C > 0 && HW > 0 && ceil_div(C, 32) < 65536, "%s: bad size", who);
  if (C % 2 == 0 && HW % 2 == 0 && ((uintptr_t)src & 3) == 0 && ((uintptr_t)dst & 3) == 0 && ceil_div(C, 64) < 65536)
    hipLaunchKernelGGL(nchw_to_pixel_major_bf16_pairs_kernel, dim3(ceil_div(HW, 64), ceil_div(C, 64), B), dim3(256), 0,
                       (hipStream_t)stream, (const uint16_t*)src, (uint16_t*)dst, C, HW);
  else
    hipLaunchKernelGGL(nchw_to_pixel_major_bf16_kernel, dim3(ceil_div(HW, 32), ceil_div(C, 32), B), dim3(256), 0,
                       (hipStream_t)stream, (const uint16_t*)src, (uint16_t*)dst, C, HW);
  WM2F_CHECK_LAUNCH(who);
  return WM2F_OK;
}

namespace {
struct BwdPlanBf16 {
  int Qp, mt, q_chunks, n_split, px_per_split, nz;
  int64_t embt_bytes, ws_bytes;
};
BwdPlanBf16 bwd_plan_bf16(int B, int Q, int C, int HW) {
  BwdPlanBf16 p;
  p.Qp = ceil_div(Q, 16) * 16;
  const int q_tiles = ceil_div(Q, 16);
  p.q_chunks = ceil_div(q_tiles, 7);
  p.mt = ceil_div(q_tiles, p.q_chunks);
  p.nz = ceil_div(C, 256);
  int want = ceil_div(1024, B * p.q_chunks * p.nz);
  const int max_split = ceil_div(HW, 256);
  if (want > max_split) want = max_split;
  if (want < 1) want = 1;
  p.px_per_split = ceil_div(ceil_div(HW, want), 32) * 32;
  p.n_split = ceil_div(HW, p.px_per_split);
  p.embt_bytes = ((int64_t)B * C * p.Qp * 2 + 255) / 256 * 256;
  p.ws_bytes = (int64_t)B * p.q_chunks * p.n_split * p.mt * 16 * C * 4;
  return p;
}
}  // namespace

extern "C" int64_t wm2f_mask_einsum_bf16_bwd_workspace(int B, int Q, int C, int HW) {
  if (B <= 0 || Q <= 0 || C <= 0 || HW <= 0) return 0;
  const BwdPlanBf16 p = bwd_plan_bf16(B, Q, C, HW);
  return p.embt_bytes + p.ws_bytes;
}

extern "C" int wm2f_mask_einsum_bf16_bwd(const void* emb, const void* pix, const void* grad_out, void* g_emb, void* g_pix,
                                         void* workspace, int B, int Q, int C, int HW, void* stream) {
  const char* who = "wm2f_mask_einsum_bf16_bwd";
  WM2F_REQUIRE(emb && pix && grad_out && workspace && (g_emb || g_pix), "%s: null pointer", who);
  WM2F_REQUIRE(B > 0 && Q > 0 && C > 0 && HW > 0, "%s: non-positive size", who);
  if (C % 64 != 0 || Q % 4 != 0 || Q > 112 || HW % 8 != 0 || (int64_t)(C + 16) * HW * 2 >= (1ll << 31) ||
      (int64_t)(Q + 16) * HW * 4 >= (1ll << 31)) {
    set_error("%s: shape outside the kernels' (C %% 64 == 0, Q %% 4 == 0, Q <= 112, HW %% 8 == 0, one image's slab < 2 GiB): C=%d Q=%d HW=%d",
              who, C, Q, HW);
    return WM2F_EUNSUPPORTED;
  }
  const BwdPlanBf16 p = bwd_plan_bf16(B, Q, C, HW);
  WM2F_REQUIRE((int64_t)B * p.q_chunks <= 65535, "%s: B exceeds the grid limit", who);
  hipStream_t st = (hipStream_t)stream;
  uint16_t* embt = (uint16_t*)workspace;
  float* ws = (float*)((char*)workspace + p.embt_bytes);
  if (g_pix) {
    const int64_t total = (int64_t)B * C * p.Qp;
    hipLaunchKernelGGL(emb_transpose_pad_bf16_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, st, (const uint16_t*)emb, embt,
                       Q, C, p.Qp, total);
    WM2F_CHECK_LAUNCH(who);
    const int n_strips = ceil_div(HW, 64);
    int spw = 1;  // strips per workgroup: keep about 2048 workgroups
    while (spw < 8 && (int64_t)ceil_div(n_strips, spw * 2) * B * p.nz >= 2048) spw *= 2;
    dim3 grid(ceil_div(n_strips, spw), B, p.nz);
#define WM2F_BP(KSv)                                                                                                      \
  case KSv:                                                                                                               \
    hipLaunchKernelGGL((mask_einsum_bf16_bwd_pix_kernel<KSv>), grid, dim3(256), 0, st, (const uint16_t*)embt, (const float*)grad_out, \
                       (uint16_t*)g_pix, Q, C, HW, spw);                                                                  \
    break;
    switch (p.Qp / 16) {
      WM2F_BP(1) WM2F_BP(2) WM2F_BP(3) WM2F_BP(4) WM2F_BP(5) WM2F_BP(6) WM2F_BP(7)
      default: set_error("%s: internal: %d query steps", who, p.Qp / 16); return WM2F_EINVAL;
    }
#undef WM2F_BP
    WM2F_CHECK_LAUNCH(who);
  }
  if (g_emb) {
    dim3 grid(p.n_split, B * p.q_chunks, p.nz);
#define WM2F_BE(MTv)                                                                                                      \
  case MTv:                                                                                                               \
    hipLaunchKernelGGL((mask_einsum_bf16_bwd_emb_kernel<MTv>), grid, dim3(256), 0, st, (const float*)grad_out, (const uint16_t*)pix, ws, \
                       Q, C, HW, p.n_split, p.px_per_split, p.q_chunks);                                                  \
    break;
    switch (p.mt) {
      WM2F_BE(1) WM2F_BE(2) WM2F_BE(3) WM2F_BE(4) WM2F_BE(5) WM2F_BE(6) WM2F_BE(7)
      default: set_error("%s: internal: %d row tiles", who, p.mt); return WM2F_EINVAL;
    }
#undef WM2F_BE
    WM2F_CHECK_LAUNCH(who);
    const int64_t total = (int64_t)B * Q * (C / 4);
    hipLaunchKernelGGL(mask_einsum_bf16_bwd_emb_reduce_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, st, (const float*)ws,
                       (uint16_t*)g_emb, Q, C, p.n_split, p.q_chunks, p.mt * 16, total);
    WM2F_CHECK_LAUNCH(who);
  }
  return WM2F_OK;
}
