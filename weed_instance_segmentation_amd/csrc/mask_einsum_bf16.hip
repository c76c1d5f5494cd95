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

// (B, C, HW) bf16 -> (B, HW, C) bf16, LDS tile transpose (32 x 32 elements), both sides coalesced
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
  hipLaunchKernelGGL(nchw_to_pixel_major_bf16_kernel, dim3(ceil_div(HW, 32), ceil_div(C, 32), B), dim3(256), 0,
                     (hipStream_t)stream, (const uint16_t*)src, (uint16_t*)dst, C, HW);
  WM2F_CHECK_LAUNCH(who);
  return WM2F_OK;
}
