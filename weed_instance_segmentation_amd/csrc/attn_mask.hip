// Attention-mask build: bilinear resize of the mask logits to the next level's size, sigmoid,
// threshold -- transformers modeling_mask2former.py:2048-2054 -- WITHOUT the x num_heads
// replication (every head holds the same mask, :2052), plus the per-row "any key open" flag that
// implements the fully-masked-row rule of :1912-1914.
//
//   logits (B, Q, H, W) fp32  ->  mask (B, Q, Hn*Wn) uint8 (1 = blocked), row_open (B, Q) int32
//
// Resize = torch upsample_bilinear2d(align_corners=False):
//   src = (dst + 0.5) * (in / out) - 0.5, clamped below at 0; i0 = (int)src; i1 = i0 + (i0 < in-1).
// One workgroup per (b, q) row; a thread produces 4 consecutive mask bytes per store.
// HBM-bound: reads <= 4*H*W per row, writes Hn*Wn bytes per row.
#include "common.h"

namespace wm2f {

__device__ __forceinline__ void src_index(int dst, float scale, int in_size, int& i0, int& i1, float& l1) {
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = src - (float)i0;
}

__global__ __launch_bounds__(256) void attn_mask_build_kernel(const float* __restrict__ logits,
                                                              uint8_t* __restrict__ mask, int* __restrict__ row_open,
                                                              int H, int W, int Hn, int Wn) {
  const int row = blockIdx.x;  // b*Q + q
  const float* src = logits + (int64_t)row * H * W;
  uint8_t* dst = mask + (int64_t)row * Hn * Wn;
  const float sh = (float)H / (float)Hn, sw = (float)W / (float)Wn;
  const int n = Hn * Wn;
  int any_open = 0;
  for (int base = threadIdx.x * 4; base < n; base += blockDim.x * 4) {
    uint32_t packed = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = base + e;
      uint32_t blocked = 0;
      if (i < n) {
        const int y = i / Wn, x = i - y * Wn;
        int y0, y1, x0, x1;
        float ly, lx;
        src_index(y, sh, H, y0, y1, ly);
        src_index(x, sw, W, x0, x1, lx);
        const float v00 = src[y0 * W + x0], v01 = src[y0 * W + x1];
        const float v10 = src[y1 * W + x0], v11 = src[y1 * W + x1];
        const float v = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
        const float sg = 1.f / (1.f + expf(-v));
        blocked = sg < 0.5f ? 1u : 0u;
        any_open |= (int)(blocked ^ 1u);
      }
      packed |= blocked << (8 * e);
    }
    if (base + 3 < n && ((n & 3) == 0)) {
      *reinterpret_cast<uint32_t*>(dst + base) = packed;  // rows are 4-byte aligned when n % 4 == 0
    } else {
      for (int e = 0; e < 4 && base + e < n; ++e) dst[base + e] = (uint8_t)((packed >> (8 * e)) & 0xff);
    }
  }
  any_open = __syncthreads_or(any_open);
  if (threadIdx.x == 0) row_open[row] = any_open ? 1 : 0;
}

}  // namespace wm2f

using namespace wm2f;

extern "C" int wm2f_attn_mask_build(const void* logits, void* mask, void* row_open, int B, int Q, int H, int W,
                                    int Hn, int Wn, void* stream) {
  const char* who = "wm2f_attn_mask_build";
  WM2F_REQUIRE(logits && mask && row_open, "%s: null pointer", who);
  WM2F_REQUIRE(B > 0 && Q > 0 && H > 0 && W > 0 && Hn > 0 && Wn > 0, "%s: non-positive size", who);
  hipLaunchKernelGGL(attn_mask_build_kernel, dim3(B * Q), dim3(256), 0, (hipStream_t)stream, (const float*)logits,
                     (uint8_t*)mask, (int*)row_open, H, W, Hn, Wn);
  WM2F_CHECK_LAUNCH(who);
  return WM2F_OK;
}
