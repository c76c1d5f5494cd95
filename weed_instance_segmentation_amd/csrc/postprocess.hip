// Device side of Mask2FormerImageProcessor.post_process_instance_segmentation
// (transformers 5.15.0 models/mask2former/image_processing_mask2former.py:627-746; callers: reference
// models/metrics.py:58-63, models/mask2former/inference.py:30).  SURVEY section 8(f) rank 2.
//
// The dependency resizes the (B, Q, h, w) mask logits to a hard-coded 384 x 384 grid (bilinear,
// align_corners = False; :680-682), takes (logit > 0) as the instance mask and the mean sigmoid over the mask
// as its quality (:703-708), resizes the binary masks to the target size with `nearest` (:715-717) and paints
// them in query order into one id map (:721-735) -- a Python loop with one host sync per query.
// Here nothing is materialised at 384 x 384: every kernel evaluates the bilinear sample it needs.
//   instance_scores : per (image, selected query) sum of sigmoid and count over the 384 x 384 grid
//   instance_any    : does a mask survive the `nearest` resize to the target size (only needed when the target
//                     is smaller than the grid; otherwise count > 0 says it)
//   instance_segmentation : id map at the target size: the LAST kept instance covering a pixel wins
//   instance_maps   : the kept binary masks at the target size (return_binary_maps)
// All HBM-bound streaming passes over small inputs; no roofline claim is made for them.
#include "common.h"

namespace wm2f {
namespace {

struct Grid {
  int h, w;      // logits
  int gh, gw;    // the dependency's fixed grid (384 x 384)
  float sh, sw;  // h / gh, w / gw: PyTorch's area_pixel_compute_scale (align_corners = False)
};

// upsample_bilinear2d(align_corners=False) at grid pixel (gy, gx): ATen UpSampleKernel.cpp HelperInterpLinear --
// source index scale * (i + 0.5) - 0.5 clamped at 0, second tap clamped to the last row / column, the x
// interpolation first, then y.  The products are kept unfused (as separate roundings).
__device__ __forceinline__ float grid_logit(const float* __restrict__ p, const Grid& g, int gy, int gx) {
  float sy = g.sh * ((float)gy + 0.5f) - 0.5f, sx = g.sw * ((float)gx + 0.5f) - 0.5f;
  sy = sy < 0.f ? 0.f : sy;
  sx = sx < 0.f ? 0.f : sx;
  int y0 = (int)sy, x0 = (int)sx;
  y0 = y0 > g.h - 1 ? g.h - 1 : y0;
  x0 = x0 > g.w - 1 ? g.w - 1 : x0;
  const int y1 = y0 + (y0 < g.h - 1 ? 1 : 0), x1 = x0 + (x0 < g.w - 1 ? 1 : 0);
  float ly = sy - (float)y0, lx = sx - (float)x0;
  ly = fminf(fmaxf(ly, 0.f), 1.f);
  lx = fminf(fmaxf(lx, 0.f), 1.f);
  const float hy = 1.f - ly, hx = 1.f - lx;
  const float v00 = p[y0 * g.w + x0], v01 = p[y0 * g.w + x1], v10 = p[y1 * g.w + x0], v11 = p[y1 * g.w + x1];
  const float t0 = __fadd_rn(__fmul_rn(hx, v00), __fmul_rn(lx, v01));
  const float t1 = __fadd_rn(__fmul_rn(hx, v10), __fmul_rn(lx, v11));
  return __fadd_rn(__fmul_rn(hy, t0), __fmul_rn(ly, t1));
}

// `nearest` resize target -> grid index: min(floor(dst * (in / out)), in - 1), float scale (UpSample.h)
__device__ __forceinline__ int nearest_src(int dst, float scale, int in_size) {
  const int s = (int)floorf((float)dst * scale);
  return s < in_size - 1 ? s : in_size - 1;
}

__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];  // fixed order: deterministic
  __syncthreads();
  return t;
}

__global__ __launch_bounds__(256) void instance_scores_kernel(const float* __restrict__ logits,
                                                              const int32_t* __restrict__ qidx,
                                                              float* __restrict__ sum_sig, float* __restrict__ cnt,
                                                              int Q, int K, Grid g) {
  __shared__ float red[4];
  const int bk = blockIdx.x, b = bk / K;
  const int q = qidx[bk];
  const float* p = logits + ((int64_t)b * Q + q) * g.h * g.w;
  float s = 0.f, c = 0.f;
  const int n = g.gh * g.gw;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int gy = i / g.gw, gx = i - gy * g.gw;
    const float v = grid_logit(p, g, gy, gx);
    if (v > 0.f) {
      s += 1.f / (1.f + __expf(-v));
      c += 1.f;
    }
  }
  s = block_sum(s, red);
  c = block_sum(c, red);
  if (threadIdx.x == 0) {
    sum_sig[bk] = s;
    cnt[bk] = c;
  }
}

__global__ __launch_bounds__(256) void instance_any_kernel(const float* __restrict__ logits,
                                                           const int32_t* __restrict__ qidx,
                                                           const uint8_t* __restrict__ cand, int32_t* __restrict__ any_out,
                                                           int Q, int K, Grid g, int Ho, int Wo) {
  const int bk = blockIdx.x, b = bk / K;
  if (!cand[bk]) {
    if (threadIdx.x == 0) any_out[bk] = 0;
    return;
  }
  const float* p = logits + ((int64_t)b * Q + qidx[bk]) * g.h * g.w;
  const float ny = (float)g.gh / (float)Ho, nx = (float)g.gw / (float)Wo;
  int found = 0;
  const int n = Ho * Wo;
  for (int i = threadIdx.x; i < n && !found; i += blockDim.x) {
    const int Y = i / Wo, X = i - Y * Wo;
    found = grid_logit(p, g, nearest_src(Y, ny, g.gh), nearest_src(X, nx, g.gw)) > 0.f;
  }
  const int any = __syncthreads_or(found);
  if (threadIdx.x == 0) any_out[bk] = any ? 1 : 0;
}

// kept_q (B, K): source query of the kept instance with id r (r < n_kept[b]), in the dependency's paint order
__global__ __launch_bounds__(256) void instance_segmentation_kernel(const float* __restrict__ logits,
                                                                    const int32_t* __restrict__ kept_q,
                                                                    const int32_t* __restrict__ n_kept,
                                                                    float* __restrict__ seg, int Q, int K, Grid g, int Ho,
                                                                    int Wo) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Ho * Wo) return;
  const int Y = i / Wo, X = i - Y * Wo;
  const int gy = nearest_src(Y, (float)g.gh / (float)Ho, g.gh), gx = nearest_src(X, (float)g.gw / (float)Wo, g.gw);
  const int n = n_kept[b];
  float id = -1.f;
  for (int r = n - 1; r >= 0; --r) {  // the last painted instance wins
    const float* p = logits + ((int64_t)b * Q + kept_q[b * K + r]) * g.h * g.w;
    if (grid_logit(p, g, gy, gx) > 0.f) {
      id = (float)r;
      break;
    }
  }
  seg[(int64_t)b * Ho * Wo + i] = id;
}

__global__ __launch_bounds__(256) void instance_maps_kernel(const float* __restrict__ logits,
                                                            const int32_t* __restrict__ kept_q, float* __restrict__ maps,
                                                            Grid g, int Ho, int Wo) {
  const int r = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Ho * Wo) return;
  const int Y = i / Wo, X = i - Y * Wo;
  const float* p = logits + (int64_t)kept_q[r] * g.h * g.w;
  const float v = grid_logit(p, g, nearest_src(Y, (float)g.gh / (float)Ho, g.gh), nearest_src(X, (float)g.gw / (float)Wo, g.gw));
  maps[(int64_t)r * Ho * Wo + i] = v > 0.f ? 1.f : 0.f;
}

int make_grid(Grid& g, int h, int w, int gh, int gw, const char* who) {
  WM2F_REQUIRE(h > 0 && w > 0 && gh > 0 && gw > 0, "%s: non-positive size", who);
  g.h = h;
  g.w = w;
  g.gh = gh;
  g.gw = gw;
  g.sh = (float)h / (float)gh;
  g.sw = (float)w / (float)gw;
  return WM2F_OK;
}

}  // namespace
}  // namespace wm2f

using namespace wm2f;

extern "C" int wm2f_instance_scores(const void* mask_logits, const int32_t* qidx, void* sum_sig, void* cnt, int B, int Q,
                                    int K, int h, int w, int gh, int gw, void* stream) {
  const char* who = "wm2f_instance_scores";
  WM2F_REQUIRE(mask_logits && qidx && sum_sig && cnt, "%s: null pointer", who);
  WM2F_REQUIRE(B > 0 && Q > 0 && K > 0, "%s: non-positive size", who);
  Grid g;
  if (int rc = make_grid(g, h, w, gh, gw, who)) return rc;
  hipLaunchKernelGGL(instance_scores_kernel, dim3(B * K), dim3(256), 0, (hipStream_t)stream, (const float*)mask_logits,
                     qidx, (float*)sum_sig, (float*)cnt, Q, K, g);
  WM2F_CHECK_LAUNCH(who);
  return WM2F_OK;
}

extern "C" int wm2f_instance_any(const void* mask_logits, const int32_t* qidx, const uint8_t* cand, int32_t* any_out, int B,
                                 int Q, int K, int h, int w, int gh, int gw, int Ho, int Wo, void* stream) {
  const char* who = "wm2f_instance_any";
  WM2F_REQUIRE(mask_logits && qidx && cand && any_out, "%s: null pointer", who);
  WM2F_REQUIRE(B > 0 && Q > 0 && K > 0 && Ho > 0 && Wo > 0, "%s: non-positive size", who);
  Grid g;
  if (int rc = make_grid(g, h, w, gh, gw, who)) return rc;
  hipLaunchKernelGGL(instance_any_kernel, dim3(B * K), dim3(256), 0, (hipStream_t)stream, (const float*)mask_logits, qidx,
                     cand, any_out, Q, K, g, Ho, Wo);
  WM2F_CHECK_LAUNCH(who);
  return WM2F_OK;
}

extern "C" int wm2f_instance_segmentation(const void* mask_logits, const int32_t* kept_q, const int32_t* n_kept,
                                          void* segmentation, int B, int Q, int K, int h, int w, int gh, int gw, int Ho,
                                          int Wo, void* stream) {
  const char* who = "wm2f_instance_segmentation";
  WM2F_REQUIRE(mask_logits && kept_q && n_kept && segmentation, "%s: null pointer", who);
  WM2F_REQUIRE(B > 0 && B < 65536 && Q > 0 && K > 0 && Ho > 0 && Wo > 0, "%s: bad size", who);
  Grid g;
  if (int rc = make_grid(g, h, w, gh, gw, who)) return rc;
  hipLaunchKernelGGL(instance_segmentation_kernel, dim3(ceil_div(Ho * Wo, 256), B), dim3(256), 0, (hipStream_t)stream,
                     (const float*)mask_logits, kept_q, n_kept, (float*)segmentation, Q, K, g, Ho, Wo);
  WM2F_CHECK_LAUNCH(who);
  return WM2F_OK;
}

extern "C" int wm2f_instance_maps(const void* image_logits, const int32_t* kept_q, int n, void* maps, int h, int w, int gh,
                                  int gw, int Ho, int Wo, void* stream) {
  const char* who = "wm2f_instance_maps";
  WM2F_REQUIRE(image_logits && kept_q && maps, "%s: null pointer", who);
  WM2F_REQUIRE(n > 0 && n < 65536 && Ho > 0 && Wo > 0, "%s: bad size", who);
  Grid g;
  if (int rc = make_grid(g, h, w, gh, gw, who)) return rc;
  hipLaunchKernelGGL(instance_maps_kernel, dim3(ceil_div(Ho * Wo, 256), n), dim3(256), 0, (hipStream_t)stream,
                     (const float*)image_logits, kept_q, (float*)maps, g, Ho, Wo);
  WM2F_CHECK_LAUNCH(who);
  return WM2F_OK;
}

// ---------------------------------------------------------------------------------------------------
// Label expansion on device: convert_segmentation_map_to_binary_masks
// (image_processing_mask2former.py:227-259 / image_processing_pil_mask2former.py:81-114) writes one binary mask per
// instance id present in the map.  The reference stores them as float (T, H, W) inside every .pt sample
// (datasets/dataset_utils.py:56-70) -- 64 MB per 1024 x 1024 image with 16 instances, the largest host-to-device
// copy of a step.  Shipping the (H, W) id map and expanding here moves 4 MB instead (1 MB as uint8 maps).
//   out[t][i] = (map[i] == ids[t]) as uint8; one pass over the map, T coalesced output streams.
namespace wm2f {
namespace {
__global__ __launch_bounds__(256) void labelmap_to_masks_kernel(const int32_t* __restrict__ map,
                                                                const int32_t* __restrict__ ids, uint8_t* __restrict__ out,
                                                                int64_t n, int T) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  int v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = i + k < n ? map[i + k] : INT32_MIN;
  for (int t = 0; t < T; ++t) {
    const int id = ids[t];
    if (i + 3 < n) {
      const uint32_t w = (uint32_t)(v[0] == id) | ((uint32_t)(v[1] == id) << 8) | ((uint32_t)(v[2] == id) << 16) |
                         ((uint32_t)(v[3] == id) << 24);
      *reinterpret_cast<uint32_t*>(out + (int64_t)t * n + i) = w;  // n % 4 == 0 is required for this store
    } else {
      for (int k = 0; k < 4 && i + k < n; ++k) out[(int64_t)t * n + i + k] = v[k] == id;
    }
  }
}
}  // namespace
}  // namespace wm2f

extern "C" int wm2f_labelmap_to_masks(const int32_t* label_map, const int32_t* ids, uint8_t* masks, int64_t n_pixels, int T,
                                      void* stream) {
  const char* who = "wm2f_labelmap_to_masks";
  WM2F_REQUIRE(label_map && ids && masks, "%s: null pointer", who);
  WM2F_REQUIRE(n_pixels > 0 && n_pixels % 4 == 0 && T > 0, "%s: need T > 0 and a pixel count divisible by 4", who);
  hipLaunchKernelGGL(labelmap_to_masks_kernel, dim3((unsigned)ceil_div64(n_pixels, 1024)), dim3(256), 0, (hipStream_t)stream,
                     label_map, ids, masks, n_pixels, T);
  WM2F_CHECK_LAUNCH(who);
  return WM2F_OK;
}
