// Fused HBM-bound passes around the stock GEMMs / convolutions.
//
//  bias_act:       y = act(x + bias[c] (+ residual))       NCHW, in place or out of place
//                  replaces the separate bias-add, residual-add and ReLU passes after a convolution whose
//                  BatchNorm has been folded into its weights (inference).
//  add_layernorm:  h = LayerNorm(x + res) * gamma + beta;  optionally also h + pos
//                  replaces residual add + nn.LayerNorm (+ the next layer's `hidden + pos`),
//                  transformers modeling_mask2former.py:1076-1078, :1086-1088.
//                  One wave per row; C must be 256 (64 lanes x float4): two-pass mean / variance in
//                  registers, exactly torch's formula (biased variance, eps inside the sqrt).
// Both move each byte once: read x (+res) and write y -> bound by HBM.
#include "common.h"

namespace wm2f {

__global__ __launch_bounds__(256) void bias_act_kernel(const float* __restrict__ x, const float* __restrict__ bias,
                                                       const float* __restrict__ res, float* __restrict__ y,
                                                       int64_t n4, int C, int HW4, int relu) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)((i / HW4) % C);
    const float b = bias[c];
    float4 v = reinterpret_cast<const float4*>(x)[i];
    v.x += b; v.y += b; v.z += b; v.w += b;
    if (res != nullptr) {
      const float4 r = reinterpret_cast<const float4*>(res)[i];
      v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    if (relu) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    reinterpret_cast<float4*>(y)[i] = v;
  }
}

// Row form for H*W/4 >= 256: a workgroup owns kBiasU * 256 float4 of ONE (image, channel) row, so the channel (and
// the bias) is a per-workgroup scalar -- no 64-bit division per element as in the grid-stride form above -- and each
// thread has kBiasU (x2 with a residual) 16-byte loads in flight before its first store.
constexpr int kBiasU = 4;
__global__ __launch_bounds__(256) void bias_act_rows_kernel(const float* __restrict__ x, const float* __restrict__ bias,
                                                            const float* __restrict__ res, float* __restrict__ y, int C,
                                                            int HW4, int chunks, int relu) {
  const int row = blockIdx.x / chunks, ch = blockIdx.x - row * chunks;
  const float b = bias[row % C];
  const int64_t base = (int64_t)row * HW4;
  const int i0 = ch * (256 * kBiasU) + threadIdx.x;
  const float4* xp = reinterpret_cast<const float4*>(x) + base;
  const float4* rp = reinterpret_cast<const float4*>(res) + base;
  float4 v[kBiasU], r[kBiasU];
#pragma unroll
  for (int u = 0; u < kBiasU; ++u) {
    const int i = i0 + u * 256;
    v[u] = i < HW4 ? xp[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (res != nullptr) {
#pragma unroll
    for (int u = 0; u < kBiasU; ++u) {
      const int i = i0 + u * 256;
      r[u] = i < HW4 ? rp[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
#pragma unroll
  for (int u = 0; u < kBiasU; ++u) {
    const int i = i0 + u * 256;
    float4 o = v[u];
    o.x += b; o.y += b; o.z += b; o.w += b;  // same order of additions as the grid-stride form: (x + bias) + residual
    if (res != nullptr) {
      o.x += r[u].x; o.y += r[u].y; o.z += r[u].z; o.w += r[u].w;
    }
    if (relu) {
      o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
    }
    if (i < HW4) reinterpret_cast<float4*>(y)[base + i] = o;
  }
}

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWave);
  return v;
}

__global__ __launch_bounds__(256) void add_layernorm256_kernel(const float* __restrict__ x,
                                                               const float* __restrict__ res,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta,
                                                               const float* __restrict__ pos, float* __restrict__ out,
                                                               float* __restrict__ out_pos, int64_t rows,
                                                               int64_t pos_rows, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float4 v = reinterpret_cast<const float4*>(x + row * 256)[lane];
  if (res != nullptr) {
    const float4 r = reinterpret_cast<const float4*>(res + row * 256)[lane];
    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
  }
  const float mean = wave_sum64(v.x + v.y + v.z + v.w) * (1.f / 256.f);
  const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
  const float var = wave_sum64(dx * dx + dy * dy + dz * dz + dw * dw) * (1.f / 256.f);
  const float rstd = rsqrtf(var + eps);
  const float4 g = reinterpret_cast<const float4*>(gamma)[lane], b = reinterpret_cast<const float4*>(beta)[lane];
  float4 o = make_float4(dx * rstd * g.x + b.x, dy * rstd * g.y + b.y, dz * rstd * g.z + b.z, dw * rstd * g.w + b.w);
  reinterpret_cast<float4*>(out + row * 256)[lane] = o;
  if (out_pos != nullptr) {
    const float4 p = reinterpret_cast<const float4*>(pos + (row % pos_rows) * 256)[lane];
    o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
    reinterpret_cast<float4*>(out_pos + row * 256)[lane] = o;
  }
}

}  // namespace wm2f

using namespace wm2f;

// out[b][i] = a[b][i] + p[i]: the positional embedding of a level added to its tokens for every image of the batch (the keys of
// the masked cross-attention, HF:1644-1650 `with_pos_embed`).  The stock broadcast add runs this at 1.1 TB/s (a strided kernel,
// one element per lane); here 16 bytes per lane, 4 independent pieces in flight.
namespace wm2f {
namespace {
__global__ __launch_bounds__(256) void add_broadcast_kernel(const float4* __restrict__ a, const float4* __restrict__ p,
                                                            float4* __restrict__ out, int64_t n4) {
  const int64_t base = (int64_t)blockIdx.y * n4;
  const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  float4 va[4], vp[4];
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (i0 + e < n4) {
      va[e] = a[base + i0 + e];
      vp[e] = p[i0 + e];
    }
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (i0 + e < n4) out[base + i0 + e] = make_float4(va[e].x + vp[e].x, va[e].y + vp[e].y, va[e].z + vp[e].z, va[e].w + vp[e].w);
}
}  // namespace
}  // namespace wm2f

extern "C" int wm2f_add_broadcast(const void* a, const void* p, void* out, int B, int64_t n, void* stream) {
  const char* who = "wm2f_add_broadcast";
  WM2F_REQUIRE(a && p && out, "%s: null pointer", who);
  WM2F_REQUIRE(B > 0 && B < 65536 && n > 0 && n % 4 == 0, "%s: B = %d, n = %lld (n must be a multiple of 4)", who, B, (long long)n);
  WM2F_REQUIRE((((uintptr_t)a | (uintptr_t)p | (uintptr_t)out) & 15) == 0, "%s: pointers must be 16-byte aligned", who);
  const int64_t n4 = n / 4, blocks = (n4 + 1023) / 1024;
  WM2F_REQUIRE(blocks < (int64_t(1) << 31), "%s: row too long", who);
  hipLaunchKernelGGL(wm2f::add_broadcast_kernel, dim3((unsigned)blocks, (unsigned)B), dim3(256), 0, (hipStream_t)stream, (const float4*)a,
                     (const float4*)p, (float4*)out, n4);
  WM2F_CHECK_LAUNCH(who);
  return WM2F_OK;
}

extern "C" int wm2f_bias_act(const void* x, const void* bias, const void* residual, void* y, int N, int C, int HW,
                             int relu, void* stream) {
  const char* who = "wm2f_bias_act";
  WM2F_REQUIRE(x && bias && y, "%s: null pointer", who);
  WM2F_REQUIRE(N > 0 && C > 0 && HW > 0, "%s: non-positive size", who);
  WM2F_REQUIRE(HW % 4 == 0, "%s: H*W=%d must be a multiple of 4", who, HW);
  const int64_t n4 = (int64_t)N * C * (HW / 4);
  const int HW4 = HW / 4;
  const int chunks = ceil_div(HW4, 256 * kBiasU);
  if (HW4 >= 256 && (int64_t)N * C * chunks < (int64_t(1) << 31)) {
    hipLaunchKernelGGL(bias_act_rows_kernel, dim3((unsigned)((int64_t)N * C * chunks)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)x, (const float*)bias, (const float*)residual, (float*)y, C, HW4, chunks, relu);
    WM2F_CHECK_LAUNCH(who);
    return WM2F_OK;
  }
  int64_t blocks = ceil_div64(n4, 256);
  if (blocks > 2048 * 4) blocks = 2048 * 4;  // grid-stride the rest
  hipLaunchKernelGGL(bias_act_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float*)x,
                     (const float*)bias, (const float*)residual, (float*)y, n4, C, HW / 4, relu);
  WM2F_CHECK_LAUNCH(who);
  return WM2F_OK;
}

extern "C" int wm2f_add_layernorm(const void* x, const void* residual, const void* gamma, const void* beta,
                                  const void* pos, void* out, void* out_plus_pos, int64_t rows, int C,
                                  int64_t pos_rows, float eps, void* stream) {
  const char* who = "wm2f_add_layernorm";
  WM2F_REQUIRE(x && gamma && beta && out, "%s: null pointer", who);
  WM2F_REQUIRE(C == 256, "%s: built for C = 256 only (got %d)", who, C);
  WM2F_REQUIRE(rows > 0, "%s: non-positive size", who);
  WM2F_REQUIRE((out_plus_pos == nullptr) || (pos != nullptr && pos_rows > 0), "%s: out_plus_pos needs pos", who);
  hipLaunchKernelGGL(add_layernorm256_kernel, dim3((unsigned)ceil_div64(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)x, (const float*)residual, (const float*)gamma, (const float*)beta,
                     (const float*)pos, (float*)out, (float*)out_plus_pos, rows, pos_rows > 0 ? pos_rows : 1, eps);
  WM2F_CHECK_LAUNCH(who);
  return WM2F_OK;
}

// ---------------------------------------------------------------------------------------------------
// tokens (B, S, C) rows [start, start + HW) of every image  ->  feature map (B, C, HW)   (inference)
// The pixel decoder hands its encoder output back to the FPN as NCHW maps (HF:1384-1391:
// `hidden[:, start:start+hw].transpose(1, 2).reshape(B, C, h, w)`); as a generic strided copy that is 0.9 ms for
// the finest level at config 2 (134 MB at ~300 GB/s).  Classic LDS tile transpose: 32 x 32 floats per tile,
// both sides coalesced.
namespace wm2f {
namespace {
__global__ __launch_bounds__(256) void tokens_to_nchw_kernel(const float* __restrict__ tok, float* __restrict__ out, int S,
                                                             int C, int start, int HW) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float* src = tok + ((int64_t)b * S + start) * C;
#pragma unroll
  for (int r = 0; r < 32; r += 8) {
    const int t = t0 + ty + r, c = c0 + tx;
    tile[ty + r][tx] = (t < HW && c < C) ? src[(int64_t)t * C + c] : 0.f;
  }
  __syncthreads();
  float* dst = out + (int64_t)b * C * HW;
#pragma unroll
  for (int r = 0; r < 32; r += 8) {
    const int c = c0 + ty + r, t = t0 + tx;
    if (c < C && t < HW) dst[(int64_t)c * HW + t] = tile[tx][ty + r];
  }
}
}  // namespace
}  // namespace wm2f

extern "C" int wm2f_tokens_to_nchw(const void* tokens, void* out, int B, int S, int C, int start, int HW, void* stream) {
  const char* who = "wm2f_tokens_to_nchw";
  WM2F_REQUIRE(tokens && out, "%s: null pointer", who);
  WM2F_REQUIRE(B > 0 && B < 65536 && S > 0 && C > 0 && HW > 0 && start >= 0 && start + HW <= S, "%s: bad size", who);
  WM2F_REQUIRE(wm2f::ceil_div(C, 32) < 65536, "%s: too many channels", who);
  hipLaunchKernelGGL(wm2f::tokens_to_nchw_kernel, dim3(wm2f::ceil_div(HW, 32), wm2f::ceil_div(C, 32), B), dim3(256), 0,
                     (hipStream_t)stream, (const float*)tokens, (float*)out, S, C, start, HW);
  WM2F_CHECK_LAUNCH(who);
  return WM2F_OK;
}

// ---------------------------------------------------------------------------------------------------
// GroupNorm tail of the FPN step (inference): HF:1395-1405 does, on the stride-4 level (B, 256, H, W),
//     lat = GroupNorm(conv1x1(feat));  out = lat + interpolate(prev, size=(H, W), mode="bilinear");
//     y   = ReLU(GroupNorm(conv3x3(out)))
// As stock ops that is a normalise pass, an upsample pass (0.87 ms at config 2: 0.8 TB/s), an add and a clamp, each
// reading and writing the 537 MB map.  Here: group statistics in one reading pass (sum and sum of squares per float4
// chunk in fp32, accumulated across chunks in fp64), then ONE pass that normalises, adds the bilinear sample of the
// coarser map (align_corners = False, PyTorch's source-index rule) and applies the ReLU.
namespace wm2f {
namespace {
constexpr int kGnU = 4;  // float4 per thread

// stats[(b * G + g) * 2 + {0, 1}] += (sum, sum of squares) of this workgroup's chunk of the group's contiguous span
// `bias` (nullable, per channel) is added to x first: the statistics of conv(x) + bias without a separate bias pass.
__global__ __launch_bounds__(256) void group_stats_kernel(const float* __restrict__ x, const float* __restrict__ bias,
                                                          double* __restrict__ stats, int64_t span4, int chunks, int G, int cpg,
                                                          int HW4) {
  const int row = blockIdx.x / chunks, ch = blockIdx.x - row * chunks;  // row = b * G + g
  const float4* xp = reinterpret_cast<const float4*>(x) + (int64_t)row * span4;
  float s = 0.f, ss = 0.f;
  const int64_t i0 = (int64_t)ch * (256 * kGnU) + threadIdx.x;
  float4 v[kGnU];
#pragma unroll
  for (int u = 0; u < kGnU; ++u) {
    const int64_t i = i0 + u * 256;
    v[u] = i < span4 ? xp[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias != nullptr && i < span4) {
      const float bb = bias[(row % G) * cpg + (int)(i / HW4)];
      v[u].x += bb; v[u].y += bb; v[u].z += bb; v[u].w += bb;
    }
  }
#pragma unroll
  for (int u = 0; u < kGnU; ++u) {
    s += (v[u].x + v[u].y) + (v[u].z + v[u].w);
    ss += (v[u].x * v[u].x + v[u].y * v[u].y) + (v[u].z * v[u].z + v[u].w * v[u].w);
  }
  s = wave_sum64(s);
  ss = wave_sum64(ss);
  __shared__ float red[2][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    red[0][wave] = s;
    red[1][wave] = ss;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(stats + 2 * row, (double)red[0][0] + (double)red[0][1] + (double)red[0][2] + (double)red[0][3]);
    atomicAdd(stats + 2 * row + 1, (double)red[1][0] + (double)red[1][1] + (double)red[1][2] + (double)red[1][3]);
  }
}

// PyTorch's bilinear source index for align_corners = False (UpSample.h area_pixel_compute_source_index):
// src = max(scale * (dst + 0.5) - 0.5, 0);  i0 = floor(src);  i1 = i0 + (i0 < size - 1);  lambda1 = src - i0
struct UpIdx { int i0, i1; float l0, l1; };
__device__ __forceinline__ UpIdx up_index(int dst, float scale, int size) {
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  src = src < 0.f ? 0.f : src;
  UpIdx r;
  r.i0 = (int)src;
  if (r.i0 > size - 1) r.i0 = size - 1;
  r.i1 = r.i0 + (r.i0 < size - 1 ? 1 : 0);
  r.l1 = src - (float)r.i0;
  r.l0 = 1.f - r.l1;
  return r;
}

// y[b][c][p] = act((x - mean) * rstd * gamma[c] + beta[c] (+ bilinear(up[b][c], p)));  a workgroup = one chunk of one (b, c) row
__global__ __launch_bounds__(256) void group_norm_apply_kernel(const float* __restrict__ x, const double* __restrict__ stats,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const float* __restrict__ up, float* __restrict__ y, int C, int G,
                                                               int H, int W, int Hs, int Ws, float eps, int relu, int chunks) {
  const int row = blockIdx.x / chunks, ch = blockIdx.x - row * chunks;  // row = b * C + c
  const int b = row / C, c = row - b * C;
  const int cpg = C / G, g = c / cpg;
  const int HW4 = (H * W) >> 2, W4 = W >> 2;
  const double n = (double)cpg * (double)H * (double)W;
  const double mean_d = stats[2 * (b * G + g)] / n;
  double var_d = stats[2 * (b * G + g) + 1] / n - mean_d * mean_d;
  var_d = var_d < 0.0 ? 0.0 : var_d;
  const float rstd = (float)(1.0 / sqrt(var_d + (double)eps));
  const float scale = rstd * gamma[c];
  const float shift = beta[c] - (float)mean_d * scale;
  const float4* xp = reinterpret_cast<const float4*>(x) + (int64_t)row * HW4;
  float4* yp = reinterpret_cast<float4*>(y) + (int64_t)row * HW4;
  const float* sp = up != nullptr ? up + (int64_t)row * Hs * Ws : nullptr;
  const float sy = (float)Hs / (float)H, sx = (float)Ws / (float)W;
  const int i0 = ch * (256 * kGnU) + threadIdx.x;
  float4 v[kGnU];
#pragma unroll
  for (int u = 0; u < kGnU; ++u) {
    const int i = i0 + u * 256;
    v[u] = i < HW4 ? xp[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int u = 0; u < kGnU; ++u) {
    const int i = i0 + u * 256;
    if (i >= HW4) continue;
    float o[4] = {v[u].x * scale + shift, v[u].y * scale + shift, v[u].z * scale + shift, v[u].w * scale + shift};
    if (sp != nullptr) {
      const int py = i / W4, px = (i - py * W4) * 4;
      const UpIdx iy = up_index(py, sy, Hs);
      const float* r0 = sp + (int64_t)iy.i0 * Ws;
      const float* r1 = sp + (int64_t)iy.i1 * Ws;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const UpIdx ix = up_index(px + k, sx, Ws);
        o[k] += iy.l0 * (ix.l0 * r0[ix.i0] + ix.l1 * r0[ix.i1]) + iy.l1 * (ix.l0 * r1[ix.i0] + ix.l1 * r1[ix.i1]);
      }
    }
    if (relu) {
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = fmaxf(o[k], 0.f);
    }
    yp[i] = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// tokens[b][start + p][c] = GroupNorm(x + bias)[b][c][p]: the input projections of the pixel decoder (HF:1341-1357:
// Conv2d 1x1 + GroupNorm per level, then flatten(2).transpose(1, 2) and a concatenation over levels) written straight
// into the (B, S, C) token buffer.  32 x 32 LDS tile transpose, both sides coalesced.
__global__ __launch_bounds__(256) void group_norm_tokens_kernel(const float* __restrict__ x, const float* __restrict__ bias,
                                                                const double* __restrict__ stats, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float* __restrict__ tok, int C, int G,
                                                                int HW, int S, int start, float eps) {
  __shared__ float tile[32][33];
  __shared__ float sc[32], sh[32];
  const int b = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int cpg = C / G;
  if (threadIdx.x < 32) {
    const int c = c0 + threadIdx.x;
    float scale = 0.f, shift = 0.f;
    if (c < C) {
      const int g = c / cpg;
      const double n = (double)cpg * (double)HW;
      const double mean_d = stats[2 * (b * G + g)] / n;
      double var_d = stats[2 * (b * G + g) + 1] / n - mean_d * mean_d;
      var_d = var_d < 0.0 ? 0.0 : var_d;
      scale = (float)(1.0 / sqrt(var_d + (double)eps)) * gamma[c];
      shift = beta[c] + ((bias != nullptr ? bias[c] : 0.f) - (float)mean_d) * scale;
    }
    sc[threadIdx.x] = scale;
    sh[threadIdx.x] = shift;
  }
  __syncthreads();
  const float* src = x + (int64_t)b * C * HW;
#pragma unroll
  for (int r = 0; r < 32; r += 8) {
    const int c = c0 + ty + r, p = p0 + tx;
    tile[ty + r][tx] = (c < C && p < HW) ? src[(int64_t)c * HW + p] * sc[ty + r] + sh[ty + r] : 0.f;
  }
  __syncthreads();
  float* dst = tok + ((int64_t)b * S + start) * C;
#pragma unroll
  for (int r = 0; r < 32; r += 8) {
    const int p = p0 + ty + r, c = c0 + tx;
    if (p < HW && c < C) dst[(int64_t)p * C + c] = tile[tx][ty + r];
  }
}
}  // namespace
}  // namespace wm2f

extern "C" int wm2f_group_norm_act(const void* x, const void* gamma, const void* beta, const void* up, void* y,
                                   void* stats_ws, int B, int C, int G, int H, int W, int Hs, int Ws, float eps, int relu,
                                   void* stream) {
  using namespace wm2f;
  const char* who = "wm2f_group_norm_act";
  WM2F_REQUIRE(x && gamma && beta && y && stats_ws, "%s: null pointer", who);
  WM2F_REQUIRE(B > 0 && C > 0 && G > 0 && H > 0 && W > 0, "%s: non-positive size", who);
  WM2F_REQUIRE(C % G == 0, "%s: C=%d is not a multiple of G=%d", who, C, G);
  WM2F_REQUIRE(W % 4 == 0, "%s: W=%d must be a multiple of 4", who, W);
  WM2F_REQUIRE(up == nullptr || (Hs > 0 && Ws > 0), "%s: the upsampled operand needs its size", who);
  const int64_t span4 = (int64_t)(C / G) * H * W / 4;
  const int64_t HW4 = (int64_t)H * W / 4;
  const int64_t s_chunks = ceil_div64(span4, 256 * kGnU), a_chunks = ceil_div64(HW4, 256 * kGnU);
  WM2F_REQUIRE((int64_t)B * G * s_chunks < (int64_t(1) << 31) && (int64_t)B * C * a_chunks < (int64_t(1) << 31) &&
                   HW4 < (int64_t(1) << 30),
               "%s: sizes exceed the grid limits", who);
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(stats_ws, 0, sizeof(double) * 2 * B * G, st);
  WM2F_REQUIRE(e == hipSuccess, "%s: memset failed: %s", who, hipGetErrorString(e));
  hipLaunchKernelGGL(group_stats_kernel, dim3((unsigned)((int64_t)B * G * s_chunks)), dim3(256), 0, st, (const float*)x,
                     (const float*)nullptr, (double*)stats_ws, span4, (int)s_chunks, G, C / G, (int)HW4);
  WM2F_CHECK_LAUNCH(who);
  hipLaunchKernelGGL(group_norm_apply_kernel, dim3((unsigned)((int64_t)B * C * a_chunks)), dim3(256), 0, st, (const float*)x,
                     (const double*)stats_ws, (const float*)gamma, (const float*)beta, (const float*)up, (float*)y, C, G, H, W,
                     Hs, Ws, eps, relu, (int)a_chunks);
  WM2F_CHECK_LAUNCH(who);
  return WM2F_OK;
}

extern "C" int wm2f_group_norm_tokens(const void* x, const void* bias, const void* gamma, const void* beta, void* tokens,
                                      void* stats_ws, int B, int C, int G, int HW, int S, int start, float eps, void* stream) {
  using namespace wm2f;
  const char* who = "wm2f_group_norm_tokens";
  WM2F_REQUIRE(x && gamma && beta && tokens && stats_ws, "%s: null pointer", who);
  WM2F_REQUIRE(B > 0 && B < 65536 && C > 0 && G > 0 && HW > 0 && S > 0, "%s: bad size", who);
  WM2F_REQUIRE(C % G == 0, "%s: C=%d is not a multiple of G=%d", who, C, G);
  WM2F_REQUIRE(HW % 4 == 0, "%s: H*W=%d must be a multiple of 4", who, HW);
  WM2F_REQUIRE(start >= 0 && (int64_t)start + HW <= S, "%s: rows [%d, %d) leave the token buffer of %d rows", who, start,
               start + HW, S);
  const int64_t span4 = (int64_t)(C / G) * HW / 4;
  const int64_t s_chunks = ceil_div64(span4, 256 * kGnU);
  WM2F_REQUIRE((int64_t)B * G * s_chunks < (int64_t(1) << 31) && ceil_div(C, 32) < 65536 && span4 < (int64_t(1) << 31),
               "%s: sizes exceed the grid limits", who);
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(stats_ws, 0, sizeof(double) * 2 * B * G, st);
  WM2F_REQUIRE(e == hipSuccess, "%s: memset failed: %s", who, hipGetErrorString(e));
  hipLaunchKernelGGL(group_stats_kernel, dim3((unsigned)((int64_t)B * G * s_chunks)), dim3(256), 0, st, (const float*)x,
                     (const float*)bias, (double*)stats_ws, span4, (int)s_chunks, G, C / G, HW / 4);
  WM2F_CHECK_LAUNCH(who);
  hipLaunchKernelGGL(group_norm_tokens_kernel, dim3(ceil_div(HW, 32), ceil_div(C, 32), B), dim3(256), 0, st, (const float*)x,
                     (const float*)bias, (const double*)stats_ws, (const float*)gamma, (const float*)beta, (float*)tokens, C, G,
                     HW, S, start, eps);
  WM2F_CHECK_LAUNCH(who);
  return WM2F_OK;
}

// ---------------------------------------------------------------------------------------------------
// Stem tail of the ResNet backbone (inference): y = MaxPool2d(3, stride 2, pad 1)(ReLU(x + bias[c])) in one pass.
// As two passes the full-resolution map (537 MB at config 2) is written and read back; max commutes with the monotone
// bias + ReLU, so the pool runs on the raw convolution output and the epilogue on the quarter-size result.
// H, W even, W % 8 == 0: a thread produces 4 adjacent outputs from 3 rows of (1 + 8) inputs (two aligned float4 and
// the left neighbour); rows above / left of the map count as -inf, exactly as PyTorch pads.
namespace wm2f {
namespace {
__global__ __launch_bounds__(256) void bias_relu_maxpool_kernel(const float* __restrict__ x, const float* __restrict__ bias,
                                                                float* __restrict__ y, int C, int H, int W, int64_t total4) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= total4) return;
  const int OW4 = W >> 3, OH = H >> 1;
  const int ox4 = (int)(t % OW4);
  const int64_t r = t / OW4;
  const int oy = (int)(r % OH);
  const int64_t nc = r / OH;
  const float* xp = x + nc * H * W + (int64_t)(8 * ox4);
  const float NEG = -INFINITY;
  float m[4] = {NEG, NEG, NEG, NEG};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int iy = 2 * oy - 1 + k;
    if (iy < 0) continue;  // iy <= H - 1 always (H even)
    const float* row = xp + (int64_t)iy * W;
    const float4 a = *reinterpret_cast<const float4*>(row), b = *reinterpret_cast<const float4*>(row + 4);
    const float left = ox4 > 0 ? row[-1] : NEG;
    m[0] = fmaxf(m[0], fmaxf(left, fmaxf(a.x, a.y)));
    m[1] = fmaxf(m[1], fmaxf(a.y, fmaxf(a.z, a.w)));
    m[2] = fmaxf(m[2], fmaxf(a.w, fmaxf(b.x, b.y)));
    m[3] = fmaxf(m[3], fmaxf(b.y, fmaxf(b.z, b.w)));
  }
  const float bb = bias[(int)(nc % C)];
  reinterpret_cast<float4*>(y)[t] = make_float4(fmaxf(m[0] + bb, 0.f), fmaxf(m[1] + bb, 0.f), fmaxf(m[2] + bb, 0.f), fmaxf(m[3] + bb, 0.f));
}
}  // namespace
}  // namespace wm2f

extern "C" int wm2f_bias_relu_maxpool(const void* x, const void* bias, void* y, int N, int C, int H, int W, void* stream) {
  using namespace wm2f;
  const char* who = "wm2f_bias_relu_maxpool";
  WM2F_REQUIRE(x && bias && y, "%s: null pointer", who);
  WM2F_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0, "%s: non-positive size", who);
  WM2F_REQUIRE(H % 2 == 0 && W % 8 == 0, "%s: needs H even and W %% 8 == 0 (got %d x %d)", who, H, W);
  const int64_t total4 = (int64_t)N * C * (H / 2) * (W / 8);
  WM2F_REQUIRE(ceil_div64(total4, 256) < (int64_t(1) << 31), "%s: sizes exceed the grid limits", who);
  hipLaunchKernelGGL(bias_relu_maxpool_kernel, dim3((unsigned)ceil_div64(total4, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)x, (const float*)bias, (float*)y, C, H, W, total4);
  WM2F_CHECK_LAUNCH(who);
  return WM2F_OK;
}

// ---------------------------------------------------------------------------------------------------
// Bilinear resize of an NCHW map (align_corners = False, PyTorch's source-index rule, no antialiasing), one thread per
// 4 adjacent outputs.  Used to bring the mask features to the three attention-mask resolutions once per forward
// (modeling.MaskPredictor.attention_mask_only); the stock kernel loops over batch x channels inside each thread of an
// output-pixel grid and takes milliseconds for a 256^2 -> 32^2 resize of 2048 maps.
namespace wm2f {
namespace {
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W,
                                                              int Ho, int Wo, int64_t total4) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= total4) return;
  const int Wo4 = Wo >> 2;
  const int ox = (int)(t % Wo4) * 4;
  const int64_t r = t / Wo4;
  const int oy = (int)(r % Ho);
  const int64_t nc = r / Ho;
  const float sy = (float)H / (float)Ho, sx = (float)W / (float)Wo;
  const UpIdx iy = up_index(oy, sy, H);
  const float* r0 = x + nc * H * W + (int64_t)iy.i0 * W;
  const float* r1 = x + nc * H * W + (int64_t)iy.i1 * W;
  float o[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const UpIdx ix = up_index(ox + k, sx, W);
    o[k] = iy.l0 * (ix.l0 * r0[ix.i0] + ix.l1 * r0[ix.i1]) + iy.l1 * (ix.l0 * r1[ix.i0] + ix.l1 * r1[ix.i1]);
  }
  reinterpret_cast<float4*>(y)[t] = make_float4(o[0], o[1], o[2], o[3]);
}
}  // namespace
}  // namespace wm2f

extern "C" int wm2f_resize_bilinear(const void* x, void* y, int NC, int H, int W, int Ho, int Wo, void* stream) {
  using namespace wm2f;
  const char* who = "wm2f_resize_bilinear";
  WM2F_REQUIRE(x && y, "%s: null pointer", who);
  WM2F_REQUIRE(NC > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "%s: non-positive size", who);
  WM2F_REQUIRE(Wo % 4 == 0, "%s: output width %d must be a multiple of 4", who, Wo);
  const int64_t total4 = (int64_t)NC * Ho * (Wo / 4);
  WM2F_REQUIRE(ceil_div64(total4, 256) < (int64_t(1) << 31), "%s: sizes exceed the grid limits", who);
  hipLaunchKernelGGL(resize_bilinear_kernel, dim3((unsigned)ceil_div64(total4, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)x, (float*)y, H, W, Ho, Wo, total4);
  WM2F_CHECK_LAUNCH(who);
  return WM2F_OK;
}

// ---------------------------------------------------------------------------------------------------
// The three attention-mask resolutions in ONE pass over the mask features.  For output sizes that are exactly 1/2, 1/4 and
// 1/8 of the input, PyTorch's source-index rule gives src = s * (dst + 0.5) - 0.5 = s * dst + (s - 1) / 2: the two taps
// are input rows / columns 2d, 2d+1 (s = 2), 4d+1, 4d+2 (s = 4), 8d+3, 8d+4 (s = 8), all with weights exactly 0.5 -- each
// output is the mean of a 2 x 2 block, bit for bit what wm2f_resize_bilinear (and torch) computes, because scaling by 0.5
// is exact.  A thread owns 4 consecutive columns x 8 rows of one map (eight fully coalesced 16-byte loads) and writes 2 x 4
// outputs of the half-size map, 1 x 2 of the quarter-size map and, together with its right-hand neighbour (columns
// 8d+3 | 8d+4 sit in different threads: one DPP move), 1 of the eighth-size map.  HBM-bound: reads the map once
// (537 MB at config 2), writes 21/64 of it.
namespace wm2f {
namespace {
__device__ __forceinline__ float mean4(float a, float b, float c, float d) { return 0.5f * (0.5f * a + 0.5f * b) + 0.5f * (0.5f * c + 0.5f * d); }

__global__ __launch_bounds__(256) void resize_pyramid_kernel(const float* __restrict__ x, float* __restrict__ y2, float* __restrict__ y4,
                                                             float* __restrict__ y8, int H, int W, int64_t total) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;  // one thread per (map, 8-row block, 4-column block)
  const bool live = t < total;
  const int W4 = W >> 2, H8 = H >> 3;
  const int64_t tt = live ? t : 0;
  const int cx = (int)(tt % W4);
  const int64_t r = tt / W4;
  const int by = (int)(r % H8);
  const int64_t nc = r / H8;
  const float* src = x + (nc * H + (int64_t)by * 8) * W + cx * 4;
  float4 v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const float4*>(src + (int64_t)i * W);
  const int W2 = W >> 1, Wq = W >> 2, We = W >> 3;
  if (live) {
    float* d2 = y2 + (nc * (H >> 1) + (int64_t)by * 4) * W2 + cx * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<float2*>(d2 + (int64_t)i * W2) = make_float2(mean4(v[2 * i].x, v[2 * i].y, v[2 * i + 1].x, v[2 * i + 1].y),
                                                                    mean4(v[2 * i].z, v[2 * i].w, v[2 * i + 1].z, v[2 * i + 1].w));
    float* d4 = y4 + (nc * (H >> 2) + (int64_t)by * 2) * Wq + cx;
    d4[0] = mean4(v[1].y, v[1].z, v[2].y, v[2].z);
    d4[Wq] = mean4(v[5].y, v[5].z, v[6].y, v[6].z);
  }
  // eighth size: rows 8d+3, 8d+4 = v[3], v[4]; columns 8d+3 (this thread's .w when cx is even) and 8d+4 (the right-hand
  // neighbour's .x).  W % 8 == 0 keeps an even thread and its neighbour in one row; 256 | blockDim keeps them in one wave.
  const float nx3 = __shfl_down(v[3].x, 1, 64), nx4 = __shfl_down(v[4].x, 1, 64);
  if (live && (cx & 1) == 0) y8[(nc * (H >> 3) + by) * We + (cx >> 1)] = mean4(v[3].w, nx3, v[4].w, nx4);
}
}  // namespace
}  // namespace wm2f

extern "C" int wm2f_resize_pyramid(const void* x, void* y2, void* y4, void* y8, int NC, int H, int W, void* stream) {
  using namespace wm2f;
  const char* who = "wm2f_resize_pyramid";
  WM2F_REQUIRE(x && y2 && y4 && y8, "%s: null pointer", who);
  WM2F_REQUIRE(NC > 0 && H > 0 && W > 0, "%s: non-positive size", who);
  WM2F_REQUIRE(H % 8 == 0 && W % 8 == 0, "%s: needs H and W divisible by 8 (got %d x %d)", who, H, W);
  const int64_t total = (int64_t)NC * (H / 8) * (W / 4);
  WM2F_REQUIRE(ceil_div64(total, 256) < (int64_t(1) << 31), "%s: sizes exceed the grid limits", who);
  hipLaunchKernelGGL(resize_pyramid_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)x, (float*)y2, (float*)y4, (float*)y8, H, W, total);
  WM2F_CHECK_LAUNCH(who);
  return WM2F_OK;
}
