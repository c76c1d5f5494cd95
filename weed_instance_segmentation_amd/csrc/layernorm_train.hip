// Residual add + LayerNorm of the pixel decoder's encoder layers for the TRAIN step, with its backward
// (HF:1076-1078 `hidden = layer_norm(hidden + attn)`, :1086-1088 `hidden = layer_norm(hidden + ffn)`, and the next layer's
// `hidden + pos` of HF:972), C = 256 features, rows = batch x tokens (344 064 at config 2).
//
// Why: in the stock form one encoder layer spends, around each of its two LayerNorms, a residual add (fp32), the LayerNorm,
// the next layer's `hidden + pos` add and -- under bf16 autocast -- a cast of every tensor a Linear consumes; backward, the
// LayerNorm gradient runs as two library kernels (695 us per LayerNorm at config 2: cuComputePartGradGammaBeta +
// cuComputeGradInput) behind casts and gradient-accumulation adds of the same 352-MB tensors.  All of it is HBM-bound
// elementwise / row-reduction work on the same rows.  Here:
//   forward  y = LayerNorm(x + res) in ONE pass that also writes what the consumers read: y in bf16 (the next Linear's operand
//            under autocast) and y + pos in bf16 / fp32 (the next layer's sampling-offset / attention-weight projection input),
//            plus (mean, rstd) per row;
//   backward ONE pass: the up-to-three gradient streams of those outputs are summed in registers, d(x + res) is written once in
//            fp32 (the residual stream) and once in x's dtype, and the gamma / beta gradients are accumulated per workgroup
//            (fixed row ranges) and added in workgroup order by a second kernel -- deterministic, no atomics.
// One wave per row, a lane owns 4 consecutive features; two-pass mean / variance in registers (biased variance, eps inside the
// square root: torch's formula).  Bound: HBM -- every operand byte moves once.
#include "common.h"

namespace wm2f {
namespace {

typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWave);
  return v;
}
__device__ __forceinline__ float4 ld4(const void* p, int bf16, int64_t row, int lane) {  // 4 features of a 256-wide row
  if (bf16) {
    const bf16x4_t v = reinterpret_cast<const bf16x4_t*>(reinterpret_cast<const unsigned short*>(p) + row * 256)[lane];
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
  }
  return reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + row * 256)[lane];
}
__device__ __forceinline__ void st4(void* p, int bf16, int64_t row, int lane, float4 v) {
  if (bf16) {
    bf16x4_t o;
    o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
    reinterpret_cast<bf16x4_t*>(reinterpret_cast<unsigned short*>(p) + row * 256)[lane] = o;
  } else {
    reinterpret_cast<float4*>(reinterpret_cast<float*>(p) + row * 256)[lane] = v;
  }
}

struct LnFwdArgs {
  const void* x;      // (rows, 256) fp32 or bf16
  const float* res;   // (rows, 256) fp32 or null
  const float* gamma;
  const float* beta;
  const float* pos;   // (pos_rows, 256) fp32 or null
  float* y;           // (rows, 256) fp32
  void* y_lp;         // (rows, 256) bf16 or null
  void* yp;           // (rows, 256) = y + pos, bf16 / fp32, or null
  float* stats;       // (rows, 2) = (mean, rstd)
  long long rows, pos_rows;
  float eps, clamp;  // clamp > 0: y is limited to [-clamp, clamp] (NaN stays NaN) -- the encoder layer's overflow guard, HF:1090-1093
  int x_bf16, yp_bf16;
};

__global__ __launch_bounds__(256) void add_layernorm_train_fwd_kernel(LnFwdArgs a) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.rows) return;
  float4 v = ld4(a.x, a.x_bf16, row, lane);
  if (a.res != nullptr) {
    const float4 r = reinterpret_cast<const float4*>(a.res + row * 256)[lane];
    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
  }
  const float mean = wsum(v.x + v.y + v.z + v.w) * (1.f / 256.f);
  const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
  const float var = wsum(dx * dx + dy * dy + dz * dz + dw * dw) * (1.f / 256.f);
  const float rstd = rsqrtf(var + a.eps);
  const float4 g = reinterpret_cast<const float4*>(a.gamma)[lane], b = reinterpret_cast<const float4*>(a.beta)[lane];
  float4 o = make_float4(dx * rstd * g.x + b.x, dy * rstd * g.y + b.y, dz * rstd * g.z + b.z, dw * rstd * g.w + b.w);
  if (a.clamp > 0.f) {  // comparisons, not fmin / fmax: a NaN must stay a NaN, as torch.clamp leaves it
    const float c = a.clamp;
    o.x = o.x > c ? c : (o.x < -c ? -c : o.x);
    o.y = o.y > c ? c : (o.y < -c ? -c : o.y);
    o.z = o.z > c ? c : (o.z < -c ? -c : o.z);
    o.w = o.w > c ? c : (o.w < -c ? -c : o.w);
  }
  reinterpret_cast<float4*>(a.y + row * 256)[lane] = o;
  if (a.y_lp != nullptr) st4(a.y_lp, 1, row, lane, o);
  if (a.yp != nullptr) {
    const float4 p = reinterpret_cast<const float4*>(a.pos + (row % a.pos_rows) * 256)[lane];
    st4(a.yp, a.yp_bf16, row, lane, make_float4(o.x + p.x, o.y + p.y, o.z + p.z, o.w + p.w));
  }
  if (lane == 0) *reinterpret_cast<float2*>(a.stats + row * 2) = make_float2(mean, rstd);
}

struct LnBwdArgs {
  const void* x;
  const float* res;
  const float* gamma;
  const float* stats;
  const float* gy;    // fp32 or null
  const void* gy_lp;  // bf16 or null
  const void* gyp;    // bf16 / fp32 or null
  float* dres;        // (rows, 256) fp32: d(x + res), or null
  void* dx;           // the same in x's dtype, or null
  float* ws;          // [blocks][2][256] partial (dgamma, dbeta)
  long long rows;
  int rows_per_block, x_bf16, gyp_bf16, dx_bf16;
};

__global__ __launch_bounds__(256) void add_layernorm_train_bwd_kernel(LnBwdArgs a) {
  __shared__ float part[4][2][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long r0 = (long long)blockIdx.x * a.rows_per_block;
  long long r1 = r0 + a.rows_per_block;
  if (r1 > a.rows) r1 = a.rows;
  const float4 g4 = reinterpret_cast<const float4*>(a.gamma)[lane];
  float4 dg = make_float4(0.f, 0.f, 0.f, 0.f), db = dg;
  for (long long row = r0 + wave; row < r1; row += 4) {
    float4 v = ld4(a.x, a.x_bf16, row, lane);
    if (a.res != nullptr) {
      const float4 r = reinterpret_cast<const float4*>(a.res + row * 256)[lane];
      v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    const float2 st = *reinterpret_cast<const float2*>(a.stats + row * 2);
    const float4 xh = make_float4((v.x - st.x) * st.y, (v.y - st.x) * st.y, (v.z - st.x) * st.y, (v.w - st.x) * st.y);
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.gy != nullptr) g = reinterpret_cast<const float4*>(a.gy + row * 256)[lane];
    if (a.gy_lp != nullptr) {
      const float4 t = ld4(a.gy_lp, 1, row, lane);
      g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w;
    }
    if (a.gyp != nullptr) {
      const float4 t = ld4(a.gyp, a.gyp_bf16, row, lane);
      g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w;
    }
    dg.x += g.x * xh.x; dg.y += g.y * xh.y; dg.z += g.z * xh.z; dg.w += g.w * xh.w;
    db.x += g.x; db.y += g.y; db.z += g.z; db.w += g.w;
    const float4 gg = make_float4(g.x * g4.x, g.y * g4.y, g.z * g4.z, g.w * g4.w);
    const float s1 = wsum(gg.x + gg.y + gg.z + gg.w) * (1.f / 256.f);
    const float s2 = wsum(gg.x * xh.x + gg.y * xh.y + gg.z * xh.z + gg.w * xh.w) * (1.f / 256.f);
    const float4 d = make_float4(st.y * (gg.x - s1 - xh.x * s2), st.y * (gg.y - s1 - xh.y * s2), st.y * (gg.z - s1 - xh.z * s2),
                                 st.y * (gg.w - s1 - xh.w * s2));
    if (a.dres != nullptr) reinterpret_cast<float4*>(a.dres + row * 256)[lane] = d;
    if (a.dx != nullptr) st4(a.dx, a.dx_bf16, row, lane, d);
  }
  // the four waves' partial (dgamma, dbeta) in wave order
  *reinterpret_cast<float4*>(&part[wave][0][4 * lane]) = dg;
  *reinterpret_cast<float4*>(&part[wave][1][4 * lane]) = db;
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 256) {
    const int which = i >> 8, c = i & 255;
    a.ws[(size_t)blockIdx.x * 512 + i] = ((part[0][which][c] + part[1][which][c]) + part[2][which][c]) + part[3][which][c];
  }
}

// dgamma[c] / dbeta[c] = sum over the workgroups' partials in a FIXED order: one workgroup per output (512 of them), thread t adds
// partials t, t + 256, ... in ascending order, then the 256 sums are added pairwise in a fixed tree through LDS.  (Eight workgroups
// walking 2 688 partials each were latency-bound: 138 us.)
__global__ __launch_bounds__(256) void add_layernorm_train_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dgamma,
                                                                         float* __restrict__ dbeta, int blocks) {
  __shared__ float part[256];
  const int c = blockIdx.x, t = threadIdx.x;  // c in [0, 512): dgamma 0..255, dbeta 256..511
  float s = 0.f;
  for (int i = t; i < blocks; i += 256) s += ws[(size_t)i * 512 + c];
  part[t] = s;
  __syncthreads();
#pragma unroll
  for (int w = 128; w >= 1; w >>= 1) {
    if (t < w) part[t] += part[t + w];
    __syncthreads();
  }
  if (t == 0) {
    if (c < 256) dgamma[c] = part[0];
    else dbeta[c - 256] = part[0];
  }
}

constexpr int kLnRowsPerBlock = 128;

}  // namespace
}  // namespace wm2f

using namespace wm2f;

extern "C" int64_t wm2f_add_layernorm_train_workspace(int64_t rows) {
  return rows > 0 ? ((rows + kLnRowsPerBlock - 1) / kLnRowsPerBlock) * 512 * 4 : 0;
}

extern "C" int wm2f_add_layernorm_train_fwd(const void* x, int x_dtype, const void* residual, const void* gamma, const void* beta,
                                            const void* pos, void* y, void* y_bf16, void* y_plus_pos, int yp_dtype, void* stats,
                                            int64_t rows, int C, int64_t pos_rows, float eps, float clamp, void* stream) {
  const char* who = "wm2f_add_layernorm_train_fwd";
  WM2F_REQUIRE(x && gamma && beta && y && stats && rows > 0, "%s: null pointer / no rows", who);
  WM2F_REQUIRE(C == 256, "%s: C = %d (built for 256)", who, C);
  WM2F_REQUIRE((x_dtype == WM2F_F32 || x_dtype == WM2F_BF16) && (yp_dtype == WM2F_F32 || yp_dtype == WM2F_BF16), "%s: dtype", who);
  WM2F_REQUIRE(!y_plus_pos || (pos && pos_rows > 0), "%s: y_plus_pos needs pos", who);
  LnFwdArgs a;
  a.x = x; a.res = (const float*)residual; a.gamma = (const float*)gamma; a.beta = (const float*)beta; a.pos = (const float*)pos;
  a.y = (float*)y; a.y_lp = y_bf16; a.yp = y_plus_pos; a.stats = (float*)stats;
  a.rows = rows; a.pos_rows = pos_rows > 0 ? pos_rows : 1; a.eps = eps; a.clamp = clamp; a.x_bf16 = x_dtype == WM2F_BF16; a.yp_bf16 = yp_dtype == WM2F_BF16;
  hipLaunchKernelGGL(add_layernorm_train_fwd_kernel, dim3((unsigned)ceil_div64(rows, 4)), dim3(256), 0, (hipStream_t)stream, a);
  WM2F_CHECK_LAUNCH(who);
  return WM2F_OK;
}

extern "C" int wm2f_add_layernorm_train_bwd(const void* x, int x_dtype, const void* residual, const void* gamma, const void* stats,
                                            const void* grad_y, const void* grad_y_bf16, const void* grad_y_plus_pos, int gyp_dtype,
                                            void* grad_sum, void* grad_x, void* grad_gamma, void* grad_beta, void* workspace,
                                            int64_t rows, int C, void* stream) {
  const char* who = "wm2f_add_layernorm_train_bwd";
  WM2F_REQUIRE(x && gamma && stats && grad_gamma && grad_beta && workspace && rows > 0, "%s: null pointer / no rows", who);
  WM2F_REQUIRE(grad_y || grad_y_bf16 || grad_y_plus_pos, "%s: no incoming gradient", who);
  WM2F_REQUIRE(C == 256, "%s: C = %d (built for 256)", who, C);
  LnBwdArgs a;
  a.x = x; a.res = (const float*)residual; a.gamma = (const float*)gamma; a.stats = (const float*)stats;
  a.gy = (const float*)grad_y; a.gy_lp = grad_y_bf16; a.gyp = grad_y_plus_pos;
  a.dres = (float*)grad_sum; a.dx = grad_x; a.ws = (float*)workspace; a.rows = rows; a.rows_per_block = kLnRowsPerBlock;
  a.x_bf16 = x_dtype == WM2F_BF16; a.gyp_bf16 = gyp_dtype == WM2F_BF16; a.dx_bf16 = a.x_bf16;
  const int blocks = (int)ceil_div64(rows, kLnRowsPerBlock);
  hipLaunchKernelGGL(add_layernorm_train_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  WM2F_CHECK_LAUNCH(who);
  hipLaunchKernelGGL(add_layernorm_train_reduce_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, (const float*)workspace,
                     (float*)grad_gamma, (float*)grad_beta, blocks);
  WM2F_CHECK_LAUNCH(who);
  return WM2F_OK;
}
