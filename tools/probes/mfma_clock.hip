// What clock does the chip hold under a dense fp32-MFMA load?  (MI355X_MICROARCH.md, "DVFS give-back", item 6: in-kernel clock =
// delta s_memtime / delta s_memrealtime x 100 MHz, stamped around the loop after seconds of back-to-back launches on random data.)
// Every fp32-MFMA "fraction of the 157.3 TFLOP/s peak" in this repository is priced at the 2.4 GHz datasheet clock; this probe
// measures what the silicon actually runs at, so that DESIGN.md can say how much of a gap is issue efficiency and how much is clock.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_clock.hip -o /tmp/mfma_clock && /tmp/mfma_clock
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

using f32x4 = __attribute__((ext_vector_type(4))) float;

// WAVES_PER_SIMD waves per SIMD, each a chain of independent 16x16x4 fp32 MFMAs on register operands (random data).
template <int LDS_READS>
__global__ __launch_bounds__(512) void mfma_loop(const float* __restrict__ in, float* __restrict__ out, unsigned long long* stamps, int iters) {
  __shared__ float lds[16384];
  const int tid = threadIdx.x;
  for (int i = tid; i < 16384; i += 512) lds[i] = in[i];
  __syncthreads();
  f32x4 acc[8];
  float a[16], b[16];  // 16 different random A and B values per lane: consecutive MFMAs see different operand bits, as a GEMM's do
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 16; ++i) { a[i] = in[(tid * 16 + i) & 16383]; b[i] = in[(4096 + tid * 16 + i) & 16383]; }
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (LDS_READS == 1) {  // one 16-byte LDS read per 8 MFMAs and wave, as an A-fragment stream would cost
      const f32x4 v = *reinterpret_cast<const f32x4*>(lds + ((it * 64 + (tid & 63)) & 4095) * 4);
      a[0] = v[0]; a[1] = v[1]; a[2] = v[2]; a[3] = v[3];
    }
    if (LDS_READS == 2) {  // the token GEMM's order: groups of 8 MFMAs on TWO accumulators that alternate (dependent distance 2)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int u = 0; u < 2; ++u)
            acc[2 * g + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(8 * u + t + 4 * (g & 1)) & 15], b[t + 4 * (g & 1)], acc[2 * g + u], 0, 0, 0);
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(4 * i + t) & 15], b[(t * 5 + i * 3) & 15], acc[i], 0, 0, 0);
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 512 + tid] = s;
  if (tid == 0) { stamps[blockIdx.x * 2] = c1 - c0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}

int main() {
  const int n_wg = 256, iters = 40000;  // 32 MFMAs per iteration and wave, 2 waves per SIMD
  float *in, *out;
  unsigned long long* st;
  hipMalloc(&in, 16384 * 4);
  hipMalloc(&out, n_wg * 512 * 4);
  hipMalloc(&st, n_wg * 16);
  std::vector<float> h(16384);
  srand(1);
  for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  hipMemcpy(in, h.data(), 16384 * 4, hipMemcpyHostToDevice);
  for (int variant = 0; variant < 3; ++variant) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 60; ++rep) {  // ~2.5 s of back-to-back launches before the last one is read
      hipEventRecord(e0);
      if (variant == 0) hipLaunchKernelGGL(mfma_loop<0>, dim3(n_wg), dim3(512), 0, 0, in, out, st, iters);
      else if (variant == 1) hipLaunchKernelGGL(mfma_loop<1>, dim3(n_wg), dim3(512), 0, 0, in, out, st, iters);
      else hipLaunchKernelGGL(mfma_loop<2>, dim3(n_wg), dim3(512), 0, 0, in, out, st, iters);
      hipEventRecord(e1);
    }
    hipDeviceSynchronize();
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> s(n_wg * 2);
    hipMemcpy(s.data(), st, n_wg * 16, hipMemcpyDeviceToHost);
    std::vector<double> ghz;
    for (int i = 0; i < n_wg; ++i) ghz.push_back((double)s[2 * i] / (double)s[2 * i + 1] * 0.1);
    std::sort(ghz.begin(), ghz.end());
    const double flop = 2.0 * 16 * 16 * 4 * 32.0 * iters * 8 * n_wg;
    printf("{\"probe\": \"fp32 mfma 16x16x4, 2 waves per SIMD, %s\", \"clock_ghz_median\": %.3f, \"clock_ghz_min\": %.3f, \"clock_ghz_max\": %.3f, "
           "\"ms\": %.3f, \"TFLOPs\": %.1f, \"cycles_per_mfma_per_simd\": %.2f}\n",
           variant == 0 ? "register operands only, 8 independent accumulators" : variant == 1 ? "one ds_read_b128 per 8 MFMAs" : "two alternating accumulators per group of 8 (the token GEMM's order)", ghz[n_wg / 2], ghz.front(), ghz.back(), ms, flop / ms / 1e9,
           ghz[n_wg / 2] * 1e9 * ms * 1e-3 / (32.0 * iters * 2));
  }
  return 0;
}
