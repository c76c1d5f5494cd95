// Do fp32 MFMAs (v_mfma_f32_16x16x4_f32) and ordinary fp32 vector instructions execute TOGETHER on one SIMD of gfx950?
// K2 (masked cross-attention, fp32) spends 32 cycles per MFMA and ~4 per vector instruction, and its counters say the
// two never overlap (SQ_VALU_MFMA_COEXEC_CYCLES = 0, MFMA busy + vector active = kernel time).  This probe asks the
// hardware directly: 512-thread workgroups = 2 waves per SIMD; waves 0-3 run a chain-free MFMA loop, waves 4-7 a
// chain-free v_fma_f32 (or v_exp_f32) loop of the same nominal length; each half is timed alone and both together.
//   together ~ max(alone)  -> the pipes overlap across waves;   together ~ sum(alone) -> they share the unit.
// mode 3 puts both streams into ONE wave (independent registers): can a wave hide its own vector work under its MFMAs?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_valu_coexec.hip -o /tmp/coexec && /tmp/coexec
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int MODE, int TRANS>
__global__ __launch_bounds__(512) void probe(const float* __restrict__ in, float* __restrict__ out, unsigned long long* stamps, int iters) {
  const int tid = threadIdx.x, wave = tid >> 6;
  const bool do_m = (MODE == 0 || MODE == 2) ? wave < 4 : (MODE == 3);
  const bool do_v = (MODE == 1 || MODE == 2) ? wave >= 4 : (MODE == 3);
  if (MODE == 3 && wave >= 4) return;
  if (!do_m && !do_v) return;
  f32x4 acc[8];
  float a[8], b[8], x[16];
  for (int i = 0; i < 8; ++i) { acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; a[i] = in[(tid * 8 + i) & 4095]; b[i] = in[(2048 + tid * 8 + i) & 4095]; }
  for (int i = 0; i < 16; ++i) x[i] = in[(tid * 16 + i) & 4095];
  const float c = in[7], d = in[9];
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 3) {  // one wave, interleaved: 1 MFMA then 8 vector instructions, all independent
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[i], acc[i], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int r = (8 * i + u) & 15;
          if (TRANS) x[r] = __builtin_amdgcn_exp2f(x[r]); else x[r] = __builtin_fmaf(x[r], c, d);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      if (do_m) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[i], acc[i], 0, 0, 0);
      }
      if (do_v) {
#pragma unroll
        for (int u = 0; u < (TRANS ? 16 : 64); ++u) {  // 64 x 4 cycles, or 16 x 16 cycles = 256 = 8 MFMAs x 32
          const int r = u & 15;
          if (TRANS) x[r] = __builtin_amdgcn_exp2f(x[r]); else x[r] = __builtin_fmaf(x[r], c, d);
        }
      }
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 16; ++i) s += x[i];
  out[blockIdx.x * 512 + tid] = s;
  if ((tid & 63) == 0) stamps[blockIdx.x * 8 + wave] = c1 - c0;
}

template <int MODE, int TRANS>
static void run(const char* name, const float* in, float* out, unsigned long long* st, int iters) {
  const int n_wg = 256;
  hipMemset(st, 0, n_wg * 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, TRANS>), dim3(n_wg), dim3(512), 0, 0, in, out, st, iters);
    hipEventRecord(e1);
  }
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> s(n_wg * 8);
  hipMemcpy(s.data(), st, n_wg * 64, hipMemcpyDeviceToHost);
  double m = 0, v = 0; int nm = 0, nv = 0;
  for (int w = 0; w < n_wg; ++w) for (int k = 0; k < 8; ++k) if (s[w * 8 + k]) { if (k < 4) { m += s[w * 8 + k]; ++nm; } else { v += s[w * 8 + k]; ++nv; } }
  printf("{\"probe\": \"%s\", \"kernel_ms\": %.3f, \"cycles_per_iter_waves0_3\": %.1f, \"cycles_per_iter_waves4_7\": %.1f}\n", name, ms,
         nm ? m / nm / iters : 0.0, nv ? v / nv / iters : 0.0);
}

int main() {
  const int iters = 20000;
  float *in, *out; unsigned long long* st;
  hipMalloc(&in, 4096 * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&st, 256 * 64);
  std::vector<float> h(4096);
  srand(1);
  for (auto& v : h) v = (float)rand() / RAND_MAX * 0.5f;
  hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  run<0, 0>("mfma_alone (8 MFMA / iter, waves 0-3)", in, out, st, iters);
  run<1, 0>("fma_alone (64 v_fma / iter, waves 4-7)", in, out, st, iters);
  run<2, 0>("mfma + fma on the two waves of a SIMD", in, out, st, iters);
  run<1, 1>("exp_alone (16 v_exp / iter, waves 4-7)", in, out, st, iters);
  run<2, 1>("mfma + exp on the two waves of a SIMD", in, out, st, iters);
  run<3, 0>("one wave: 8 x (1 MFMA + 8 v_fma)", in, out, st, iters);
  run<3, 1>("one wave: 8 x (1 MFMA + 8 v_exp)", in, out, st, iters);
  return 0;
}
