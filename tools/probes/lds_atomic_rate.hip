// Probe: cycles per wave-instruction of LDS atomics (int add with return, float add) vs plain read-modify-write,
// conflict-free addresses, 8 waves per workgroup, 1 workgroup per CU.
// hipcc --offload-arch=gfx950 -O2 -o /tmp/ldsat tools/probes/lds_atomic_rate.hip && /tmp/ldsat
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
  __shared__ float f[16384];
  __shared__ unsigned u[16384];
  for (int i = threadIdx.x; i < 16384; i += 512) { f[i] = 0.f; u[i] = 0; }
  __syncthreads();
  const int lane = threadIdx.x;
  float acc = 0.f; unsigned ua = 0;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int a = (lane + 512 * ((it + j) & 31)) & 16383;   // every lane its own word
      if (MODE == 0) ua += atomicAdd(&u[a], 1u);              // ds_add_rtn_u32
      if (MODE == 1) atomicAdd(&f[a], 1.0f);                  // ds_add_f32
      if (MODE == 2) { f[a] = f[a] + 1.0f; }                  // ds_read_b32 + ds_write_b32
      if (MODE == 3) atomicAdd(&u[a], 1u);                    // ds_add_u32 (no return)
      if (MODE == 4) { const int b = (a & ~3); float4 v = *reinterpret_cast<float4*>(&f[b]); v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f; *reinterpret_cast<float4*>(&f[b]) = v; }  // b128 RMW (lanes collide by 4: timing only)
    }
  }
  __syncthreads();
  long long t1 = clock64();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 512 + threadIdx.x] = acc + f[lane] + (float)(ua + u[lane]);
}
int main() {
  float* out; long long* cyc; long long h[256];
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
  const int iters = 200;
  const char* names[5] = {"ds_add_rtn_u32", "ds_add_f32", "ds_read_b32+ds_write_b32", "ds_add_u32", "b128 read + b128 write"};
  for (int m = 0; m < 5; ++m) {
    for (int rep = 0; rep < 2; ++rep) {
      if (m == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, out, cyc, iters);
      if (m == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, out, cyc, iters);
      if (m == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, out, cyc, iters);
      if (m == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, out, cyc, iters);
      if (m == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 0, 0, out, cyc, iters);
      hipDeviceSynchronize();
    }
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 256; ++i) s += h[i];
    // per CU: 8 waves x iters x 16 wave-instructions (x2 for the RMW pairs)
    printf("%-28s %8.1f cycles per workgroup; %.1f cycles per wave-op (8 waves share the CU)\n", names[m], s / 256,
           s / 256 / (8.0 * iters * 16));
  }
  return 0;
}
