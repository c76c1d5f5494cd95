// Probe: what does `buffer_load_dwordx4 ... lds` write for lanes whose offset fails the range check?
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 -o /tmp/oob tools/probes/lds_dma_oob.hip && /tmp/oob
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void k(const float* src, float* out, int nbytes) {
  __shared__ __attribute__((aligned(16))) float4 w[64];
  w[threadIdx.x] = make_float4(-7.f, -7.f, -7.f, -7.f);  // sentinel
  __syncthreads();
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  const unsigned off = (threadIdx.x & 1) ? 0x80000000u : threadIdx.x * 16u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)w, 16, (int)off, 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  out[threadIdx.x] = w[threadIdx.x].x;
}
int main() {
  float h[256], o[64], *d, *dout;
  for (int i = 0; i < 256; ++i) h[i] = (float)(i + 1);
  hipMalloc(&d, sizeof(h)); hipMalloc(&dout, sizeof(o));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, dout, (int)sizeof(h));
  hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  for (int i = 0; i < 8; ++i) printf("lane %d: %g\n", i, o[i]);
  int zeros = 0, sentinel = 0;
  for (int i = 1; i < 64; i += 2) { zeros += o[i] == 0.f; sentinel += o[i] == -7.f; }
  printf("OOB lanes: %d wrote 0, %d kept the sentinel (of 32)\n", zeros, sentinel);
  return 0;
}
