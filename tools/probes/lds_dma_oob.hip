// What a buffer_load ... lds (LDS-DMA) lane does when its offset is out of the descriptor's range: does the LDS slot get zeros
// (like a register destination) or keep its contents?  The gather-GEMM's DMA variant relies on ZEROS (TF's SAME padding and the
// tails of M and N are poisoned offsets).   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_dma_oob.hip -o /tmp/lds_dma_oob
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* a, float* out, int n) {
  __shared__ __attribute__((aligned(16))) float s[1024];
  for (int i = threadIdx.x; i < 1024; i += 256) s[i] = 777.f;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a), 0, n * 4, 0x00020000);
  unsigned off = threadIdx.x * 16u;
  if (threadIdx.x & 1) off = 0x80000000u;                  // every second lane: poisoned
  if (threadIdx.x < 64)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)s, 16, off, 0, 0, 0);
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 256) out[i] = s[i];
}
int main() {
  float *a, *o;
  (void)hipMalloc(&a, 1 << 20); (void)hipMalloc(&o, 4096);
  float h[1024];
  for (int i = 0; i < 1024; ++i) h[i] = i + 1;
  (void)hipMemcpy(a, h, sizeof h, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, a, o, 1024);
  float r[1024];
  (void)hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
  printf("lane 0 (in range): %g %g %g %g   lane 1 (poisoned): %g %g %g %g   lane 2: %g %g   beyond the wave (untouched): %g\n", r[0], r[1], r[2], r[3], r[4],
         r[5], r[6], r[7], r[8], r[9], r[300]);
  int zeros = 0, kept = 0;
  for (int l = 1; l < 64; l += 2) { zeros += r[4 * l] == 0.f; kept += r[4 * l] == 777.f; }
  printf("poisoned lanes: %d wrote zeros, %d left the slot untouched\n", zeros, kept);
  return 0;
}
