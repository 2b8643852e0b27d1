// What does an instruction of another kind cost the fp32 matrix pipe?  A register-only MFMA loop (four independent
// accumulators, 16 x v_mfma_f32_32x32x2_f32 per iteration) with K extra instructions of one kind per iteration, at 3 waves per
// SIMD (192-thread workgroups, 1024 of them: the band blur's shape) and at 4.  If the extra instructions issue in the shadow of
// the MFMAs the rate stays at the peak; if they take issue slots from the pipe it drops by (their cost) / (16 * 64 cycles).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_coissue.hip -o /tmp/mfma_coissue && /tmp/mfma_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
enum Kind { NONE, DS_READ, DS_READ2, DS_WRITE, VALU, SALU, VMEM_LOAD, VMEM_STORE, ACC_READ };

template <int KIND, int K>
__global__ void loop(float* out, const float* in, int iters, int stride) {
  __shared__ float lds[4096];
  floatx16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const int lane = threadIdx.x;
  lds[lane] = lane; lds[lane + 1024] = 1.f;
  __syncthreads();
  float a[4], b[4];
  for (int u = 0; u < 4; ++u) { a[u] = 1.f + lane * 1e-3f + u; b[u] = 2.f - lane * 1e-3f + u; }
  float sink = 0.f;
  int sidx = lane;
  float* op = out + (size_t)blockIdx.x * 4096 + lane * 4;
  const float* ip = in + (size_t)blockIdx.x * 4096 + lane * 4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + i) & 3], acc[i], 0, 0, 0);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (KIND == DS_READ) { float v; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(lane * 4), "n"(k * 256)); sink += v; }
      if (KIND == DS_READ2) { float2 v; asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(lane * 4), "n"(k * 2), "n"(k * 2 + 1)); sink += v.x; }
      if (KIND == DS_WRITE) { asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(lane * 4), "v"(sink), "n"(8192 + k * 256)); }
      if (KIND == VALU) { asm volatile("v_add_u32 %0, %0, 1" : "+v"(sidx)); }
      if (KIND == SALU) { asm volatile("s_add_u32 s40, s40, 1" ::: "s40"); }
      if (KIND == VMEM_LOAD) { float4 v = *reinterpret_cast<const float4*>(ip + ((it * K + k) & 3) * 1024 * 0 + stride * ((it + k) & 7)); sink += v.x; }
      if (KIND == VMEM_STORE) { *reinterpret_cast<float4*>(op + stride * ((it + k) & 7)) = make_float4(sink, 1.f, 2.f, 3.f); }
      if (KIND == ACC_READ) { float v; asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(v)); sink += v; }
    }
    if (KIND == DS_READ || KIND == DS_READ2) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  float s = sink + sidx;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) out[0] = s;
}

template <int KIND, int K>
static void run(const char* name, int threads, int wgs, float* out, const float* in) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((loop<KIND, K>), dim3(wgs), dim3(threads), 0, 0, out, in, iters / 4, 256);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((loop<KIND, K>), dim3(wgs), dim3(threads), 0, 0, out, in, iters, 256);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double fl = (double)wgs * (threads / 64) * iters * 16 * 2.0 * 32 * 32 * 2;
  printf("%-12s K=%2d  %3d threads x %4d wgs  %.3f ms  %.1f TFLOP/s\n", name, K, threads, wgs, best, fl / best / 1e9);
}

int main() {
  float *out, *in;
  hipMalloc(&out, (size_t)2048 * 4096 * 4 + 65536);
  hipMalloc(&in, (size_t)2048 * 4096 * 4 + 65536);
  hipMemset(in, 0, (size_t)2048 * 4096 * 4 + 65536);
  for (int threads : {192, 256}) {
    const int wgs = 1024;
    run<NONE, 0>("none", threads, wgs, out, in);
    run<DS_READ, 8>("ds_read", threads, wgs, out, in);
    run<DS_READ, 20>("ds_read", threads, wgs, out, in);
    run<DS_READ2, 10>("ds_read2", threads, wgs, out, in);
    run<DS_WRITE, 8>("ds_write", threads, wgs, out, in);
    run<VALU, 16>("valu", threads, wgs, out, in);
    run<VALU, 64>("valu", threads, wgs, out, in);
    run<SALU, 64>("salu", threads, wgs, out, in);
    run<ACC_READ, 16>("accvgpr_read", threads, wgs, out, in);
    run<VMEM_LOAD, 1>("vmem_load", threads, wgs, out, in);
    run<VMEM_LOAD, 4>("vmem_load", threads, wgs, out, in);
    run<VMEM_STORE, 1>("vmem_store", threads, wgs, out, in);
    run<VMEM_STORE, 4>("vmem_store", threads, wgs, out, in);
  }
  return 0;
}
