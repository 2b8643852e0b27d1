// Can the fp32 VECTOR pipe add work beside the fp32 MATRIX pipe?  Both are rated 64 FLOP/clk/SIMD on gfx950 (157.3 TFLOP/s each:
// v_pk_fma_f32 = 2 FMAs per lane per 4-cycle issue; v_mfma_f32_32x32x2_f32 = 4096 flop per 64 cycles).  A register-only loop issues
// F independent v_pk_fma_f32 after every MFMA of a dependent chain (four waves per SIMD, random operands) and reports the combined
// rate and the shader clock (s_memtime / s_memrealtime).  If the sum goes well past 157 TFLOP/s, a conv kernel that gives part of
// its tile to the vector pipe has headroom; if the clock falls or the MFMA rate drops by what the fillers add, it has none.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_valu_dual.hip -o /tmp/mfma_valu_dual && /tmp/mfma_valu_dual
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef float floatx4v __attribute__((ext_vector_type(4)));

__device__ inline float hash01(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return (float)(x & 0xffffff) * (2.0f / 16777216.0f) - 1.0f;
}

// MODE 0: v_pk_fma_f32 fillers; 1: v_add_u32 (integer ALU); 2: v_cndmask_b32 + v_cmp pairs (what the gather-GEMM's loader issued);
// 3: ds_read_b128 (conflict-free, results never waited for inside the loop); 4: ds_write_b128; 5: buffer_load_dwordx4 from a small L2-hot buffer
template <int F, int MODE = 0>
__global__ __launch_bounds__(256) void dual_loop(float* out, unsigned long long* clk, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[256 * 36 + 64];
  for (int i = threadIdx.x; i < 256 * 36 + 64; i += 256) lds[i] = (float)i;
  __syncthreads();
  const unsigned lds_addr = (unsigned)(size_t)(lds + (threadIdx.x & 31) * 36 + (threadIdx.x >> 5) * 4) & 0xffffu;   // the gather-GEMM's fragment pattern
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out + 1024, 0, 1 << 20, 0x00020000);
  floatx4v sink = {0.f, 0.f, 0.f, 0.f};
  floatx16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  floatx2 va[16], vb[4], vc[4];
  const unsigned id = blockIdx.x * 256 + threadIdx.x;
  float a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = hash01(id * 64 + i); b[i] = hash01(id * 64 + 8 + i); }
  for (int i = 0; i < 16; ++i) va[i] = floatx2{0.f, 0.f};
  for (int i = 0; i < 4; ++i) { vb[i] = floatx2{hash01(id * 64 + 16 + i), hash01(id * 64 + 24 + i)}; vc[i] = floatx2{hash01(id * 64 + 32 + i) * 1e-3f, hash01(id * 64 + 40 + i) * 1e-3f}; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
#pragma unroll
      for (int f = 0; f < F; ++f) {
        if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(va[(u * F + f) & 15]) : "v"(vb[f & 3]), "v"(vc[(f + u) & 3]));
        else if (MODE == 1) asm volatile("v_add_u32 %0, %1, %0" : "+v"(va[(u * F + f) & 15].x) : "v"(vb[f & 3].x));
        else if (MODE == 2) asm volatile("v_cmp_gt_u32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(va[(u * F + f) & 15].x) : "v"(vb[f & 3].x) : "vcc");
        else if (MODE == 3) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(sink) : "v"(lds_addr), "n"(((0) & 7) * 32) : "memory");
        else if (MODE == 4) asm volatile("ds_write_b128 %0, %1 offset:4608" ::"v"(lds_addr), "v"(sink) : "memory");
        else asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(sink) : "v"((threadIdx.x & 63) * 16u), "s"(rs) : "memory");
      }
      if (MODE >= 3 && u == 3) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      if (false) {
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc[r];
  for (int i = 0; i < 16; ++i) s += va[i].x + va[i].y;
  s += sink[0] + sink[1] + sink[2] + sink[3];
  if (s == 12345.678f) out[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int F, int MODE = 0>
static void run(float* out, unsigned long long* clk, int wgs, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((dual_loop<F, MODE>), dim3(wgs), dim3(256), 0, 0, out, clk, iters / 8);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL((dual_loop<F, MODE>), dim3(wgs), dim3(256), 0, 0, out, clk, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= 10;
  unsigned long long h[2];
  (void)hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
  const double waves = 4.0 * wgs, mf = waves * iters * 4 * 2.0 * 32 * 32 * 2, vf = waves * iters * 4 * F * 64 * 2 * 2.0;
  printf("mode %d (%s)  F=%2d fillers per MFMA: %.3f ms  matrix %.1f + vector %.1f = %.1f TFLOP/s   clock %.3f GHz   cycles per MFMA slot %.1f\n", MODE, MODE == 0 ? "v_pk_fma_f32" : (MODE == 1 ? "v_add_u32" : (MODE == 2 ? "v_cmp + v_cndmask (x2 instructions)" : (MODE == 3 ? "ds_read_b128" : (MODE == 4 ? "ds_write_b128" : "buffer_load_dwordx4")))), F, ms, mf / ms / 1e9,
         MODE == 0 ? vf / ms / 1e9 : 0.0, (mf + (MODE == 0 ? vf : 0.0)) / ms / 1e9, (double)h[0] / (double)h[1] * 0.1, (double)h[0] / (iters * 4.0) / 4.0);
}

int main() {
  float* out; unsigned long long* clk;
  (void)hipMalloc(&out, 8 << 20); (void)hipMalloc(&clk, 16);
  const int wgs = 1024, iters = 20000;
  run<0>(out, clk, wgs, iters);
  run<2>(out, clk, wgs, iters);
  run<4>(out, clk, wgs, iters);
  run<6>(out, clk, wgs, iters);
  run<8>(out, clk, wgs, iters);
  run<10>(out, clk, wgs, iters);
  run<12>(out, clk, wgs, iters);
  run<16>(out, clk, wgs, iters);
  run<2, 1>(out, clk, wgs, iters);
  run<4, 1>(out, clk, wgs, iters);
  run<8, 1>(out, clk, wgs, iters);
  run<1, 2>(out, clk, wgs, iters);
  run<2, 2>(out, clk, wgs, iters);
  run<4, 2>(out, clk, wgs, iters);
  run<1, 3>(out, clk, wgs, iters);
  run<2, 3>(out, clk, wgs, iters);
  run<1, 4>(out, clk, wgs, iters);
  run<1, 5>(out, clk, wgs, iters);
  return 0;
}
