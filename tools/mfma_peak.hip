// Sustained fp32-MFMA issue rate on this card: a register-only loop of v_mfma_f32_32x32x2_f32 (no memory traffic).
// Gives the practical ceiling the conv kernels are priced against in DESIGN.md next to the 157.3 TFLOP/s datasheet peak.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma32_loop(float* out, int iters, float a0, float b0) {
  floatx16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) out[0] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void mfma16_loop(float* out, int iters, float a0, float b0) {
  floatx4 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 4; ++r) s += acc[i][r];
  if (s == 12345.678f) out[0] = s;
}

// Random operands (8 A and 8 B registers per lane, hashed from the lane id): the chip holds a lower clock on toggling data
// than on constants, so THIS is the ceiling real kernels are measured against.
__device__ inline float hash01(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return (float)(x & 0xffffff) * (2.0f / 16777216.0f) - 1.0f;
}

template <int NACC>
__global__ __launch_bounds__(256) void mfma32_rand_loop(float* out, int iters) {
  floatx16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a[8], b[8];
  const unsigned id = blockIdx.x * 256 + threadIdx.x;
  for (int i = 0; i < 8; ++i) { a[i] = hash01(id * 16 + i); b[i] = hash01(id * 16 + 8 + i); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + i) & 7], acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) out[0] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void mfma16_rand_loop(float* out, int iters) {
  floatx4 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  float a[8], b[8];
  const unsigned id = blockIdx.x * 256 + threadIdx.x;
  for (int i = 0; i < 8; ++i) { a[i] = hash01(id * 16 + i); b[i] = hash01(id * 16 + 8 + i); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[(u + i) & 7], acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 4; ++r) s += acc[i][r];
  if (s == 12345.678f) out[0] = s;
}

// bf16 pipe, random operands: the rate an fp32 product emulated with 3 x bf16 pieces (6 cross terms) would be priced against
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC>
__global__ __launch_bounds__(256) void mfma_bf16_rand_loop(float* out, int iters) {
  floatx16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a[4], b[4];
  const unsigned id = blockIdx.x * 256 + threadIdx.x;
  for (int i = 0; i < 4; ++i)
    for (int q = 0; q < 8; ++q) { a[i][q] = (__bf16)hash01(id * 64 + i * 8 + q); b[i][q] = (__bf16)hash01(id * 64 + 32 + i * 8 + q); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u], b[(u + i) & 3], acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) out[0] = s;
}

template <typename F>
static void run(const char* name, F launch, double flop_per_wg_iter, int wgs, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(wgs, iters / 8);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    launch(wgs, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  // a longer sustained run (thermal / power steady state)
  hipEventRecord(e0);
  for (int rep = 0; rep < 20; ++rep) launch(wgs, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms20; hipEventElapsedTime(&ms20, e0, e1);
  const double fl = flop_per_wg_iter * wgs * iters;
  printf("%-28s wgs=%5d  best %.3f ms = %.1f TFLOP/s   sustained(20x) %.1f TFLOP/s\n", name, wgs, best, fl / best / 1e9, fl * 20 / ms20 / 1e9);
}

int main() {
  float* out; hipMalloc(&out, 4);
  const int iters = 20000;
  for (int wgs : {256, 512, 1024, 2048}) {
    run("32x32x2 f32, 4 acc/wave", [&](int g, int it) { hipLaunchKernelGGL(mfma32_loop<4>, dim3(g), dim3(256), 0, 0, out, it, 1.f, 2.f); },
        4.0 /*waves*/ * 4 /*unroll*/ * 4 /*acc*/ * 2.0 * 32 * 32 * 2, wgs, iters);
  }
  run("32x32x2 f32, 1 acc/wave", [&](int g, int it) { hipLaunchKernelGGL(mfma32_loop<1>, dim3(g), dim3(256), 0, 0, out, it, 1.f, 2.f); },
      4.0 * 4 * 1 * 2.0 * 32 * 32 * 2, 1024, iters);
  run("16x16x4 f32, 4 acc/wave", [&](int g, int it) { hipLaunchKernelGGL(mfma16_loop<4>, dim3(g), dim3(256), 0, 0, out, it, 1.f, 2.f); },
      4.0 * 4 * 4 * 2.0 * 16 * 16 * 4, 1024, iters);
  run("32x32x2 f32 RANDOM, 1 acc", [&](int g, int it) { hipLaunchKernelGGL(mfma32_rand_loop<1>, dim3(g), dim3(256), 0, 0, out, it); },
      4.0 * 8 * 1 * 2.0 * 32 * 32 * 2, 1024, iters / 2);
  run("32x32x2 f32 RANDOM, 4 acc", [&](int g, int it) { hipLaunchKernelGGL(mfma32_rand_loop<4>, dim3(g), dim3(256), 0, 0, out, it); },
      4.0 * 8 * 4 * 2.0 * 32 * 32 * 2, 1024, iters / 8);
  run("16x16x4 f32 RANDOM, 1 acc", [&](int g, int it) { hipLaunchKernelGGL(mfma16_rand_loop<1>, dim3(g), dim3(256), 0, 0, out, it); },
      4.0 * 8 * 1 * 2.0 * 16 * 16 * 4, 1024, iters);
  run("16x16x4 f32 RANDOM, 4 acc", [&](int g, int it) { hipLaunchKernelGGL(mfma16_rand_loop<4>, dim3(g), dim3(256), 0, 0, out, it); },
      4.0 * 8 * 4 * 2.0 * 16 * 16 * 4, 1024, iters / 4);
  run("16x16x4 f32 RANDOM, 8 acc", [&](int g, int it) { hipLaunchKernelGGL(mfma16_rand_loop<8>, dim3(g), dim3(256), 0, 0, out, it); },
      4.0 * 8 * 8 * 2.0 * 16 * 16 * 4, 1024, iters / 8);
  for (int wgs : {256, 512, 1024})
    run("16x16x4 f32 RANDOM, 3 acc", [&](int g, int it) { hipLaunchKernelGGL(mfma16_rand_loop<3>, dim3(g), dim3(256), 0, 0, out, it); },
        4.0 * 8 * 3 * 2.0 * 16 * 16 * 4, wgs, iters / 3);
  run("32x32x16 bf16 RANDOM, 4 acc", [&](int g, int it) { hipLaunchKernelGGL(mfma_bf16_rand_loop<4>, dim3(g), dim3(256), 0, 0, out, it); },
      4.0 * 4 * 4 * 2.0 * 32 * 32 * 16, 1024, iters);
  run("32x32x16 bf16 RANDOM, 1 acc", [&](int g, int it) { hipLaunchKernelGGL(mfma_bf16_rand_loop<1>, dim3(g), dim3(256), 0, 0, out, it); },
      4.0 * 4 * 1 * 2.0 * 32 * 32 * 16, 1024, iters * 4);
  return 0;
}
