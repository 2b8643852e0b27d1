#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float floatx4 __attribute__((ext_vector_type(4)));
__device__ inline void transpose4x4_lanegroups(floatx4& v) {
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  u2 a = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1]), false, false);
  u2 b = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v[2]), __builtin_bit_cast(unsigned, v[3]), false, false);
  u2 c = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false);
  u2 d = __builtin_amdgcn_permlane32_swap(a[1], b[1], false, false);
  v[0] = __builtin_bit_cast(float, c[0]);
  v[1] = __builtin_bit_cast(float, d[0]);
  v[2] = __builtin_bit_cast(float, c[1]);
  v[3] = __builtin_bit_cast(float, d[1]);
}
__global__ void k(const float* in, float* out) {
  const int lane = threadIdx.x;
  floatx4 v;
  for (int i = 0; i < 4; ++i) v[i] = in[lane * 4 + i];
  transpose4x4_lanegroups(v);
  for (int i = 0; i < 4; ++i) out[lane * 4 + i] = v[i];
}
int main() {
  float h[256], o[256];
  for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) h[l * 4 + i] = 1000.f * (l >> 4) + 100.f * i + (l & 15);
  float *di, *dout; hipMalloc(&di, sizeof h); hipMalloc(&dout, sizeof h);
  hipMemcpy(di, h, sizeof h, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
  hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) {
    const float want = 1000.f * i + 100.f * (l >> 4) + (l & 15);
    if (o[l * 4 + i] != want) { if (bad < 6) printf("lane %d reg %d got %.0f want %.0f\n", l, i, o[l * 4 + i], want); ++bad; }
  }
  printf("transpose (data from memory) mismatches: %d\n", bad);
  return 0;
}
