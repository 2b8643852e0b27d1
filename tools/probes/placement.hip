// Where the hardware puts the workgroups of a grid: (XCC, SE, SH, CU) of every workgroup and the order in which slots are
// refilled, for the grid shapes of the gather-GEMM (256 threads, ~37 KB LDS -> 4 workgroups per CU).
//   hipcc --offload-arch=gfx950 -O2 tools/probes/placement.hip -o tools/_build/placement
//   tools/_build/placement <gx> <gz> <lds_bytes> <spin_us_base> [spin pattern: "const" | "z" (spin grows with blockIdx.z)]
// Prints one line per workgroup: linear id, x, z, xcc, se, sh, cu, start tick (relative), end tick.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

struct Rec { unsigned long long t0, t1; unsigned hw, xcc; };

__global__ __launch_bounds__(256) void probe(Rec* out, int spin_ticks, int zmode) {
  extern __shared__ float lds[];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  lds[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  const long ticks = zmode ? (long)spin_ticks * (1 + (int)blockIdx.z) : spin_ticks;
  while ((long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) { __builtin_amdgcn_s_sleep(8); }
  if (threadIdx.x == 0) {
    Rec r;
    r.t0 = t0; r.t1 = __builtin_amdgcn_s_memrealtime();
    r.hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
    r.xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));
    out[(size_t)blockIdx.z * gridDim.x + blockIdx.x] = r;
  }
  if (lds[(threadIdx.x + 1) & 255] < 0.f) out[0].t0 = 0;
}

int main(int argc, char** argv) {
  const int gx = argc > 1 ? atoi(argv[1]) : 256, gz = argc > 2 ? atoi(argv[2]) : 4;
  const int lds = argc > 3 ? atoi(argv[3]) : 37 * 1024, us = argc > 4 ? atoi(argv[4]) : 50;
  const int zmode = argc > 5 && !strcmp(argv[5], "z");
  const size_t n = (size_t)gx * gz;
  Rec* d;
  hipMalloc(&d, n * sizeof(Rec));
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  for (int rep = 0; rep < 2; ++rep) {          // memrealtime ticks at 100 MHz: us * 100
    hipLaunchKernelGGL(probe, dim3(gx, 1, gz), dim3(256), lds, 0, d, us * 100, zmode);
    hipDeviceSynchronize();
  }
  std::vector<Rec> h(n);
  hipMemcpy(h.data(), d, n * sizeof(Rec), hipMemcpyDeviceToHost);
  unsigned long long tmin = ~0ull;
  for (auto& r : h) tmin = r.t0 < tmin ? r.t0 : tmin;
  printf("# gx %d gz %d lds %d spin_us %d zmode %d\n# id x z xcc se sh cu t0 t1\n", gx, gz, lds, us, zmode);
  for (size_t i = 0; i < n; ++i) {
    const Rec& r = h[i];
    printf("%zu %zu %zu %u %u %u %u %llu %llu\n", i, i % gx, i / gx, r.xcc & 15, (r.hw >> 13) & 7, (r.hw >> 12) & 1, (r.hw >> 8) & 15,
           r.t0 - tmin, r.t1 - tmin);
  }
  return 0;
}
