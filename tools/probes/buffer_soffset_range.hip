// Is the SCALAR offset of a raw buffer load part of the range check on gfx950?  A descriptor over the first 4096 bytes of an
// allocation whose second half holds a sentinel; loads at (voffset, soffset) pairs whose sum crosses num_records.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/buffer_soffset_range.hip -o /tmp/bsr && /tmp/bsr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const float* base, float* out) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 4096, 0x00020000);
  const int voff[6] = {4000, 4000, 0, 4092, 2048, 0x7ffffff0};
  const int soff[6] = {0, 200, 4096, 4, 2048, 64};
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int s = __builtin_amdgcn_readfirstlane(soff[i]);
      out[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff[i], s, 0));
    }
    // b96 at the edge: bytes 4088..4099 straddle num_records
    typedef unsigned u3 __attribute__((ext_vector_type(3)));
    const u3 v = __builtin_amdgcn_raw_buffer_load_b96(rs, 0, __builtin_amdgcn_readfirstlane(4088), 0);
    out[6] = __builtin_bit_cast(float, v.x); out[7] = __builtin_bit_cast(float, v.y); out[8] = __builtin_bit_cast(float, v.z);
  }
}
int main() {
  float *d, *o;
  hipMalloc(&d, 8192); hipMalloc(&o, 64);
  std::vector<float> h(2048);
  for (int i = 0; i < 1024; ++i) h[i] = 1.0f + i;       // in range: element i = 1 + i
  for (int i = 1024; i < 2048; ++i) h[i] = -7.0f;       // past num_records: sentinel
  hipMemcpy(d, h.data(), 8192, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(d, o);
  float r[9]; hipMemcpy(r, o, 36, hipMemcpyDeviceToHost);
  const char* what[6] = {"voff 4000 + soff 0 (in range, expect 1001)", "voff 4000 + soff 200 (sum past the end)", "voff 0 + soff 4096 (sum = num_records)",
                         "voff 4092 + soff 4 (sum = num_records)", "voff 2048 + soff 2048 (sum = num_records)", "voff 0x7ffffff0 + soff 64"};
  for (int i = 0; i < 6; ++i) printf("%-48s -> %g\n", what[i], r[i]);
  printf("b96 at soff 4088 (last 8 bytes in range, 4 past)   -> %g %g %g\n", r[6], r[7], r[8]);
  printf("sentinel -7 = read past num_records (soffset NOT range-checked); 0 = dropped by the range check\n");
  return 0;
}
