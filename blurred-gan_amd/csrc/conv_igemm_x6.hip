// EXPERIMENT, off by default (BGAN_CONV_MATH=bf16x6 turns it on for the layers it covers): the gather-GEMM of
// conv_igemm.hip with every fp32 operand split exactly into three bf16 pieces (8 + 8 + 8 mantissa bits, same exponent range)
// and the product rebuilt from the six largest cross terms on the bf16 matrix pipe, accumulated in fp32:
//     a*b ~= a1*b1 + a1*b2 + a2*b1 + a1*b3 + a3*b1 + a2*b2          (dropped: a2*b3 + a3*b2 + a3*b3 <= 2^-23 |a*b|)
// v_mfma_f32_32x32x16_bf16 issues in 32 cycles for 16 k's, the native fp32 instruction needs 8 x 64 cycles for the same block:
// 6 x 32 against 512 cycles.  Measured numbers and the parity of this path against the fp64 oracle are in DESIGN.md section 9;
// the product path stays exact-fp32 until the contract says otherwise.
//
// Workgroup = 4 waves, 128 x 128 tile (64 x 64 per wave: every fragment feeds two MFMAs), 16 k per step.  Operands are
// loaded as fp32 (same buffer-descriptor gather as the fp32 kernel: zero padding = poisoned offsets), split in registers and
// staged as three bf16 images per operand: [row][16 k] = 32 B per row, the two 16-B halves swapped on every other group of 8
// rows so the per-lane ds_read_b128 of a fragment is bank-conflict free.
#include "conv_common.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace {

using bg::GatherParams;
using bg::GatherPhase;
using bg::RowAnchor;

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Split3 { bf16x8 p[3]; };

__device__ inline void split3(const float (&v)[8], Split3& o) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 h = (__bf16)v[i];
    const float r1 = v[i] - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    o.p[0][i] = h;
    o.p[1][i] = m;
    o.p[2][i] = (__bf16)r2;
  }
}

__global__ __launch_bounds__(256) void conv_igemm_x6_kernel(const GatherParams p) {
  constexpr int BM = 128, BN = 128, BK = 16;
  constexpr int ROWB = 32;                                 // bytes per row of one piece
  constexpr int PIECE = BM * ROWB;                          // 4 KiB
  constexpr int STAGE = 6 * PIECE;                          // A: 3 pieces, B: 3 pieces
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];
  __shared__ int rowdst[BM];
  __shared__ int taplist[bg::kMaxTaps];

  const int phase = blockIdx.z;
  const GatherPhase& g = p.ph[phase];
  const int Mph = p.B * g.Ha * g.Wa;
  const int mt = p.mtiles;
  const int L = p.xcd_swizzle ? bg::xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int n_tile = L / mt, m_tile = L - n_tile * mt;
  const int m0 = m_tile * BM;
  if (m0 >= Mph) return;
  const int n0 = n_tile * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  constexpr unsigned kOob = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Wt), 0, (int)p.w_bytes, 0x00020000);
  // loader: thread = (row, half): 8 consecutive k of one A row and of one B row
  const int lrow = tid >> 1, lh = tid & 1;
  RowAnchor ra;
  int dstrow;
  bg::decode_row(p, g, m0 + lrow, Mph, ra, dstrow);
  const int a_y = ra.ay, a_x = ra.ax;
  const unsigned a_off = (unsigned)(((ra.b * p.Hs + ra.ay) * p.Ws + ra.ax) * p.Ck + lh * 8) * 4u;
  const int nrow = n0 + lrow;
  const unsigned b_off = nrow < p.N ? (unsigned)(nrow * p.Ck + lh * 8) * 4u : kOob;
  if (lh == 0) rowdst[lrow] = dstrow;
  if (tid < g.ntaps) taplist[tid] = g.tap[tid];
  __syncthreads();
  const int kchunks = p.Ck / BK, nsteps = g.ntaps * kchunks;

  float ra8[8], rb8[8];
  auto gload = [&](int step) {
    const bool live = step < nsteps;
    const int tq = min(step / kchunks, g.ntaps - 1), kc = step - (step / kchunks) * kchunks;
    const int tp = taplist[tq];
    const int dy = bg::tap_dy(tp), dx = bg::tap_dx(tp);
    const unsigned tapoff = (unsigned)(((dy * p.Ws + dx) * p.Ck + kc * BK) * 4);
    const unsigned woff = (unsigned)((bg::tap_wi(tp) * p.N * p.Ck + kc * BK) * 4);
    const bool ok = live && (unsigned)(a_y + dy) < (unsigned)p.Hs && (unsigned)(a_x + dx) < (unsigned)p.Ws;
    const unsigned ao = ok ? a_off + tapoff : kOob, bo = (b_off == kOob || !live) ? kOob : b_off + woff;
    const float4 a0 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsA, ao, 0, 0));
    const float4 a1 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsA, ao == kOob ? kOob : ao + 16, 0, 0));
    const float4 b0 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsB, bo, 0, 0));
    const float4 b1 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsB, bo == kOob ? kOob : bo + 16, 0, 0));
    ra8[0] = a0.x; ra8[1] = a0.y; ra8[2] = a0.z; ra8[3] = a0.w; ra8[4] = a1.x; ra8[5] = a1.y; ra8[6] = a1.z; ra8[7] = a1.w;
    rb8[0] = b0.x; rb8[1] = b0.y; rb8[2] = b0.z; rb8[3] = b0.w; rb8[4] = b1.x; rb8[5] = b1.y; rb8[6] = b1.z; rb8[7] = b1.w;
  };
  // physical 16-B chunk of (row, half): halves swap on odd groups of 8 rows
  auto chunk_off = [](int row, int half) { return row * ROWB + ((half ^ ((row >> 3) & 1)) << 4); };
  auto lstore = [&](int buf) {
    Split3 sa, sb;
    split3(ra8, sa);
    split3(rb8, sb);
    unsigned char* base = smem + buf * STAGE;
    const int off = chunk_off(lrow, lh);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      *reinterpret_cast<bf16x8*>(base + q * PIECE + off) = sa.p[q];
      *reinterpret_cast<bf16x8*>(base + (3 + q) * PIECE + off) = sb.p[q];
    }
  };

  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  gload(0);
  lstore(0);
  __syncthreads();
  const int fi = lane & 31, fk = lane >> 5;
  for (int step = 0; step < nsteps; ++step) {
    const int cur = step & 1;
    gload(step + 1);                                        // next tile's fp32 values fly under this tile's MFMAs
    const unsigned char* base = smem + cur * STAGE;
    bf16x8 af[2][3], bf[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = wm * 64 + i * 32 + fi;
#pragma unroll
      for (int q = 0; q < 3; ++q) af[i][q] = *reinterpret_cast<const bf16x8*>(base + q * PIECE + chunk_off(row, fk));
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = wn * 64 + j * 32 + fi;
#pragma unroll
      for (int q = 0; q < 3; ++q) bf[j][q] = *reinterpret_cast<const bf16x8*>(base + (3 + q) * PIECE + chunk_off(row, fk));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        // smallest terms first
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
      }
    lstore(cur ^ 1);
    __syncthreads();
  }

  const int col = lane & 31, rhalf = (lane >> 5) * 4;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 64 + j * 32 + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + rhalf;
        const int dst = rowdst[row];
        if (dst >= 0 && n < p.N) {
          const size_t idx = (size_t)dst * p.N + n;
          p.C[idx] = bg::apply_epilogue(p, acc[i][j][r], idx, n);
        }
      }
    }
}

}  // namespace

namespace bg {

// Runs the split-bf16 kernel when BGAN_CONV_MATH=bf16x6 and the shape is covered (Ck % 16 == 0, N >= 64); *taken = 0 otherwise.
int try_conv_x6(GatherParams& p, void* stream, const char* name, int* taken) {
  *taken = 0;
  static const int on = getenv("BGAN_CONV_MATH") && !strcmp(getenv("BGAN_CONV_MATH"), "bf16x6");
  if (!on || p.Ck % 16 != 0 || p.N < 64) return BG_OK;
  int mmax = 0, maxpos = 0;
  for (int i = 0; i < p.nphase; ++i) {
    mmax = std::max(mmax, p.B * p.ph[i].Ha * p.ph[i].Wa);
    maxpos = std::max(maxpos, p.ph[i].Ha * p.ph[i].Wa);
  }
  // first cut of the experiment: no padding-tap skipping, no split-K, one tile shape -> only the layers where those do not
  // matter (feature maps above 8x8, at least two workgroups per CU); everything else stays on the fp32 kernel
  static const int all = getenv("BGAN_X6_ALL") ? 1 : 0;
  if (!all && (maxpos <= 64 || (long)cdiv(mmax, 128) * cdiv(p.N, 128) * p.nphase < 512)) return BG_OK;
  p.ksplit = 1; p.slab = nullptr; p.pos_major = 0; p.pmerge = 1;
  p.mtiles = (int)cdiv(mmax, 128);
  p.xcd_swizzle = 1;
  double flops = 0;
  for (int i = 0; i < p.nphase; ++i) flops += 2.0 * p.B * p.ph[i].Ha * p.ph[i].Wa * (double)p.N * p.Ck * p.ph[i].ntaps;
  dim3 grid(p.mtiles * cdiv(p.N, 128), 1, p.nphase);
  Launch L(stream, name, flops, 0);
  hipLaunchKernelGGL(conv_igemm_x6_kernel, grid, dim3(256), 0, L.s, p);
  *taken = 1;
  return L.done(name);
}

}  // namespace bg
