// Conv2D filter gradient, dw[kh,kw,ci,co] = sum_{b,oh,ow} x[b, oh*s+kh-pt, ow*s+kw-pl, ci] * dy[b,oh,ow,co]
// -- the tape gradient w.r.t. kernels of wgan.py:140,166 and the per-layer wgrad of the gradient-penalty
// second order (SURVEY.md 8a "GP second-order derivation", step 2).
//
//   conv_wgrad_kernel   per tap a TN GEMM on fp32 MFMA: C[ci,co] = X_tap[pix,ci]^T * DY[pix,co], the pixel
//                       axis is the contraction: split over workgroups (grid.z, partial slabs in the caller's
//                       workspace, summed by wgrad_reduce_kernel -> deterministic) and optionally over the
//                       4 waves of a workgroup (small channel tiles), reduced through LDS.
//   wgrad_direct_kernel thin / odd channel counts: one thread per (tap, ci, co) output and pixel slice.
#include "conv_common.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <utility>

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct WgradParams {
  const float* X;    // [B][H][W][Ci]
  const float* DY;   // [B][Ho][Wo][Co]
  float* out;        // slabs [ksplit][k*k][Ci][Co] or dw itself when ksplit == 1
  int B, H, W, Ci, Ho, Wo, Co;
  int k, s, pt, pl;
  int M;             // B*Ho*Wo
  int chunk;         // pixels per grid.z slice (multiple of BKP)
  int ksplit;
  int tiles_m, tiles_n, xcd_swizzle;
  float beta, scale; // applied only when ksplit == 1
  unsigned x_bytes, dy_bytes;
  unsigned mul_hw, sh_hw, mul_w, sh_w;   // magic-number division by Ho*Wo and Wo (dividends < 2^31)
  int lg_w, lg_h, lg_b, pow2;            // log2(Wo), log2(Ho), log2(B); pow2 = 1: Ho, Wo powers of two; 2: + B, small map (position-major)
  // position-major launches of the per-tap kernel: taps sorted by their live output positions (a corner tap of a 4x4 map sees 4
  // of the 16 positions, the centre tap all 16), heaviest first, and dealt to the CUs in a snake -- see GatherParams::pp_order
  int order_n;                           // 0 = plain (tile, tap, split) order; else k*k
  unsigned char tap_order[32];
};

// floor(m / d) for m < 2^31 with host-computed (mul, sh): q = (m * mul) >> sh
__device__ inline int fastdiv(int m, unsigned mul, unsigned sh) { return (int)(((unsigned long long)(unsigned)m * mul) >> sh); }

// Main filter-gradient kernel (grid.y = tap): same pipeline as the forward kernel -- buffer loads with the
// hardware range check doing the zero padding / tails, branch-free single-block loop body, one barrier per step:
//   step s: MFMA(first half of chunk s) | ds_write chunk s+1 | issue loads of chunk s+2 | MFMA(second half) | barrier
// P2: 0 = any geometry (incremental per-thread row state); 1 = Ho, Wo powers of two (scalar chunk origin + per-thread
// constants); 2 = additionally B a power of two and a small feature map: pixels are visited position-major (all images at
// one output position, then the next position), so a chunk is zero padding for the tap either entirely or not at all and
// the all-padding chunks are skipped (no loads, no MFMAs) -- about half of the chunks on 4x4 maps.
template <int BM, int BN, int BKP, int WAVES_M, int WAVES_N, int WAVES_K, int P2>
__global__ __launch_bounds__(256) void conv_wgrad_v3_kernel(const WgradParams p) {
  static_assert(WAVES_M * WAVES_N * WAVES_K == 4, "4 waves");
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int MI = WTM / 32, NI = WTN / 32;
  static_assert(MI >= 1 && NI >= 1 && WTM % 32 == 0 && WTN % 32 == 0, "wave tile");
  constexpr int TPR_A = BM / 4, RPP_A = 256 / TPR_A, AP = BKP / RPP_A;
  constexpr int TPR_B = BN / 4, RPP_B = 256 / TPR_B, BP = BKP / RPP_B;
  static_assert(AP >= 1 && BP >= 1 && BKP % RPP_A == 0 && BKP % RPP_B == 0, "loader passes must tile the chunk exactly");
  constexpr int STAGE = BKP * (BM + BN);
  constexpr int RED = (WAVES_K > 1) ? (WAVES_K - 1) * WAVES_M * WAVES_N * MI * NI * 1024 : 0;
  constexpr int SMEM = (2 * STAGE > RED) ? 2 * STAGE : RED;
  constexpr unsigned kOob = 0x80000000u;
  __shared__ __attribute__((aligned(16))) float smem[SMEM];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave % WAVES_K, wmn = wave / WAVES_K;
  const int wm = wmn / WAVES_N, wn = wmn % WAVES_N;
  // 1-D grid, logical order (tile, tap, split) with the pixel split slowest; the XCD remap gives every XCD whole
  // splits, so the workgroups that re-read the same x / dy chunk (all taps and channel tiles of a split) share an L2
  const int ntile = p.tiles_m * p.tiles_n, ntap = p.k * p.k;
  int tile, tap, zsplit;
  if (P2 == 2 && p.order_n > 0) {
    // workgroup w sits on CU slot w % 256 in residency round w / 256 (tools/probes/placement.hip): cost-sorted snake
    const int w = blockIdx.x, r = w >> 8;
    // odd rounds walk the CUs backwards but keep the XCD (w % 8), as in conv_igemm_kernel: an XCD sees the same channel tiles in every round
    const int u = ((r & 1) && ((r + 1) << 8) <= (int)gridDim.x) ? (r << 8) + ((31 - ((w & 255) >> 3)) << 3) + (w & 7) : w;
    tile = u % ntile;
    const int rest = u / ntile;
    zsplit = rest % p.ksplit;
    tap = p.tap_order[rest / p.ksplit];
  } else {
    const int L = p.xcd_swizzle ? bg::xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    tile = L % ntile; tap = (L / ntile) % ntap; zsplit = L / (ntile * ntap);
  }
  const int tile_m = tile / p.tiles_n, tile_n = tile % p.tiles_n;
  const int ci0 = tile_m * BM, co0 = tile_n * BN;
  const int kh = tap / p.k, kw = tap % p.k;
  const int m_begin = zsplit * p.chunk;
  const int m_end = min(p.M, m_begin + p.chunk);
  const int nsteps = (m_end - m_begin + BKP - 1) / BKP;

  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.DY), 0, (int)p.dy_bytes, 0x00020000);
  const int arow = tid / TPR_A, aq = tid % TPR_A;
  const int brow = tid / TPR_B, bq = tid % TPR_B;
  const int HoWo = p.Ho * p.Wo;
  const bool a_col_ok = ci0 + aq * 4 < p.Ci, b_col_ok = co0 + bq * 4 < p.Co;
  const int a_coloff = (ci0 + aq * 4) * 4, b_coloff = (co0 + bq * 4) * 4;
  const int dyk = kh - p.pt, dxk = kw - p.pl;
  float4 regA[AP], regB[BP];

  // Row state advanced incrementally from chunk to chunk (adds / compares / selects only: integer multiplies and
  // divisions are quarter-rate VALU ops that would compete with the MFMA issue slots).  Row of pass i is output
  // pixel m = (b, oh, ow); we keep oh*s, ow*s and the byte offset of x[b, oh*s + dyk, ow*s + dxk, ci0 + 4*aq].
  const int WoS = p.Wo * p.s, HoS = p.Ho * p.s;
  int dstep_ow, dstep_oh, dstep_b;
  {
    dstep_b = BKP / HoWo;
    const int r = BKP - dstep_b * HoWo;
    dstep_oh = r / p.Wo;
    dstep_ow = r - dstep_oh * p.Wo;
  }
  const int d_owS = dstep_ow * p.s, d_ohS = dstep_oh * p.s;
  const int rowB = p.W * p.Ci * 4;                                     // bytes per x row
  const int d_off = (dstep_b * p.H + d_ohS) * rowB + d_owS * p.Ci * 4;
  const int corr_w = p.s * rowB - WoS * p.Ci * 4;                       // ow wrapped: next output row
  const int corr_h = (p.H - HoS) * rowB;                                // oh wrapped: next image
  int a_m[AP], a_owS[AP], a_ohS[AP];
  unsigned a_off[AP], b_off[BP];
  int b_m[BP];
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    const int m = m_begin + arow + i * RPP_A;
    const int b = fastdiv(m, p.mul_hw, p.sh_hw);
    const int rem = m - b * HoWo;
    const int oh = fastdiv(rem, p.mul_w, p.sh_w);
    const int ow = rem - oh * p.Wo;
    a_m[i] = m;
    a_ohS[i] = oh * p.s;
    a_owS[i] = ow * p.s;
    a_off[i] = (unsigned)(((b * p.H + oh * p.s + dyk) * p.W + ow * p.s + dxk) * p.Ci * 4 + a_coloff);
  }
#pragma unroll
  for (int i = 0; i < BP; ++i) {
    b_m[i] = m_begin + brow + i * RPP_B;
    b_off[i] = (unsigned)(b_m[i] * p.Co * 4 + b_coloff);
  }
  const int b_dstep = BKP * p.Co * 4;

  // P2 (Ho, Wo powers of two -- every layer of the 64/128-pixel models): a chunk of BKP pixels starting at a multiple of
  // BKP decomposes as (chunk origin: wave-uniform, kept in SGPRs) + (row inside the chunk: per-thread constant) with no
  // carries, so the per-step bookkeeping is scalar and each load costs an add and two range checks, as in the forward kernel.
  int t_oy[AP], t_ox[AP], t_r[AP];
  unsigned t_off[AP];
  if (P2 == 1) {
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const int r = arow + i * RPP_A;
      const int bb = r >> (p.lg_w + p.lg_h), oh = (r >> p.lg_w) & (p.Ho - 1), ow = r & (p.Wo - 1);
      t_r[i] = r;
      t_oy[i] = oh * p.s;
      t_ox[i] = ow * p.s;
      t_off[i] = (unsigned)(((bb * p.H + oh * p.s) * p.W + ow * p.s) * p.Ci * 4 + a_coloff);
    }
  }
  if (P2 == 2) {       // position-major: row r of a chunk = image (b0 + r) at the chunk's position
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const int r = arow + i * RPP_A;
      t_r[i] = r;
      t_oy[i] = 0;
      t_ox[i] = 0;
      t_off[i] = a_col_ok ? (unsigned)(r * p.H * p.W * p.Ci * 4 + a_coloff) : kOob;
    }
#pragma unroll
    for (int i = 0; i < BP; ++i) b_off[i] = b_col_ok ? (unsigned)((brow + i * RPP_B) * HoWo * p.Co * 4 + b_coloff) : kOob;
  }
  int s_m0 = m_begin;                                       // chunk origin of the NEXT gload (wave-uniform)
  // P2 == 2 (position-major): only the chunks whose output position is inside the image for this tap are visited.  They
  // form a rectangle of positions x (B / BKP) image chunks; the rectangle's chunk list is cut evenly over the ksplit
  // workgroups of this (tile, tap), so corner taps (a quarter of the positions live) and centre taps stay balanced.
  int q_oh = 0, q_ow = 0, q_bc = 0, q_left = 0, q_ow_lo = 0, q_ow_hi = 0, q_nb = 1;
  int nsteps_live = 0;
  if (P2 == 2) {
    const int oh_lo = max(0, (-dyk + p.s - 1) / p.s), oh_hi = min(p.Ho - 1, (p.H - 1 - dyk) / p.s);
    const int ow_lo = max(0, (-dxk + p.s - 1) / p.s), ow_hi = min(p.Wo - 1, (p.W - 1 - dxk) / p.s);
    const int rh = max(0, oh_hi - oh_lo + 1), rw = max(0, ow_hi - ow_lo + 1);
    q_nb = p.B / BKP;
    const int nlive = rh * rw * q_nb;
    const int cpz = (nlive + p.ksplit - 1) / p.ksplit;
    const int q_begin = min(nlive, zsplit * cpz), q_end = min(nlive, q_begin + cpz);
    nsteps_live = q_end - q_begin;
    q_left = nsteps_live;
    const int pi = q_begin / q_nb;
    q_bc = q_begin - pi * q_nb;
    q_oh = oh_lo + (rw > 0 ? pi / rw : 0);
    q_ow = ow_lo + (rw > 0 ? pi % rw : 0);
    q_ow_lo = ow_lo; q_ow_hi = ow_hi;
  }

  auto gload = [&](int) {     // loads the NEXT chunk in sequence (called once per step, in order)
    if (P2 == 2) {
      if (q_left > 0) {                                      // wave-uniform
        const int b0 = q_bc * BKP, pos = q_oh * p.Wo + q_ow;
        const unsigned offS = (unsigned)(((b0 * p.H + q_oh * p.s + dyk) * p.W + q_ow * p.s + dxk) * p.Ci * 4);
        const unsigned offY = (unsigned)((b0 * HoWo + pos) * p.Co * 4);
        // the per-thread part of an offset never changes here (t_off / b_off, poisoned for channel quads past Ci / Co); the chunk's
        // origin is wave-uniform and non-negative (only live positions are visited): it rides in the instruction's scalar offset
        // and the loads cost no vector instruction at all
#pragma unroll
        for (int i = 0; i < AP; ++i)
          regA[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsX, t_off[i], (int)offS, 0));
#pragma unroll
        for (int i = 0; i < BP; ++i)
          regB[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsY, b_off[i], (int)offY, 0));
        --q_left;
        if (++q_bc == q_nb) {
          q_bc = 0;
          if (++q_ow > q_ow_hi) { q_ow = q_ow_lo; ++q_oh; }
        }
      }
      return;
    }
    if (P2 == 1) {
      const int m0 = __builtin_amdgcn_readfirstlane(s_m0);
      const int b0 = m0 >> (p.lg_w + p.lg_h), oh0 = (m0 >> p.lg_w) & (p.Ho - 1), ow0 = m0 & (p.Wo - 1);
      const int yS = oh0 * p.s + dyk, xS = ow0 * p.s + dxk;
      const unsigned offS = (unsigned)(((b0 * p.H + yS) * p.W + xS) * p.Ci * 4);
      const int left = m_end - m0;                            // rows of this chunk inside the split
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        const bool ok = a_col_ok && t_r[i] < left && (unsigned)(t_oy[i] + yS) < (unsigned)p.H && (unsigned)(t_ox[i] + xS) < (unsigned)p.W;
        regA[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsX, ok ? t_off[i] + offS : kOob, 0, 0));
      }
      s_m0 = m0 + BKP;
    } else {
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const bool ok = a_col_ok && a_m[i] < m_end && (unsigned)(a_ohS[i] + dyk) < (unsigned)p.H &&
                      (unsigned)(a_owS[i] + dxk) < (unsigned)p.W;
      regA[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsX, ok ? a_off[i] : kOob, 0, 0));
      // advance to the same row of the next chunk
      a_m[i] += BKP;
      int ow = a_owS[i] + d_owS, oh = a_ohS[i] + d_ohS;
      unsigned off = a_off[i] + (unsigned)d_off;
      const bool c1 = ow >= WoS;
      ow -= c1 ? WoS : 0;
      oh += c1 ? p.s : 0;
      off += c1 ? (unsigned)corr_w : 0u;
      const bool c2 = oh >= HoS;
      oh -= c2 ? HoS : 0;
      off += c2 ? (unsigned)corr_h : 0u;
      a_owS[i] = ow; a_ohS[i] = oh; a_off[i] = off;
    }
    }
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      const bool ok = b_col_ok && b_m[i] < m_end;
      regB[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsY, ok ? b_off[i] : kOob, 0, 0));
      b_m[i] += BKP;
      b_off[i] += (unsigned)b_dstep;
    }
  };
  auto lstore = [&](int buf) {
    float* sa = smem + buf * STAGE;
    float* sb = sa + BKP * BM;
#pragma unroll
    for (int i = 0; i < AP; ++i) *reinterpret_cast<float4*>(sa + (arow + i * RPP_A) * BM + aq * 4) = regA[i];
#pragma unroll
    for (int i = 0; i < BP; ++i) *reinterpret_cast<float4*>(sb + (brow + i * RPP_B) * BN + bq * 4) = regB[i];
  };

  floatx16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  gload(0);
  lstore(0);
  gload(1);
  __syncthreads();
  const int fcol = lane & 31, fk = lane >> 5;
  const float* sa0 = smem + wm * WTM + fcol;
  const float* sb0 = smem + BKP * BM + wn * WTN + fcol;
  constexpr int ITERS = BKP / 2 / WAVES_K;                 // k-pairs per wave per chunk
  constexpr int PF = (MI * NI >= 4) ? 1 : (MI * NI >= 2 ? 2 : 4);   // k-pairs per fragment group: >= 4 MFMAs behind every LDS wait
  constexpr int NG = ITERS / PF;
  static_assert(ITERS % PF == 0 && NG >= 2, "fragment grouping");
  const int nsteps_run = P2 == 2 ? nsteps_live : nsteps;
  for (int step = 0; step < nsteps_run; ++step) {
    const int cur = step & 1;
    const float* sa = sa0 + cur * STAGE;
    const float* sb = sb0 + cur * STAGE;
    // Fragment groups are fetched one group ahead of the MFMAs that consume them (register double buffer); the
    // sched_barrier pins that order -- left alone, the scheduler sinks each ds_read to just before its MFMA and
    // the LDS latency is exposed once per k-pair.
    float af[2][PF][MI], bf[2][PF][NI];
    typedef const volatile __attribute__((address_space(3))) float* lds_cvf;      // volatile, but still an LDS (ds_read) access
    auto fetch = [&](int slot, int grp) {
#pragma unroll
      for (int q = 0; q < PF; ++q) {
        const int krow = 2 * (wk + (grp * PF + q) * WAVES_K) + fk;
        // volatile: each element stays ONE ds_read_b32 with its 16-bit immediate offset.  Left to the compiler, the two elements of
        // a k-row merge into a ds_read2_b32, whose 8-bit offsets do not reach the next k-row -- so it rebuilt the address with a
        // v_add_u32 per read, 28 vector instructions per 64 MFMAs, and on gfx950 every vector instruction in an fp32-MFMA loop is
        // matrix time lost one for one (tools/probes/mfma_valu_dual.hip); an LDS read costs a quarter of that
#pragma unroll
        for (int i = 0; i < MI; ++i) af[slot][q][i] = *(lds_cvf)(sa + krow * BM + i * 32);
#pragma unroll
        for (int j = 0; j < NI; ++j) bf[slot][q][j] = *(lds_cvf)(sb + krow * BN + j * 32);
      }
    };
    fetch(0, 0);
#pragma unroll
    for (int grp = 0; grp < NG; ++grp) {
      const int c = grp & 1;
      if (grp + 1 < NG) fetch(c ^ 1, grp + 1);
      if (grp == NG / 2) {
        lstore(cur ^ 1);
        gload(step + 2);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < PF; ++q)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][q][i], bf[c][q][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }

  if (WAVES_K > 1) {
    if (wk > 0) {
      float* red = smem + (((wk - 1) * WAVES_M * WAVES_N + wmn) * MI * NI) * 1024;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[((i * NI + j) * 16 + r) * 64 + lane] = acc[i][j][r];
    }
    __syncthreads();
    if (wk > 0) return;
#pragma unroll
    for (int w = 1; w < WAVES_K; ++w) {
      const float* red = smem + (((w - 1) * WAVES_M * WAVES_N + wmn) * MI * NI) * 1024;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] += red[((i * NI + j) * 16 + r) * 64 + lane];
    }
  }

  float* out = p.out + (size_t)zsplit * p.k * p.k * p.Ci * p.Co + (size_t)tap * p.Ci * p.Co;
  const bool direct = p.ksplit == 1;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int co = co0 + wn * WTN + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = ci0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (ci < p.Ci && co < p.Co) {
          float* q = out + (size_t)ci * p.Co + co;
          const float v = acc[i][j][r];
          if (direct) *q = (p.beta != 0.f ? p.beta * *q : 0.f) + p.scale * v;
          else *q = v;
        }
      }
    }
}

// Tap-grouped variant for 32-channel input tiles: one workgroup accumulates the TG = k taps of one kernel row
// (grid.y = kh) for a 32-wide ci tile.  Each loader thread owns ONE pixel row of the chunk for all TG taps, so the
// row bookkeeping is paid once per TG tap tiles, the dy tile is shared by the TG taps, and every k-pair feeds
// TG MFMAs per wave -- the loader VALU per MFMA drops ~4x against the per-tap kernel on 32/64-channel layers.
template <int TG, int BN, int WAVES_N, int WAVES_K>
__global__ __launch_bounds__(256) void conv_wgrad_tg_kernel(const WgradParams p) {
  static_assert(WAVES_N * WAVES_K == 4 && BN == 32 * WAVES_N, "one 32-wide co tile per wave column");
  constexpr int BKP = 32, BMc = 32, BM = TG * BMc;
  constexpr int TPR_B = BN / 4, RPP_B = 256 / TPR_B, BP = BKP / RPP_B;
  static_assert(BP >= 1 && BKP % RPP_B == 0, "B loader");
  constexpr int STAGE = BKP * (BM + BN);
  constexpr int RED = (WAVES_K > 1) ? (WAVES_K - 1) * WAVES_N * TG * 1024 : 0;
  constexpr int SMEM = (2 * STAGE > RED) ? 2 * STAGE : RED;
  constexpr unsigned kOob = 0x80000000u;
  __shared__ __attribute__((aligned(16))) float smem[SMEM];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave % WAVES_K, wn = wave / WAVES_K;
  const int L = p.xcd_swizzle ? bg::xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;   // (tile, kh, split), split slowest
  const int ntile = p.tiles_m * p.tiles_n;
  const int tile = L % ntile, kh = (L / ntile) % p.k, zsplit = L / (ntile * p.k);
  const int tile_m = tile / p.tiles_n, tile_n = tile % p.tiles_n;
  const int ci0 = tile_m * BMc, co0 = tile_n * BN;
  const int m_begin = zsplit * p.chunk;
  const int m_end = min(p.M, m_begin + p.chunk);
  const int nsteps = (m_end - m_begin + BKP - 1) / BKP;

  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.DY), 0, (int)p.dy_bytes, 0x00020000);
  const int arow = tid >> 3, aq = tid & 7;                 // one pixel row per thread, 8 float4 across the 32 ci
  const int brow = tid / TPR_B, bq = tid % TPR_B;
  const int HoWo = p.Ho * p.Wo;
  const bool a_col_ok = ci0 + aq * 4 < p.Ci, b_col_ok = co0 + bq * 4 < p.Co;
  const int dyk = kh - p.pt;
  const int WoS = p.Wo * p.s, HoS = p.Ho * p.s;
  const int dstep_b = BKP / HoWo;
  const int dstep_r = BKP - dstep_b * HoWo;
  const int dstep_oh = dstep_r / p.Wo, dstep_ow = dstep_r - dstep_oh * p.Wo;
  const int d_owS = dstep_ow * p.s, d_ohS = dstep_oh * p.s;
  const int pixB = p.Ci * 4, rowB = p.W * pixB;
  const int d_off = (dstep_b * p.H + d_ohS) * rowB + d_owS * pixB;
  const int corr_w = p.s * rowB - WoS * pixB, corr_h = (p.H - HoS) * rowB;
  int a_m, a_owS, a_ohS;
  unsigned a_off;                                          // byte offset of x[b, oh*s + dyk, ow*s - pl, ci0 + 4*aq] (tap kw = 0)
  {
    const int m = m_begin + arow;
    const int b = fastdiv(m, p.mul_hw, p.sh_hw);
    const int rem = m - b * HoWo;
    const int oh = fastdiv(rem, p.mul_w, p.sh_w);
    const int ow = rem - oh * p.Wo;
    a_m = m; a_ohS = oh * p.s; a_owS = ow * p.s;
    a_off = (unsigned)(((b * p.H + oh * p.s + dyk) * p.W + ow * p.s - p.pl) * pixB + (ci0 + aq * 4) * 4);
  }
  int b_m[BP];
  unsigned b_off[BP];
#pragma unroll
  for (int i = 0; i < BP; ++i) {
    b_m[i] = m_begin + brow + i * RPP_B;
    b_off[i] = (unsigned)(b_m[i] * p.Co * 4 + (co0 + bq * 4) * 4);
  }
  const int b_dstep = BKP * p.Co * 4;
  float4 regA[TG], regB[BP];

  auto gload = [&]() {
    const bool row_ok = a_col_ok && a_m < m_end && (unsigned)(a_ohS + dyk) < (unsigned)p.H;
#pragma unroll
    for (int t = 0; t < TG; ++t) {
      const bool ok = row_ok && (unsigned)(a_owS + t - p.pl) < (unsigned)p.W;
      regA[t] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsX, ok ? a_off + (unsigned)(t * pixB) : kOob, 0, 0));
    }
    a_m += BKP;
    int ow = a_owS + d_owS, oh = a_ohS + d_ohS;
    unsigned off = a_off + (unsigned)d_off;
    const bool c1 = ow >= WoS;
    ow -= c1 ? WoS : 0;
    oh += c1 ? p.s : 0;
    off += c1 ? (unsigned)corr_w : 0u;
    const bool c2 = oh >= HoS;
    oh -= c2 ? HoS : 0;
    off += c2 ? (unsigned)corr_h : 0u;
    a_owS = ow; a_ohS = oh; a_off = off;
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      const bool ok = b_col_ok && b_m[i] < m_end;
      regB[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsY, ok ? b_off[i] : kOob, 0, 0));
      b_m[i] += BKP;
      b_off[i] += (unsigned)b_dstep;
    }
  };
  auto lstore = [&](int buf) {
    float* sa = smem + buf * STAGE;
    float* sb = sa + BKP * BM;
#pragma unroll
    for (int t = 0; t < TG; ++t) *reinterpret_cast<float4*>(sa + arow * BM + t * BMc + aq * 4) = regA[t];
#pragma unroll
    for (int i = 0; i < BP; ++i) *reinterpret_cast<float4*>(sb + (brow + i * RPP_B) * BN + bq * 4) = regB[i];
  };

  floatx16 acc[TG];
#pragma unroll
  for (int t = 0; t < TG; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  gload();
  lstore(0);
  gload();
  __syncthreads();
  const int fcol = lane & 31, fk = lane >> 5;
  const float* sa0 = smem + fcol;
  const float* sb0 = smem + BKP * BM + wn * 32 + fcol;
  constexpr int ITERS = BKP / 2 / WAVES_K;
  for (int step = 0; step < nsteps; ++step) {
    const int cur = step & 1;
    const float* sa = sa0 + cur * STAGE;
    const float* sb = sb0 + cur * STAGE;
    float af[2][TG], bf[2];
    {
      const int krow = 2 * wk + fk;
#pragma unroll
      for (int t = 0; t < TG; ++t) af[0][t] = sa[krow * BM + t * BMc];
      bf[0] = sb[krow * BN];
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int c = it & 1;
      if (it + 1 < ITERS) {
        const int krow = 2 * (wk + (it + 1) * WAVES_K) + fk;
#pragma unroll
        for (int t = 0; t < TG; ++t) af[c ^ 1][t] = sa[krow * BM + t * BMc];
        bf[c ^ 1] = sb[krow * BN];
      }
      if (it == ITERS / 2) {
        lstore(cur ^ 1);
        gload();
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < TG; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][t], bf[c], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }

  if (WAVES_K > 1) {
    if (wk > 0) {
      float* red = smem + (((wk - 1) * WAVES_N + wn) * TG) * 1024;
#pragma unroll
      for (int t = 0; t < TG; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(t * 16 + r) * 64 + lane] = acc[t][r];
    }
    __syncthreads();
    if (wk > 0) return;
#pragma unroll
    for (int w = 1; w < WAVES_K; ++w) {
      const float* red = smem + (((w - 1) * WAVES_N + wn) * TG) * 1024;
#pragma unroll
      for (int t = 0; t < TG; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] += red[(t * 16 + r) * 64 + lane];
    }
  }

  const bool direct = p.ksplit == 1;
  const int co = co0 + wn * 32 + (lane & 31);
#pragma unroll
  for (int t = 0; t < TG; ++t) {
    float* out = p.out + (size_t)zsplit * p.k * p.k * p.Ci * p.Co + (size_t)(kh * p.k + t) * p.Ci * p.Co;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ci = ci0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (ci < p.Ci && co < p.Co) {
        float* q = out + (size_t)ci * p.Co + co;
        const float v = acc[t][r];
        if (direct) *q = (p.beta != 0.f ? p.beta * *q : 0.f) + p.scale * v;
        else *q = v;
      }
    }
  }
}

// MODE 0: grid.y = tap; A = x rows shifted by the tap (float4 over ci), B = dy rows.
// MODE 1: thin Ci (<= 4): the tap is folded into the M index, i = tap*Ci + ci (im2col columns gathered with
//         scalar loads); B = dy rows.  One launch covers all taps.
// MODE 2: thin Co (<= 4), stride 1: contraction runs over INPUT pixels, A = x rows as they lie in memory,
//         the tap is folded into the N index, j = tap*Co + co, B[k][j] = dy[pixel - tap offset][co].
template <int BM, int BN, int BKP, int WAVES_M, int WAVES_N, int WAVES_K, int MODE>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradParams p) {
  static_assert(WAVES_M * WAVES_N * WAVES_K == 4, "4 waves");
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int MI = WTM / 32, NI = WTN / 32;
  static_assert(MI >= 1 && NI >= 1 && WTM % 32 == 0 && WTN % 32 == 0, "wave tile");
  constexpr bool A_IM2COL = MODE == 1, B_IM2COL = MODE == 2;
  // float4 row loaders
  constexpr int TPR_A = BM / 4, RPP_A = 256 / TPR_A, AP = A_IM2COL ? 1 : (BKP + RPP_A - 1) / RPP_A;
  constexpr int TPR_B = BN / 4, RPP_B = 256 / TPR_B, BP = B_IM2COL ? 1 : (BKP + RPP_B - 1) / RPP_B;
  // scalar im2col loader: thread = (column, row group)
  constexpr int GW = A_IM2COL ? BM : BN;            // gathered width (padded tap*channel count)
  constexpr int NRG = 256 / GW;                     // row groups
  constexpr int GP = (A_IM2COL || B_IM2COL) ? (BKP + NRG - 1) / NRG : 1;
  constexpr int STAGE = BKP * (BM + BN);
  constexpr int RED = (WAVES_K > 1) ? (WAVES_K - 1) * WAVES_M * WAVES_N * MI * NI * 1024 : 0;
  constexpr int SMEM = (2 * STAGE > RED) ? 2 * STAGE : RED;
  __shared__ __attribute__((aligned(16))) float smem[SMEM];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave % WAVES_K, wmn = wave / WAVES_K;
  const int wm = wmn / WAVES_N, wn = wmn % WAVES_N;
  const int tile_m = blockIdx.x / p.tiles_n, tile_n = blockIdx.x % p.tiles_n;
  const int ci0 = tile_m * BM, co0 = tile_n * BN;
  const int tap = blockIdx.y;
  const int kh = tap / p.k, kw = tap % p.k;
  const int m_begin = blockIdx.z * p.chunk;
  const int m_end = min(p.M, m_begin + p.chunk);
  const int nsteps = (m_end - m_begin + BKP - 1) / BKP;

  const int arow = tid / TPR_A, aq = tid % TPR_A;
  const int brow = tid / TPR_B, bq = tid % TPR_B;
  const int HoWo = p.Ho * p.Wo;
  // im2col column owned by this thread
  const int gcol = tid % GW, grg = tid / GW;
  const int Cthin = A_IM2COL ? p.Ci : p.Co;
  const bool gvalid = (A_IM2COL || B_IM2COL) && grg < NRG && gcol < p.k * p.k * Cthin;
  const int gtap = gcol / max(Cthin, 1), gc = gcol - gtap * max(Cthin, 1);
  const int gkh = gtap / p.k, gkw = gtap - gkh * p.k;
  // pixel grid the contraction index runs over, and the gathered source tensor
  const int Hk = B_IM2COL ? p.H : p.Ho, Wk = B_IM2COL ? p.W : p.Wo;
  const int Hsrc = A_IM2COL ? p.H : p.Ho, Wsrc = A_IM2COL ? p.W : p.Wo;
  const float* gsrc = A_IM2COL ? p.X : p.DY;

  float4 regA[AP], regB[BP];
  float regG[GP];

  auto gather = [&](int mb) {
    // rows grg, grg+NRG, ...: decode the first, then step incrementally
    int m = mb + grg;
    int b = m / (Hk * Wk);
    int rem = m - b * (Hk * Wk);
    int py = rem / Wk, px = rem - py * Wk;
#pragma unroll
    for (int i = 0; i < GP; ++i) {
      float v = 0.f;
      const int r = grg + i * NRG;
      if (gvalid && r < BKP && m < m_end) {
        int sy, sx;
        if (A_IM2COL) { sy = py * p.s + gkh - p.pt; sx = px * p.s + gkw - p.pl; }
        else { sy = py - gkh + p.pt; sx = px - gkw + p.pl; }
        if ((unsigned)sy < (unsigned)Hsrc && (unsigned)sx < (unsigned)Wsrc)
          v = gsrc[((size_t)(b * Hsrc + sy) * Wsrc + sx) * Cthin + gc];
      }
      regG[i] = v;
      m += NRG;
      px += NRG;
      while (px >= Wk) { px -= Wk; if (++py == Hk) { py = 0; ++b; } }
    }
  };

  auto gload = [&](int step) {
    const int mb = m_begin + step * BKP;
    if (A_IM2COL || B_IM2COL) gather(mb);
    if (!A_IM2COL) {
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        const int r = arow + i * RPP_A;
        const int m = mb + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < BKP && m < m_end && ci0 + aq * 4 < p.Ci) {
          if (MODE == 2) {
            v = *reinterpret_cast<const float4*>(p.X + (size_t)m * p.Ci + ci0 + aq * 4);
          } else {
            const int b = m / HoWo;
            const int rem = m - b * HoWo;
            const int oh = rem / p.Wo;
            const int ow = rem - oh * p.Wo;
            const int iy = oh * p.s + kh - p.pt, ix = ow * p.s + kw - p.pl;
            if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
              v = *reinterpret_cast<const float4*>(p.X + ((size_t)(b * p.H + iy) * p.W + ix) * p.Ci + ci0 + aq * 4);
          }
        }
        regA[i] = v;
      }
    }
    if (!B_IM2COL) {
#pragma unroll
      for (int i = 0; i < BP; ++i) {
        const int r = brow + i * RPP_B;
        const int m = mb + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < BKP && m < m_end && co0 + bq * 4 < p.Co) v = *reinterpret_cast<const float4*>(p.DY + (size_t)m * p.Co + co0 + bq * 4);
        regB[i] = v;
      }
    }
  };
  auto lstore = [&](int buf) {
    float* sa = smem + buf * STAGE;
    float* sb = sa + BKP * BM;
    if (A_IM2COL || B_IM2COL) {
      float* sg = A_IM2COL ? sa : sb;
      if (grg < NRG) {
#pragma unroll
        for (int i = 0; i < GP; ++i) {
          const int r = grg + i * NRG;
          if (r < BKP) sg[r * GW + gcol] = regG[i];
        }
      }
    }
    if (!A_IM2COL) {
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        const int r = arow + i * RPP_A;
        if (r < BKP) *reinterpret_cast<float4*>(sa + r * BM + aq * 4) = regA[i];
      }
    }
    if (!B_IM2COL) {
#pragma unroll
      for (int i = 0; i < BP; ++i) {
        const int r = brow + i * RPP_B;
        if (r < BKP) *reinterpret_cast<float4*>(sb + r * BN + bq * 4) = regB[i];
      }
    }
  };

  floatx16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (nsteps > 0) {
    gload(0);
    lstore(0);
  }
  __syncthreads();
  const int fcol = lane & 31, fk = lane >> 5;
  for (int step = 0; step < nsteps; ++step) {
    const int cur = step & 1;
    if (step + 1 < nsteps) gload(step + 1);
    const float* sa = smem + cur * STAGE + wm * WTM + fcol;
    const float* sb = smem + cur * STAGE + BKP * BM + wn * WTN + fcol;
#pragma unroll 4
    for (int it = 0; it < BKP / 2 / WAVES_K; ++it) {
      const int krow = 2 * (wk + it * WAVES_K) + fk;
      float af[MI], bf[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = sa[krow * BM + i * 32];
#pragma unroll
      for (int j = 0; j < NI; ++j) bf[j] = sb[krow * BN + j * 32];
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (step + 1 < nsteps) lstore(cur ^ 1);
    __syncthreads();
  }

  // ---- reduce the WAVES_K partial accumulators through LDS (staging buffers are dead now)
  if (WAVES_K > 1) {
    if (wk > 0) {
      float* red = smem + (((wk - 1) * WAVES_M * WAVES_N + wmn) * MI * NI) * 1024;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[((i * NI + j) * 16 + r) * 64 + lane] = acc[i][j][r];
    }
    __syncthreads();
    if (wk > 0) return;
#pragma unroll
    for (int w = 1; w < WAVES_K; ++w) {
      const float* red = smem + (((w - 1) * WAVES_M * WAVES_N + wmn) * MI * NI) * 1024;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] += red[((i * NI + j) * 16 + r) * 64 + lane];
    }
  }

  // ---- store: acc reg r of lane l holds C[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]
  const int kk = p.k * p.k;
  float* out = p.out + (size_t)blockIdx.z * kk * p.Ci * p.Co;
  const bool direct = p.ksplit == 1;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int col = wn * WTN + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        size_t idx;
        bool ok;
        if (MODE == 0) {
          const int ci = ci0 + row, co = co0 + col;
          ok = ci < p.Ci && co < p.Co;
          idx = ((size_t)tap * p.Ci + ci) * p.Co + co;
        } else if (MODE == 1) {
          const int co = co0 + col;
          ok = row < kk * p.Ci && co < p.Co;                 // row = tap*Ci + ci
          idx = (size_t)row * p.Co + co;
        } else {
          const int ci = ci0 + row;
          const int t = col / p.Co, co = col - t * p.Co;     // col = tap*Co + co
          ok = ci < p.Ci && col < kk * p.Co;
          idx = ((size_t)t * p.Ci + ci) * p.Co + co;
        }
        if (ok) {
          float* q = out + idx;
          const float v = acc[i][j][r];
          if (direct) *q = (p.beta != 0.f ? p.beta * *q : 0.f) + p.scale * v;
          else *q = v;
        }
      }
    }
}

// one thread per output element (tap, ci, co) and pixel slice (grid.y)
__global__ __launch_bounds__(256) void wgrad_direct_kernel(const WgradParams p) {
  const int nout = p.k * p.k * p.Ci * p.Co;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= nout) return;
  const int co = e % p.Co;
  const int ci = (e / p.Co) % p.Ci;
  const int tap = e / (p.Co * p.Ci);
  const int kh = tap / p.k, kw = tap % p.k;
  const int m_begin = blockIdx.y * p.chunk, m_end = min(p.M, m_begin + p.chunk);
  const int HoWo = p.Ho * p.Wo;
  int b = m_begin / HoWo;
  int rem = m_begin - b * HoWo;
  int oh = rem / p.Wo, ow = rem - oh * p.Wo;
  float acc = 0.f;
  for (int m = m_begin; m < m_end; ++m) {
    const int iy = oh * p.s + kh - p.pt, ix = ow * p.s + kw - p.pl;
    if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
      acc = fmaf(p.X[((size_t)(b * p.H + iy) * p.W + ix) * p.Ci + ci], p.DY[(size_t)m * p.Co + co], acc);
    if (++ow == p.Wo) {
      ow = 0;
      if (++oh == p.Ho) { oh = 0; ++b; }
    }
  }
  float* q = p.out + (size_t)blockIdx.y * nout + e;
  if (p.ksplit == 1) *q = (p.beta != 0.f ? p.beta * *q : 0.f) + p.scale * acc;
  else *q = acc;
}

// ------------------------------------------------------------------------------------------------
// thin-Co filter gradient on MFMA (stride 1, k*Co <= 16, Ci = 16 or 32: the generator's last conv 32->3):
// the (kw, co) pairs are the 16 MFMA columns and the contraction runs over the pixels of one input row,
//   dW[kh][kw][ci][co] = sum_{b, y', x'} x[b][y'][x'][ci] * dy[b][y' - kh + pt][x' - kw + pl][co]
//                     = sum_{b, y'}  X_row(b, y')^T [ci x x']  *  B_kh [x' x (kw,co)],   B_kh[x'][kw*Co+co] = dyrow[(x' + pl - kw)*Co + co]
// so each x element is loaded once (registers, A layout of v_mfma_f32_16x16x4_f32) and re-used by the k kernel rows; the dy
// rows of a block of image rows sit in LDS with a zero halo (out-of-image rows / columns read as 0, no branches).
// One workgroup = blocks of kTcRows image rows, one input row per wave at a time; partial dW per workgroup -> slab.
// MFMA-bound: 2*M*k*k*Ci*16 flop (15 of the 16 columns useful for RGB).
// ------------------------------------------------------------------------------------------------
typedef float floatx4 __attribute__((ext_vector_type(4)));
constexpr int kTcRows = 16, kTcHalo = 8;

template <int MT, int K>     // Ci = 16 * MT, K = kernel size (rows of taps kept in accumulators)
__global__ __launch_bounds__(256) void conv_wgrad_thin_co_kernel(const WgradParams p, int nblocks, int blocks_per_img) {
  extern __shared__ __attribute__((aligned(16))) float tc_lds[];
  const int W = p.W, H = p.H, Ci = 16 * MT, Co = p.Co;
  const int RS = W * Co + 2 * kTcHalo;                       // dy row stride in LDS (zero halo on both sides)
  const int nrows = kTcRows + K - 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  floatx4 acc[K][MT];
#pragma unroll
  for (int kh = 0; kh < K; ++kh)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[kh][mt] = floatx4{0.f, 0.f, 0.f, 0.f};
  // column n = kw*Co + co of the B operand reads dyrow[(x' + pl - kw)*Co + co]; unused columns read a halo zero
  const int ncol = K * Co;
  const int kw_n = li / Co, co_n = li - kw_n * Co;
  const int b_base = li < ncol ? kTcHalo + (kq + p.pl - kw_n) * Co + co_n : 0;
  const int b_step = li < ncol ? 4 * Co : 0;                 // per k-step of 4 pixels
  const int ksteps = W / 4;

  for (int blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
    const int b = blk / blocks_per_img, y0 = (blk - b * blocks_per_img) * kTcRows;
    __syncthreads();                                          // previous block's fragments consumed
    // dy rows y0 - pt .. y0 + kTcRows - 1 + (K - 1 - pt), zero outside the image, zero halos
    if (((W * Co) & 3) == 0) {                                // float4-addressable rows (see conv_wgrad_thin_ci_kernel)
      const int q4 = RS / 4;
      bg::stage_rows_f4<4>(tc_lds, nrows, q4, RS, tid, [&](int r, int c4) -> const float* {
        const int oy = y0 - (K - 1 - p.pt) + r, e = c4 * 4 - kTcHalo;
        return ((unsigned)oy < (unsigned)p.Ho && (unsigned)e < (unsigned)(W * Co)) ? p.DY + ((size_t)b * p.Ho + oy) * W * Co + e : nullptr;
      });
    } else {
      for (int idx = tid; idx < nrows * RS; idx += 256) {
        const int r = idx / RS, c = idx - r * RS;
        const int oy = y0 - (K - 1 - p.pt) + r, e = c - kTcHalo;
        float v = 0.f;
        if ((unsigned)oy < (unsigned)p.Ho && (unsigned)e < (unsigned)(W * Co)) v = p.DY[((size_t)b * p.Ho + oy) * W * Co + e];
        tc_lds[idx] = v;
      }
    }
    __syncthreads();
    // this wave's work list: (row r = wave + 4*j, pixel group g of 32) flattened; the x registers of item it+1 are
    // loaded while the MFMAs of item it run
    const int groups = (ksteps + 7) / 8;
    const int myrows = min(kTcRows, H - y0);
    const int nitems = ((myrows - wave + 3) / 4) * groups;       // rows wave, wave+4, ... < myrows
    const float* xblk = p.X + ((size_t)b * H + y0) * W * Ci + kq * Ci + li;
    auto xload = [&](int it, float (&a)[8][MT]) {
      const int j = it / groups, g = it - j * groups;
      const float* xrow = xblk + (size_t)(wave + 4 * j) * W * Ci + (size_t)g * 32 * Ci;
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[q][mt] = (it < nitems && g * 8 + q < ksteps) ? xrow[(size_t)q * 4 * Ci + mt * 16] : 0.f;
    };
    auto compute = [&](int it, const float (&a)[8][MT]) {
      const int j = it / groups, g = it - j * groups;
      const int r = wave + 4 * j;
      // x row y' pairs with dy row y' - kh + pt = LDS row r + (K - 1) - kh
#pragma unroll
      for (int kh = 0; kh < K; ++kh) {
        const float* brow = tc_lds + (r + (K - 1) - kh) * RS + b_base + g * 8 * b_step;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float bv = brow[q * b_step];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[kh][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q][mt], bv, acc[kh][mt], 0, 0, 0);
        }
      }
    };
    float a0[8][MT], a1[8][MT];
    xload(0, a0);
    for (int it = 0; it < nitems; it += 2) {
      xload(it + 1, a1);
      compute(it, a0);
      xload(it + 2, a0);
      if (it + 1 < nitems) compute(it + 1, a1);
    }
  }
  // cross-wave sum through LDS, then the slab in dW layout [kh][kw][ci][co]; reg r of lane l = D[ci = 4*(l>>4) + r][n = l&15]
  __syncthreads();
  float* red = tc_lds;                                       // [4 waves][K*MT*4][64]
#pragma unroll
  for (int kh = 0; kh < K; ++kh)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) red[((wave * K + kh) * MT * 4 + mt * 4 + rr) * 64 + lane] = acc[kh][mt][rr];
  __syncthreads();
  float* out = p.out + (size_t)blockIdx.x * K * K * Ci * Co;
  for (int e = tid; e < K * MT * 4 * 64; e += 256) {
    const int l = e & 63, slot = e >> 6;                      // slot = (kh*MT + mt)*4 + rr
    const int rr = slot & 3, mt = (slot >> 2) % MT, kh = slot / (4 * MT);
    const int n = l & 15;
    if (n >= ncol) continue;
    const float v = (red[(0 * K * MT * 4 + slot) * 64 + l] + red[(1 * K * MT * 4 + slot) * 64 + l]) +
                    (red[(2 * K * MT * 4 + slot) * 64 + l] + red[(3 * K * MT * 4 + slot) * 64 + l]);
    const int kw = n / Co, co = n - kw * Co, ci = mt * 16 + 4 * (l >> 4) + rr;
    out[((size_t)(kh * K + kw) * Ci + ci) * Co + co] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// thin-Ci filter gradient on MFMA (k*Ci <= 16, Co a multiple of 16: the critic's first conv RGB -> 32, MNIST 1 -> 64):
// the mirror image of the kernel above -- the (kw, ci) pairs are the 16 MFMA rows, the contraction runs over the
// output pixels of one row, and because x is NHWC with <= 3 channels the A operand is a contiguous window of the x row:
//   dW[kh][kw][ci][co] = sum_{b, oy, ox} xrow(b, oy*s + kh - pt)[(ox*s - pl)*Ci + (kw*Ci + ci)] * dy[b][oy][ox][co]
// dy elements are loaded once into registers (B layout) and re-used by the k kernel rows; x rows sit in LDS with a zero halo.
// ------------------------------------------------------------------------------------------------
constexpr int kTiHalo = 16;

template <int NT, int K>     // Co = 16 * NT
__global__ __launch_bounds__(256) void conv_wgrad_thin_ci_kernel(const WgradParams p, int nblocks, int blocks_per_img, int RB) {
  extern __shared__ __attribute__((aligned(16))) float ti_lds[];
  const int W = p.W, H = p.H, Ci = p.Ci, Co = 16 * NT, Ho = p.Ho, Wo = p.Wo, st = p.s;
  const int RS = W * Ci + 2 * kTiHalo;
  const int nrows = (RB - 1) * st + K;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  floatx4 acc[K][NT];
#pragma unroll
  for (int kh = 0; kh < K; ++kh)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[kh][nt] = floatx4{0.f, 0.f, 0.f, 0.f};
  const int nrow_used = K * Ci;                               // MFMA rows in use: i = kw*Ci + ci
  const int a_base = li < nrow_used ? kTiHalo + (kq * st - p.pl) * Ci + li : 0;
  const int a_step = li < nrow_used ? 4 * st * Ci : 0;       // per k-step of 4 output pixels
  const int ksteps = Wo / 4;
  const int groups = (ksteps + 7) / 8;

  for (int blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
    const int b = blk / blocks_per_img, oy0 = (blk - b * blocks_per_img) * RB;
    __syncthreads();
    // x rows oy0*s - pt .. , zero outside the image, zero halos
    if (((W * Ci) & 3) == 0) {
      // rows are float4-addressable (x is 16-byte aligned, the halo is 16 floats): a quarter of the load instructions of the
      // scalar copy below, which issued 31 dword loads per thread and block on the 128-pixel critic (round 3)
      const int q4 = RS / 4;
      bg::stage_rows_f4<4>(ti_lds, nrows, q4, RS, tid, [&](int r, int c4) -> const float* {
        const int yy = oy0 * st - p.pt + r, e = c4 * 4 - kTiHalo;
        return ((unsigned)yy < (unsigned)H && (unsigned)e < (unsigned)(W * Ci)) ? p.X + ((size_t)b * H + yy) * W * Ci + e : nullptr;
      });
    } else {
      for (int idx = tid; idx < nrows * RS; idx += 256) {
        const int r = idx / RS, c = idx - r * RS;
        const int yy = oy0 * st - p.pt + r, e = c - kTiHalo;
        float v = 0.f;
        if ((unsigned)yy < (unsigned)H && (unsigned)e < (unsigned)(W * Ci)) v = p.X[((size_t)b * H + yy) * W * Ci + e];
        ti_lds[idx] = v;
      }
    }
    __syncthreads();
    const int myrows = min(RB, Ho - oy0);
    const int nitems = ((myrows - wave + 3) / 4) * groups;
    const float* dyblk = p.DY + ((size_t)b * Ho + oy0) * Wo * Co + kq * Co + li;
    auto yload = [&](int it, float (&bv)[8][NT]) {
      const int j = it / groups, g = it - j * groups;
      const float* dyrow = dyblk + (size_t)(wave + 4 * j) * Wo * Co + (size_t)g * 32 * Co;
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bv[q][nt] = (it < nitems && g * 8 + q < ksteps) ? dyrow[(size_t)q * 4 * Co + nt * 16] : 0.f;
    };
    auto compute = [&](int it, const float (&bv)[8][NT]) {
      const int j = it / groups, g = it - j * groups;
      const int r = wave + 4 * j;
#pragma unroll
      for (int kh = 0; kh < K; ++kh) {
        const float* arow = ti_lds + (r * st + kh) * RS + a_base + g * 8 * a_step;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float av = arow[q * a_step];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[kh][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[q][nt], acc[kh][nt], 0, 0, 0);
        }
      }
    };
    float b0[8][NT], b1[8][NT];
    yload(0, b0);
    for (int it = 0; it < nitems; it += 2) {
      yload(it + 1, b1);
      compute(it, b0);
      yload(it + 2, b0);
      if (it + 1 < nitems) compute(it + 1, b1);
    }
  }
  // cross-wave sum through LDS in wave order (deterministic); reg rr of lane l = D[i = 4*(l>>4) + rr][j = l&15]
  float* red = ti_lds;                                       // [K*NT*4][64]
  for (int w = 0; w < 4; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int kh = 0; kh < K; ++kh)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            float* q = red + ((kh * NT + nt) * 4 + rr) * 64 + lane;
            *q = (w == 0 ? 0.f : *q) + acc[kh][nt][rr];
          }
    }
  }
  __syncthreads();
  float* out = p.out + (size_t)blockIdx.x * K * K * Ci * Co;
  for (int e = tid; e < K * NT * 4 * 64; e += 256) {
    const int l = e & 63, slot = e >> 6;                      // slot = (kh*NT + nt)*4 + rr
    const int rr = slot & 3, nt = (slot >> 2) % NT, kh = slot / (4 * NT);
    const int i = 4 * (l >> 4) + rr;
    if (i >= nrow_used) continue;
    const int kw = i / Ci, ci = i - kw * Ci;
    out[((size_t)(kh * K + kw) * Ci + ci) * Co + nt * 16 + (l & 15)] = red[e];
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// 16 -> 32 channel, 5x5, stride-2 filter gradient (the 128x128 critic's second conv and the generator's ConvT 32 -> 16):
//   dW[kh][kw][ci][co] = sum_{b, oy, ox} x[b, 2oy + kh - 1, 2ox + kw - 1, ci] * dy[b, oy, ox, co]
// The tap-grouped kernel pads the 16 input channels to a 32-row MFMA tile and re-reads x once per kernel row; here the 16
// channels ARE the rows of v_mfma_f32_16x16x4_f32, the contraction runs over 4 output pixels of a row per MFMA, and a strip's 7
// input rows are staged in LDS once (register-prefetched, double-buffered) and serve all 25 taps.  x rows are de-interleaved by
// column parity so that the 4 pixels of a k-step (2 columns apart) read 64 consecutive words.  Every wave keeps the full
// 25 x 2 accumulator set (200 registers): wave w owns output row (w >> 1) of the strip and half (w & 1) of its pixels; the
// four partial sets are summed through LDS once, at the end of the persistent loop, into the workgroup's slab.
// ------------------------------------------------------------------------------------------------------------------------
constexpr int kC16Rows = 7;

template <int WO>
__global__ __launch_bounds__(256) void conv_wgrad_c16_kernel(const WgradParams p, int nstrips, int strips_per_img) {
  constexpr int W = 2 * WO, PS = (WO + 2) * 16, RS = 2 * PS, KSW = WO / 8;
  constexpr int PFX = kC16Rows * W * 4 / 256;                // float4 of x per thread and strip
  constexpr unsigned kOob = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) float c16_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int r = wave >> 1, half = wave & 1;
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.DY), 0, (int)p.dy_bytes, 0x00020000);

  for (int i = tid; i < 2 * kC16Rows * 2 * 2 * 16; i += 256) {   // halo slots (column -1 / -2 and W / W+1) of every plane: zero, never written
    const int e = i & 15, side = (i >> 4) & 1, plane = i >> 5;     // plane = (buffer, row, parity)
    c16_lds[plane * PS + (side ? (WO + 1) * 16 : 0) + e] = 0.f;
  }
  floatx4 acc[25][2];
#pragma unroll
  for (int t = 0; t < 25; ++t) { acc[t][0] = floatx4{0.f, 0.f, 0.f, 0.f}; acc[t][1] = floatx4{0.f, 0.f, 0.f, 0.f}; }

  float4 pf[PFX];
  float bn[KSW][2], bc[KSW][2];
  auto prefetch = [&](int strip) {
    const int b = strip / strips_per_img, oy0 = (strip - b * strips_per_img) * 2;
    const bool live = strip < nstrips;
#pragma unroll
    for (int i = 0; i < PFX; ++i) {
      const int item = i * 256 + tid;
      const int rr = item / (W * 4), rem = item - rr * (W * 4);
      const int y = 2 * oy0 - 1 + rr;
      const bool ok = live && (unsigned)y < (unsigned)p.H;
      pf[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsX, ok ? (unsigned)(((b * p.H + y) * W) * 64 + rem * 16) : kOob, 0, 0));
    }
    const unsigned ybase = (unsigned)((((b * p.Ho + oy0 + r) * WO) + half * (WO / 2) + kq) * 128 + li * 4);
#pragma unroll
    for (int ks = 0; ks < KSW; ++ks)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        bn[ks][nt] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsY, live ? ybase + (unsigned)(ks * 4 * 128 + nt * 64) : kOob, 0, 0));
  };
  auto stash = [&](int buf) {
    float* dst = c16_lds + buf * kC16Rows * RS;
#pragma unroll
    for (int i = 0; i < PFX; ++i) {
      const int item = i * 256 + tid;
      const int rr = item / (W * 4), rem = item - rr * (W * 4);
      const int x = rem >> 2, q = rem & 3;
      *reinterpret_cast<float4*>(dst + rr * RS + (x & 1) * PS + ((x >> 1) + 1) * 16 + q * 4) = pf[i];
    }
  };

  int strip = blockIdx.x;
  prefetch(strip);
  stash(0);
  __syncthreads();
  int buf = 0;
  for (; strip < nstrips; strip += gridDim.x, buf ^= 1) {
#pragma unroll
    for (int ks = 0; ks < KSW; ++ks) { bc[ks][0] = bn[ks][0]; bc[ks][1] = bn[ks][1]; }
    prefetch(strip + gridDim.x);
    const float* xb = c16_lds + buf * kC16Rows * RS + (2 * r) * RS + (half * (WO / 2) + kq + 1) * 16 + li;
#pragma unroll
    for (int ks = 0; ks < KSW; ++ks)
#pragma unroll
      for (int kh = 0; kh < 5; ++kh)
#pragma unroll
        for (int kw = 0; kw < 5; ++kw) {
          // column 2*ox + kw - 1: parity (kw + 1) & 1, half-column ox + floor((kw - 1) / 2)
          const int par = (kw + 1) & 1, fl = kw == 0 ? -1 : (kw - 1) / 2;
          const float a = xb[kh * RS + par * PS + (4 * ks + fl) * 16];
          acc[kh * 5 + kw][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bc[ks][0], acc[kh * 5 + kw][0], 0, 0, 0);
          acc[kh * 5 + kw][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bc[ks][1], acc[kh * 5 + kw][1], 0, 0, 0);
        }
    stash(buf ^ 1);
    __syncthreads();
  }
  // cross-wave sum in wave order (deterministic); reg rr of lane l = D[ci = 4*(l>>4) + rr][co = l & 15]
  float* red = c16_lds;                                       // [25*2*4][64]
  for (int w = 0; w < 4; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < 25; ++t)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            float* q = red + ((t * 2 + nt) * 4 + rr) * 64 + lane;
            *q = (w == 0 ? 0.f : *q) + acc[t][nt][rr];
          }
    }
  }
  __syncthreads();
  float* out = p.out + (size_t)blockIdx.x * 25 * 16 * 32;
  for (int e = tid; e < 25 * 2 * 4 * 64; e += 256) {
    const int l = e & 63, slot = e >> 6;
    const int rr = slot & 3, nt = (slot >> 2) & 1, t = slot >> 3;
    out[(size_t)(t * 16 + 4 * (l >> 4) + rr) * 32 + nt * 16 + (l & 15)] = red[e];
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// Strip-resident filter gradient: 5x5, stride 2, even H / W, Ci a multiple of 32, Co a multiple of 64, Wo = 8 / 16 / 32 -- the
// critic's inner convs and the generator's inner transposed convs (demo_celeba.py:62-87,104-119) on maps of 16 ... 64 pixels.
//   dW[kh][kw][ci][co] = sum_{b, oy, ox} x[b, 2oy + kh - 1, 2ox + kw - 1, ci] * dy[b, oy, ox, co]
// The per-tap kernel above re-reads x AND dy from L2 once per tap (25 x) and issues one 1-KiB vector-memory instruction per
// 2.7 ... 8 MFMAs on the 32- and 64-channel layers; every such instruction costs the SIMD's matrix pipe ~58 cycles.  Here a
// STRIP of 64 output pixels (R = 64 / Wo whole output rows of one image) is staged in LDS once -- the 2R + 3 input rows of a
// 32-channel slice with a zero halo column on each side, and the strip's dy rows of a 64-channel slice -- and serves all 25 taps:
// the tap shift is an LDS address (an immediate), one vector-memory instruction per ~90 MFMAs.  A workgroup = one (32 ci x 64 co)
// channel tile and a contiguous range of strips; its 4 waves = (tap half) x (co half), each with 13 accumulator tiles of 32 x 32:
// 12 taps of its own and the 25th tap, which the two halves share pair by pair (12.5 taps' worth of MFMAs each); LDS is double-buffered and the
// next strip is loaded in pieces BETWEEN the k-steps of the current one (a few registers in flight, none held across a strip).
// One partial slab per workgroup -> wgrad_reduce_* (fixed order: deterministic).
// ------------------------------------------------------------------------------------------------------------------------
template <int LGWO>
struct StripGeom {
  static constexpr int WO = 1 << LGWO, W = 2 * WO, R = 64 / WO, RX = 2 * R + 3, WP = W + 3;
  static constexpr int XROW = WP * 32, XBUF = RX * XROW, YBUF = 64 * 64, BUF = XBUF + YBUF;
  static constexpr int NX = RX * W * 8, NLX = (NX + 255) / 256, NL = NLX + 4;      // float4 loads per thread and strip: x, then dy
  static constexpr size_t lds_bytes = (size_t)2 * BUF * sizeof(float);
};

typedef float floatx2 __attribute__((ext_vector_type(2)));

// Ordered building blocks of the strip kernel's k loop: volatile asm statements keep their program order, so the LDS reads sit
// exactly where they are written -- between the MFMAs (left to itself the scheduler sinks every read burst to just in front of
// its first use and the matrix pipe waits out an LDS round trip per pair; sched_group_barrier patterns made it worse).
template <int O0>
__device__ inline void strip_read2(floatx2& dst, unsigned addr) {      // dwords at addr + O0 * 256 B and addr + (O0 + 2) * 256 B
  asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(dst) : "v"(addr), "n"(O0), "n"(O0 + 2));
}
__device__ inline void strip_mfma(floatx16& acc, float a, float b) {
  asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// fragments of pair Q (pixels 4Q .. 4Q + 3): T < 13 the tap's x fragments, T == 13 the dy fragments
template <int LGWO, int Q, int T>
__device__ inline void strip_fetch(floatx2 (&af)[13], floatx2& bf, const unsigned (&aaddr)[13], unsigned baddr) {
  constexpr int WO = 1 << LGWO, WP = 2 * WO + 3, pix = 4 * Q, oyl = pix >> LGWO, ox = pix & (WO - 1);
  if constexpr (T < 13) strip_read2<oyl * WP + ox>(af[T], aaddr[T]);
  else strip_read2<pix>(bf, baddr);
}
// The next strip's pieces are ordered asm statements too: a buffer load the compiler knows nothing about (so it cannot drain
// vmcnt in front of the MFMAs), written to LDS LAG pairs later behind a counted `s_waitcnt vmcnt`.
typedef float floatx4v __attribute__((ext_vector_type(4)));
__device__ inline void strip_gload(floatx4v& dst, unsigned off, __amdgpu_buffer_rsrc_t rs) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(off), "s"(rs));
}
template <int VM>
__device__ inline void strip_lwrite(unsigned addr, const floatx4v& v) {
  asm volatile("s_waitcnt vmcnt(%2)\n\tds_write_b128 %0, %1" : : "v"(addr), "v"(v), "n"(VM) : "memory");
}
template <int NL_, int PPP_>
struct StripPieces {                                            // per-thread state of the next strip's load, all in registers
  static constexpr int NL = NL_, PPP = PPP_, LAG = 3;
  floatx4v pf[NL_];
  unsigned off[NL_];                                            // global byte offsets of the NEXT strip's pieces (0x80000000: reads 0)
  unsigned dst[NL_];                                            // LDS byte addresses in the OTHER buffer
  __amdgpu_buffer_rsrc_t rsX, rsY;
  int nlx;
};
// MFMA number I (0 .. 25) of pair Q from slot C; behind it ride, in this order of I: the 14 fragment reads of pair Q + 1
// (I = 0 .. 13: the last is then 12 MFMAs old at the next pair's wait), the pair's PPP requests of next-strip pieces
// (I = 14, 16, 18) and the LDS writes of the pieces requested LAG pairs earlier (I = 19, 21, 23).  Nothing but MFMAs and these
// single instructions sits between two MFMAs: with the loads and writes in FRONT of a pair (first version) the matrix pipe
// idled ~200 cycles per pair (SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES) = 0.866).
template <int LGWO, int Q, int I, class PC>
__device__ inline void strip_steps(floatx16 (&acc)[13], floatx2 (&af)[2][13], floatx2 (&bf)[2], const unsigned (&aaddr)[13], unsigned baddr,
                                   bool mine, PC& pc) {
  if constexpr (I < 26) {
    constexpr int C = Q & 1, H = I / 13, T = I % 13;
    // slot 12 is the tap both wave halves share (25 taps = 12 + 12 + 1): each half runs it on every other pair, so that both
    // issue 12.5 taps' worth of MFMAs per pair on average; the two partial tiles are summed in the epilogue
    if (T < 12 || mine) {                                           // wave-uniform
      if constexpr (H == 0) strip_mfma(acc[T], af[C][T].x, bf[C].x);
      else strip_mfma(acc[T], af[C][T].y, bf[C].y);
    }
    if constexpr (Q + 1 < 16 && I <= 13) strip_fetch<LGWO, Q + 1, I>(af[C ^ 1], bf[C ^ 1], aaddr, baddr);
    constexpr int P = PC::PPP, NL = PC::NL, LAG = PC::LAG;
    if constexpr (I >= 14 && I <= 18 && ((I - 14) & 1) == 0) {
      constexpr int u = (I - 14) / 2, l = Q * P + u;
      if constexpr (u < P && l < NL) {
        if (l < pc.nlx) strip_gload(pc.pf[l], pc.off[l], pc.rsX);   // compile-time after inlining: nlx is a constant of the instantiation
        else strip_gload(pc.pf[l], pc.off[l], pc.rsY);
      }
    }
    if constexpr (I >= 19 && I <= 23 && ((I - 19) & 1) == 0) {
      constexpr int u = (I - 19) / 2, lw = (Q - LAG) * P + u;
      if constexpr (Q >= LAG && u < P && lw < NL) {
        constexpr int issued = (Q + 1) * P < NL ? (Q + 1) * P : NL;   // pieces requested so far (this pair's included)
        strip_lwrite<issued - lw - 1>(pc.dst[lw], pc.pf[lw]);
      }
    }
    strip_steps<LGWO, Q, I + 1>(acc, af, bf, aaddr, baddr, mine, pc);
  }
}
template <int LGWO, int T>
__device__ inline void strip_fetch_first(floatx2 (&af)[2][13], floatx2 (&bf)[2], const unsigned (&aaddr)[13], unsigned baddr) {
  if constexpr (T <= 13) {
    strip_fetch<LGWO, 0, T>(af[0], bf[0], aaddr, baddr);
    strip_fetch_first<LGWO, T + 1>(af, bf, aaddr, baddr);
  }
}
template <int LGWO, int Q, class PC>
__device__ inline void strip_pairs(floatx16 (&acc)[13], floatx2 (&af)[2][13], floatx2 (&bf)[2], const unsigned (&aaddr)[13], unsigned baddr,
                                   PC& pc, int th) {
  if constexpr (Q < 16) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // this pair's fragments: issued during the previous pair, 12+ MFMAs old
    strip_steps<LGWO, Q, 0>(acc, af, bf, aaddr, baddr, (Q & 1) == th, pc);
    strip_pairs<LGWO, Q + 1>(acc, af, bf, aaddr, baddr, pc, th);
  }
}

template <int LGWO>
__global__ __launch_bounds__(256, 1) void conv_wgrad_strip_kernel(const WgradParams p, int nstrips, int strips_per_img, int strips_per_wg,
                                                                   int ntile) {
  using G = StripGeom<LGWO>;
  constexpr int WO = G::WO, W = G::W, R = G::R, RX = G::RX, WP = G::WP, XROW = G::XROW, XBUF = G::XBUF, BUF = G::BUF;
  constexpr int NLX = G::NLX, NL = G::NL, NT = 13;
  constexpr unsigned kOob = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) float strip_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: the tap-half branch is a scalar branch
  const int th = wave >> 1, nh = wave & 1;
  // (tile, split): every channel tile of a pixel split on ONE XCD (blocks b and b + 8 share an XCD), so that the tiles that read
  // the same strips share an L2; needs the split count to be a multiple of 8
  int tile, z;
  {
    const int idx = blockIdx.x;
    if ((p.ksplit & 7) == 0) {
      const int j = idx >> 3;
      tile = j % ntile;
      z = (idx & 7) + 8 * (j / ntile);
    } else {
      tile = idx % ntile;
      z = idx / ntile;
    }
  }
  const int tile_m = tile / p.tiles_n, tile_n = tile - tile_m * p.tiles_n;
  const int ci0 = tile_m * 32, co0 = tile_n * 64;
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.DY), 0, (int)p.dy_bytes, 0x00020000);

  // halo columns x = -1, W, W + 1 of every row of both buffers: zero, never written again
  for (int i = tid; i < 2 * RX * 3 * 32; i += 256) {
    const int e = i & 31, rest = i >> 5, c = rest % 3, row = rest / 3;      // row over (buffer, input row)
    const int bufi = row / RX, rr = row - bufi * RX;
    strip_lds[bufi * BUF + rr * XROW + (c == 0 ? 0 : (W + c)) * 32 + e] = 0.f;
  }

  floatx16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // per-thread pieces of a strip load.  x piece l: item = l * 256 + tid over (input row, column, float4 of the 32-channel slice);
  // a piece past the strip's last one reads out of range (zeros) and lands on a halo cell, which holds zeros anyway
  constexpr int NP = 16, LAG = 3, PPP = (NL + NP - LAG - 1) / (NP - LAG) > 2 ? (NL + NP - LAG - 1) / (NP - LAG) : 2;   // pieces per pair
  static_assert(PPP <= 3 && PPP * (NP - LAG) >= NL, "every piece of the next strip must be requested and written within one strip");
  StripPieces<NL, PPP> pc;
  pc.rsX = rsX; pc.rsY = rsY; pc.nlx = NLX;
  int x_rr[NLX], x_dst[NLX];
  unsigned x_off[NLX];
#pragma unroll
  for (int l = 0; l < NLX; ++l) {
    const int item = l * 256 + tid;
    const int rr = item / (W * 8), rem = item - rr * (W * 8);
    const int xc = rem >> 3, q = rem & 7;
    x_rr[l] = item < G::NX ? rr : 1 << 20;
    x_dst[l] = item < G::NX ? rr * XROW + (xc + 1) * 32 + q * 4 : (tid & 7) * 4;
    x_off[l] = (unsigned)(((rr * W + xc) * p.Ci + ci0 + q * 4) * 4);
  }
  const int y_dst = XBUF + (tid >> 4) * 64 + (tid & 15) * 4;    // dy piece l: pixel l * 16 + tid / 16, float4 tid % 16 of the 64-channel slice
  const unsigned y_off = (unsigned)(((tid >> 4) * p.Co + co0 + (tid & 15) * 4) * 4);

  const int s_begin = z * strips_per_wg, s_end = min(nstrips, s_begin + strips_per_wg);
  // global offsets of strip `sidx`'s pieces (wave-uniform strip coordinates) and their LDS addresses in buffer `buf`
  auto plan = [&](int sidx, int buf) {
    const bool live = sidx < s_end;
    const int b = sidx / strips_per_img, oy0 = (sidx - b * strips_per_img) * R;
    const unsigned xbase = (unsigned)(((b * p.H + 2 * oy0 - 1) * W) * p.Ci * 4);
    const unsigned ybase = (unsigned)(((b * p.Ho + oy0) * WO) * p.Co * 4);
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      if (l < NLX) {
        const int y = 2 * oy0 - 1 + x_rr[l];
        const bool ok = live && (unsigned)y < (unsigned)p.H;
        pc.off[l] = ok ? xbase + x_off[l] : kOob;
        pc.dst[l] = (unsigned)((buf * BUF + x_dst[l]) * 4);
      } else {
        pc.off[l] = live ? ybase + (unsigned)((l - NLX) * 16 * p.Co * 4) + y_off : kOob;
        pc.dst[l] = (unsigned)((buf * BUF + y_dst + (l - NLX) * 16 * 64) * 4);
      }
    }
  };

  // first strip: everything at once
  plan(s_begin, 0);
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    if (l < NLX) strip_gload(pc.pf[l], pc.off[l], rsX);
    else strip_gload(pc.pf[l], pc.off[l], rsY);
  }
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    if (l == 0) strip_lwrite<0>(pc.dst[l], pc.pf[l]);            // vmcnt(0): every piece has arrived
    else asm volatile("ds_write_b128 %0, %1" : : "v"(pc.dst[l]), "v"(pc.pf[l]) : "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();

  // fragment addresses: A = x[ci = lane % 32] of pixel (2j + lane / 32) shifted by the tap; B = dy[co] of the same pixel
  const int kk = lane >> 5, li = lane & 31;
  int a_tap[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int tap = t < 12 ? th * 13 + t : 12;                  // slots 0..11: taps 0..11 / 13..24; slot 12: the shared tap 12
    const int kh = tap / 5, kw = tap - kh * 5;
    a_tap[t] = (kh * WP + kw) * 32 + kk * 64 + li;
  }
  const int b_base = XBUF + kk * 64 + nh * 32 + li;
  // The k-steps run in PAIRS (pixels 4q .. 4q + 3: the two fragments of a tap lie 2 * 2 * 32 floats apart, one ds_read2st64_b32
  // fetches both); what rides between the MFMAs of a pair: strip_steps.

  int buf = 0;
  for (int sidx = s_begin; sidx < s_end; ++sidx, buf ^= 1) {
    // byte addresses of this buffer's fragments (13 + 1 adds per strip; the pair / tap offsets are immediates)
    unsigned aaddr[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) aaddr[t] = (unsigned)((buf * BUF + a_tap[t]) * 4);
    const unsigned baddr = (unsigned)((buf * BUF + b_base) * 4);
    floatx2 af[2][NT], bf[2];
    plan(sidx + 1, buf ^ 1);
    asm volatile("" ::: "memory");
    strip_fetch_first<LGWO, 0>(af, bf, aaddr, baddr);
    strip_pairs<LGWO, 0>(acc, af, bf, aaddr, baddr, pc, th);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // this wave's LDS writes of the next strip (asm: the compiler does not count them)
    __syncthreads();
  }

  // the shared tap: the second half's partial tile goes through LDS (free now) and is added by the first half
  if (th == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) strip_lds[(nh * 16 + r) * 64 + lane] = acc[12][r];
  }
  __syncthreads();
  if (th == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[12][r] += strip_lds[(nh * 16 + r) * 64 + lane];
  }
  // one slab per workgroup: out[z][tap][ci][co]; register r of lane l = D[ci = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][co = l & 31]
  float* out = p.out + (size_t)z * 25 * p.Ci * p.Co;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int tap = t < 12 ? th * 13 + t : 12;
    if (t < 12 || th == 0) {
      float* o = out + (size_t)tap * p.Ci * p.Co + (size_t)ci0 * p.Co + co0 + nh * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[(size_t)((r & 3) + 8 * (r >> 2) + 4 * kk) * p.Co] = acc[t][r];
    }
  }
}

// dw = beta*dw + scale * sum_z slabs[z]; block = 64 float4 columns x 4 slab groups (LDS tree), so few-output layers
// with many slabs still expose enough parallelism
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw, int n,
                                                           int ksplit, float beta, float scale) {
  __shared__ float4 red[4][64];
  const int lane = threadIdx.x & 63, zg = threadIdx.x >> 6;
  const int e = (blockIdx.x * 64 + lane) * 4;             // n is a multiple of 4 on this path, slabs 16-B aligned
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e < n) {
    int z = zg;                                                // this lane group's slabs zg, zg + 4, ...: four loads in flight, added in order
    for (; z + 12 < ksplit; z += 16) {
      float4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const float4*>(slabs + (size_t)(z + 4 * i) * n + e);
#pragma unroll
      for (int i = 0; i < 4; ++i) { s.x += v[i].x; s.y += v[i].y; s.z += v[i].z; s.w += v[i].w; }
    }
    {
      float4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        v[i] = z + 4 * i < ksplit ? *reinterpret_cast<const float4*>(slabs + (size_t)(z + 4 * i) * n + e) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (z + 4 * i < ksplit) { s.x += v[i].x; s.y += v[i].y; s.z += v[i].z; s.w += v[i].w; }
    }
  }
  red[zg][lane] = s;
  __syncthreads();
  if (zg != 0 || e >= n) return;
  const float4 a = red[0][lane], b = red[1][lane], c = red[2][lane], d = red[3][lane];
  float4 o = make_float4(scale * ((a.x + b.x) + (c.x + d.x)), scale * ((a.y + b.y) + (c.y + d.y)),
                         scale * ((a.z + b.z) + (c.z + d.z)), scale * ((a.w + b.w) + (c.w + d.w)));
  if (beta != 0.f) {
    const float4 q = *reinterpret_cast<const float4*>(dw + e);
    o.x += beta * q.x; o.y += beta * q.y; o.z += beta * q.z; o.w += beta * q.w;
  }
  *reinterpret_cast<float4*>(dw + e) = o;
}

// Few outputs, many slabs (the thin layers: 2400 outputs x 768 slabs): 16 float4 columns x 64 slab groups per block,
// LDS tree over the groups (fixed order -> deterministic)
__global__ __launch_bounds__(1024) void wgrad_reduce_tall_kernel(const float* __restrict__ slabs, float* __restrict__ dw, int n,
                                                                 int ksplit, float beta, float scale) {
  __shared__ float4 red[64][16];
  const int c = threadIdx.x & 15, zg = threadIdx.x >> 4;
  const int e = (blockIdx.x * 16 + c) * 4;
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
  if (e < n) {
    int z = zg;
    for (; z + 64 < ksplit; z += 128) {                    // two independent loads in flight
      const float4 v0 = *reinterpret_cast<const float4*>(slabs + (size_t)z * n + e);
      const float4 v1 = *reinterpret_cast<const float4*>(slabs + (size_t)(z + 64) * n + e);
      s0.x += v0.x; s0.y += v0.y; s0.z += v0.z; s0.w += v0.w;
      s1.x += v1.x; s1.y += v1.y; s1.z += v1.z; s1.w += v1.w;
    }
    if (z < ksplit) {
      const float4 v0 = *reinterpret_cast<const float4*>(slabs + (size_t)z * n + e);
      s0.x += v0.x; s0.y += v0.y; s0.z += v0.z; s0.w += v0.w;
    }
  }
  red[zg][c] = make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w);
  __syncthreads();
  for (int h = 32; h > 0; h >>= 1) {
    if (zg < h) {
      const float4 a = red[zg][c], b = red[zg + h][c];
      red[zg][c] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
    __syncthreads();
  }
  if (zg != 0 || e >= n) return;
  const float4 a = red[0][c];
  float4 o = make_float4(scale * a.x, scale * a.y, scale * a.z, scale * a.w);
  if (beta != 0.f) {
    const float4 q = *reinterpret_cast<const float4*>(dw + e);
    o.x += beta * q.x; o.y += beta * q.y; o.z += beta * q.z; o.w += beta * q.w;
  }
  *reinterpret_cast<float4*>(dw + e) = o;
}

__global__ __launch_bounds__(256) void wgrad_reduce_scalar_kernel(const float* __restrict__ slabs, float* __restrict__ dw, int n,
                                                                  int ksplit, float beta, float scale) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  float s = 0.f;
  for (int z = 0; z < ksplit; ++z) s += slabs[(size_t)z * n + e];
  dw[e] = (beta != 0.f ? beta * dw[e] : 0.f) + scale * s;
}

struct WgradPlan {
  int mode;     // 0 direct, 1..: mfma config id
  int ksplit, chunk, tiles_m, tiles_n, bkp, taps_in_grid;
  int always_slab;   // the kernel writes slabs even for one split (beta / scale are applied by the reduce pass)
};

// 5x5 stride-2 layers the strip-resident kernel takes: Wo = 8 / 16 / 32, whole strips of 64 output pixels
inline bool strip_ok(int H, int W, int Ci, int Co, int k, int s) {
  if (k != 5 || s != 2 || (H & 1) || (W & 1) || Ci % 32 || Co % 64) return false;
  const int Wo = W / 2, Ho = H / 2;
  if (Wo != 8 && Wo != 16 && Wo != 32) return false;
  return Ho % (64 / Wo) == 0;
}

WgradPlan plan_wgrad(int B, int H, int W, int Ci, int Co, int k, int s) {
  WgradPlan pl{};
  int Ho, Wo, pt, pp;
  bg::same_pads(H, k, s, &Ho, &pt);
  bg::same_pads(W, k, s, &Wo, &pp);
  const long M = (long)B * Ho * Wo;
  const bool mfma = (Ci % 4 == 0) && (Co % 4 == 0) && Ci >= 16 && Co >= 16;
  const int kk = k * k;
  const bool thin_ci = Ci <= 4 && Co % 4 == 0 && Co >= 16 && kk * Ci <= 128;
  const bool thin_co = Co <= 4 && Ci % 4 == 0 && Ci >= 16 && s == 1 && kk * Co <= 128;
  int bm = 0, bn = 0;
  if (!mfma && !thin_ci && !thin_co) {
    pl.mode = 0;
    const long nblk = bg::cdiv((size_t)k * k * Ci * Co, 256);
    long want = std::max(1L, 2048 / nblk);
    long chunk = std::max(256L, (M + want - 1) / want);
    pl.chunk = (int)chunk;
    pl.ksplit = (int)((M + chunk - 1) / chunk);
    return pl;
  }
  if (strip_ok(H, W, Ci, Co, k, s) && !getenv("BG_WGRAD_NO_STRIP")) {
    pl.mode = 33;                                             // strip-resident kernel, one slab per workgroup
    pl.bkp = 2;
    pl.tiles_m = Ci / 32;
    pl.tiles_n = Co / 64;
    const long nstrips = (long)B * (Ho / (64 / Wo));
    const long ntile = (long)pl.tiles_m * pl.tiles_n;
    // one workgroup per CU (150 KB of LDS): about 256 workgroups in all, every one with the same number of strips where it divides
    long ks = std::max(1L, std::min(nstrips, std::max(1L, 256 / ntile)));
    long spw = (nstrips + ks - 1) / ks;
    ks = (nstrips + spw - 1) / spw;                           // no workgroup without a strip (its slab would never be written)
    pl.ksplit = (int)ks;
    pl.chunk = (int)spw;                                      // strips per workgroup
    pl.taps_in_grid = 0;
    pl.always_slab = 1;
    return pl;
  }
  if (Ci == 16 && Co == 32 && k == 5 && s == 2 && !(H & 1) && !(W & 1) && (Wo == 32 || Wo == 64) && !(Ho & 1) && !getenv("BG_NO_C16")) {
    pl.mode = 32;                                             // row-staged 16-channel kernel, persistent workgroups, one slab each
    pl.bkp = 4;
    const long nstrips = (long)B * (Ho / 2);
    pl.ksplit = (int)std::max<long>(2, std::min<long>(nstrips, 256));
    pl.chunk = (int)nstrips;
    pl.tiles_m = pl.tiles_n = 1;
    pl.taps_in_grid = 0;
    return pl;
  }
  if (thin_ci && k * Ci <= 16 && (k == 5 || k == 3) && (Co == 16 || Co == 32 || Co == 64) && Wo % 4 == 0 && !getenv("BG_WGRAD_NO_TC")) {
    pl.mode = 31;                                             // row-MFMA kernel, one slab per workgroup
    pl.bkp = 4;
    const int rb = s == 2 ? 8 : 16;                          // output rows per block
    const long nblocks = (long)B * bg::cdiv(Ho, rb);
    pl.ksplit = (int)std::max<long>(2, std::min<long>(nblocks, 1024));
    pl.chunk = (int)nblocks;
    pl.tiles_m = pl.tiles_n = 1;
    pl.taps_in_grid = 0;
    return pl;
  }
  if (thin_ci) {
    const int gw = kk * Ci <= 32 ? 32 : (kk * Ci <= 96 ? 96 : 128);
    pl.bkp = 64;
    if (gw == 32) { pl.mode = 10; bm = 32; bn = 64; }         // e.g. MNIST 1 -> 64
    else if (gw == 96) { pl.mode = 11; bm = 96; bn = 32; }    // RGB -> 16/32
    else { pl.mode = 12; bm = 128; bn = 32; }
    pl.tiles_m = 1;
    pl.tiles_n = bg::cdiv(Co, bn);
    pl.taps_in_grid = 0;
  } else if (thin_co && (Ci == 16 || Ci == 32) && (k == 5 || k == 3) && k * Co <= 16 && W % 4 == 0 && !getenv("BG_WGRAD_NO_TC")) {
    // row-MFMA kernel: one slab per workgroup, workgroups loop over blocks of kTcRows image rows
    pl.mode = 30;
    pl.bkp = 4;
    const long nblocks = (long)B * bg::cdiv(H, kTcRows);
    pl.ksplit = (int)std::max<long>(2, std::min<long>(nblocks, 1024));   // >= 2: beta / scale are applied by the slab reduce
    pl.chunk = (int)nblocks;
    pl.tiles_m = pl.tiles_n = 1;
    pl.taps_in_grid = 0;
    return pl;
  } else if (thin_co) {
    const int gw = kk * Co <= 32 ? 32 : (kk * Co <= 96 ? 96 : 128);
    pl.bkp = 64;
    if (gw == 32) { pl.mode = 20; bm = 64; bn = 32; }
    else if (gw == 96) { pl.mode = 21; bm = 32; bn = 96; }    // 16/32 -> RGB
    else { pl.mode = 22; bm = 32; bn = 128; }
    pl.tiles_m = bg::cdiv(Ci, bm);
    pl.tiles_n = 1;
    pl.taps_in_grid = 0;
  } else {
  static const int no_tg = getenv("BG_WGRAD_NO_TG") ? 1 : 0;
  if (k == 5 && Ci <= 64 && !no_tg && M >= 131072) {
    pl.bkp = 32; bm = 32;
    if (Co > 32) { pl.mode = 6; bn = 64; } else { pl.mode = 7; bn = 32; }
    pl.tiles_m = bg::cdiv(Ci, bm);
    pl.tiles_n = bg::cdiv(Co, bn);
    pl.taps_in_grid = 2;                       // grid.y = k (one kernel row of taps per workgroup)
  } else {
  if (Ci > 64 && Co > 64) { pl.mode = 1; bm = 128; bn = 128; pl.bkp = 32; }
  else if (Ci > 32 && Co > 32) { pl.mode = 2; bm = 64; bn = 64; pl.bkp = 32; }
  else if (Ci > 32) { pl.mode = 3; bm = 64; bn = 32; pl.bkp = 64; }
  else if (Co > 32) { pl.mode = 4; bm = 32; bn = 64; pl.bkp = 64; }
  else { pl.mode = 5; bm = 32; bn = 32; pl.bkp = 128; }
  pl.tiles_m = bg::cdiv(Ci, bm);
  pl.tiles_n = bg::cdiv(Co, bn);
  pl.taps_in_grid = 1;
  }
  }
  const long base = (long)pl.tiles_m * pl.tiles_n * (pl.taps_in_grid == 1 ? kk : (pl.taps_in_grid == 2 ? k : 1));
  const long steps = (M + pl.bkp - 1) / pl.bkp;
  long want;
  static const int old_plan = getenv("BG_WGRAD_OLD_PLAN") ? 1 : 0;
  // position-major small maps skip their all-padding chunks: workgroup lengths differ 4x between centre and corner taps, and
  // more, shorter workgroups balance better than the round model predicts (measured: G1 0.33 ms at 800 workgroups, 0.40 at 400)
  static const int pm_pow2_plan = getenv("BG_WGRAD_PM_POW2") ? 1 : 0;
  const bool skipping = Ho * Wo <= 16 && B >= 128 && (pm_pow2_plan ? (B & (B - 1)) == 0 : B % 128 == 0);
  if (old_plan || pl.taps_in_grid != 1 || skipping) {     // tap-grouped kernel: measured slower with the model's single full round (0.35 vs 0.28 ms)
    static const int tgt_env = getenv("BG_WGRAD_TARGET") ? atoi(getenv("BG_WGRAD_TARGET")) : 0;   // tuning aid
    // ~3 workgroups per CU; the tap-grouped kernel (3 resident per CU) measured best at two full rounds (G5: 0.30 -> 0.27 ms)
    const int tgt = tgt_env ? tgt_env : (pl.taps_in_grid == 2 && M >= 200000 ? 1536 : 768);
    want = std::max(1L, (tgt + base - 1) / base);
    // at least 4 K-steps per workgroup; 8 on the tap-skipping small maps, where about half of them are skipped and every
    // split costs a 25-tap slab (critic 128->256 on a 4x4 map at batch 128: 16 splits = 52 MB of slabs, 55 us; 8 splits 47 us)
    want = std::min(want, std::max(1L, steps / (skipping ? 8 : 4)));
    if ((size_t)kk * Ci * Co * sizeof(float) > (8u << 20)) want = std::min(want, 2L);   // big slabs: the reduce pass costs more than idle CUs
  } else {
    // Pixel split from a cost model instead of a fixed workgroup target: the grid runs in ROUNDS of (workgroups resident per
    // CU) x 256, and a grid that spills a little into the next round pays for a whole one (800 workgroups on 512 slots ran
    // at 78 %).  cost(ks) = rounds x (K steps per workgroup + prologue/epilogue) x step time  +  slab reduce.
    const int bm_eff = pl.taps_in_grid == 2 ? k * bm : bm;           // the tap-grouped kernel holds k tap tiles of x per step
    const int lds = 2 * pl.bkp * (bm_eff + bn) * 4;
    const int per_cu = std::max(1, std::min(160 * 1024 / lds, 5));
    const double slots = 256.0 * per_cu;
    const double t_step = 2.0 * bm_eff * bn * pl.bkp * per_cu / 614e9 * 1e6;      // us per K step with the CU fully resident
    const double ovh = 5.0 / t_step;                                                // prologue + tile store, in steps
    const double nout_bytes = (double)kk * Ci * Co * 4.0;
    double best = 1e30;
    want = 1;
    for (long ks = 1; ks <= std::min<long>(64, std::max(1L, steps / 4)); ++ks) {
      const double rounds = std::ceil(base * ks / slots);
      const double per_wg = std::ceil((double)steps / ks);
      const double cost = rounds * (per_wg + ovh) * t_step + (ks > 1 ? 5.0 + (ks + 1) * nout_bytes / 3.5e6 + ks * nout_bytes / 5e6 : 0.0);
      if (cost < best * 0.999) { best = cost; want = ks; }
    }
  }
  long chunk = ((M + want - 1) / want + pl.bkp - 1) / pl.bkp * pl.bkp;
  pl.chunk = (int)chunk;
  pl.ksplit = (int)((M + chunk - 1) / chunk);
  return pl;
}

}  // namespace

extern "C" {

size_t bg_conv2d_bwd_filter_workspace_bytes(int B, int H, int W, int Cin, int Cout, int ksize, int stride) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || ksize <= 0 || stride <= 0) return 0;
  WgradPlan pl = plan_wgrad(B, H, W, Cin, Cout, ksize, stride);
  if (pl.ksplit <= 1 && !pl.always_slab) return 0;
  return (size_t)pl.ksplit * ksize * ksize * Cin * Cout * sizeof(float);
}

int bg_conv2d_bwd_filter(const float* x, const float* dy, float* dw, int B, int H, int W, int Cin, int Cout, int ksize,
                         int stride, float beta, float scale, void* ws_d, size_t ws_bytes, void* stream) {
  BG_REQUIRE(x && dy && dw, BG_ERR_NULL, "bg_conv2d_bwd_filter: null pointer");
  BG_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, BG_ERR_BAD_SHAPE, "bg_conv2d_bwd_filter: B=%d H=%d W=%d Cin=%d Cout=%d", B, H, W, Cin, Cout);
  BG_REQUIRE(ksize >= 1 && (ksize & 1) && ksize * ksize <= bg::kMaxTaps, BG_ERR_UNSUPPORTED, "bg_conv2d_bwd_filter: kernel size %d", ksize);
  BG_REQUIRE(stride == 1 || stride == 2, BG_ERR_UNSUPPORTED, "bg_conv2d_bwd_filter: stride %d", stride);
  BG_REQUIRE(bg::aligned16(x) && bg::aligned16(dy) && bg::aligned16(dw), BG_ERR_BAD_ALIGNMENT, "bg_conv2d_bwd_filter: pointers must be 16-byte aligned");
  bg::UsefulScope useful(bg::conv_useful_flops(B, H, W, Cin, Cout, ksize, stride));
  WgradPlan pl = plan_wgrad(B, H, W, Cin, Cout, ksize, stride);
  const size_t nout = (size_t)ksize * ksize * Cin * Cout;
  const bool slabs = pl.ksplit > 1 || pl.always_slab;
  const size_t need = slabs ? (size_t)pl.ksplit * nout * sizeof(float) : 0;
  BG_REQUIRE(need == 0 || (ws_d && ws_bytes >= need), BG_ERR_WORKSPACE, "bg_conv2d_bwd_filter: workspace %zu bytes < %zu needed", ws_bytes, need);
  WgradParams p;
  memset(&p, 0, sizeof p);
  p.X = x; p.DY = dy;
  p.B = B; p.H = H; p.W = W; p.Ci = Cin; p.Co = Cout; p.k = ksize; p.s = stride;
  bg::same_pads(H, ksize, stride, &p.Ho, &p.pt);
  bg::same_pads(W, ksize, stride, &p.Wo, &p.pl);
  p.M = B * p.Ho * p.Wo;
  BG_REQUIRE((size_t)B * H * W * (size_t)Cin < (1ull << 29) && (size_t)p.M * (size_t)Cout < (1ull << 29), BG_ERR_UNSUPPORTED,
             "bg_conv2d_bwd_filter: tensor exceeds 2 GiB");
  {
    auto magic = [](unsigned d, unsigned* mul, unsigned* sh) {
      unsigned sft = 0;
      while ((1ull << sft) < d) ++sft;
      *mul = (unsigned)((1ull << (31 + sft)) / d + 1);
      *sh = 31 + sft;
    };
    magic((unsigned)(p.Ho * p.Wo), &p.mul_hw, &p.sh_hw);
    magic((unsigned)p.Wo, &p.mul_w, &p.sh_w);
    auto lg2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return l; };
    p.lg_w = lg2(p.Wo); p.lg_h = lg2(p.Ho);
    p.pow2 = ((1 << p.lg_w) == p.Wo && (1 << p.lg_h) == p.Ho) ? 1 : 0;
    p.lg_b = lg2(B);
    static const int no_pm = getenv("BG_NO_POS_MAJOR") ? 1 : 0;
    // position-major: a chunk = BKP images at ONE output position, so the batch only has to be whole chunks (the critic's merged
    // pass runs 3 x 256 = 768 samples); BG_WGRAD_PM_POW2=1 restores the power-of-two rule of rounds 1-4
    static const int pm_pow2 = getenv("BG_WGRAD_PM_POW2") ? 1 : 0;
    const bool b_ok = pm_pow2 ? (1 << p.lg_b) == B : B % 128 == 0;
    if (p.pow2 && !no_pm && b_ok && B >= 128 && p.Ho * p.Wo <= 16) p.pow2 = 2;   // 8x8: the same-channel stride of position-major rows costs more than the skipped chunks save
    p.x_bytes = (unsigned)((size_t)B * H * W * Cin * sizeof(float));
    p.dy_bytes = (unsigned)((size_t)p.M * Cout * sizeof(float));
  }
  p.chunk = pl.chunk; p.ksplit = pl.ksplit; p.tiles_n = pl.tiles_n; p.tiles_m = pl.tiles_m;
  {
    static const int swz = getenv("BG_WGRAD_SWZ") ? atoi(getenv("BG_WGRAD_SWZ")) : 0;   // per-tap kernel: measured slower with the remap
    static const int swz_tg = getenv("BG_WGRAD_SWZ_TG") ? atoi(getenv("BG_WGRAD_SWZ_TG")) : 1;
    p.xcd_swizzle = (pl.mode == 6 || pl.mode == 7) ? swz_tg : swz;
  }
  p.beta = beta; p.scale = scale;
  p.order_n = 0;
  {
    static const int no_sort = getenv("BG_NO_TILE_SORT") ? 1 : 0;
    if (p.pow2 == 2 && pl.taps_in_grid == 1 && pl.mode >= 1 && pl.mode <= 5 && !no_sort && ksize * ksize <= 32) {
      int live[32], idx[32];
      for (int t = 0; t < ksize * ksize; ++t) {
        const int dyk = t / ksize - p.pt, dxk = t % ksize - p.pl;
        // a tap that lies wholly outside a tiny map has H - 1 - dyk < 0: truncating division would give 0 instead of -1
        auto floordiv = [](int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); };
        const int oh_lo = std::max(0, (-dyk + stride - 1) / stride), oh_hi = std::min(p.Ho - 1, floordiv(H - 1 - dyk, stride));
        const int ow_lo = std::max(0, (-dxk + stride - 1) / stride), ow_hi = std::min(p.Wo - 1, floordiv(W - 1 - dxk, stride));
        live[t] = std::max(0, oh_hi - oh_lo + 1) * std::max(0, ow_hi - ow_lo + 1);
        idx[t] = t;
      }
      std::stable_sort(idx, idx + ksize * ksize, [&](int a, int b) { return live[a] > live[b]; });
      for (int t = 0; t < ksize * ksize; ++t) p.tap_order[t] = (unsigned char)idx[t];
      p.order_n = ksize * ksize;
    }
  }
  p.out = slabs ? static_cast<float*>(ws_d) : dw;
  const double flops = 2.0 * p.M * (double)nout;
  // algorithmic bytes: x and dy read once, dw written once (read too when it is accumulated into)
  const double abytes = 4.0 * ((double)B * H * W * Cin + (double)p.M * Cout + (double)nout * (beta != 0.f ? 2 : 1));
  int rc;
  if (pl.mode == 0) {
    bg::Launch L(stream, "conv_wgrad_direct", flops, abytes);
    bg::launch(wgrad_direct_kernel, dim3(bg::cdiv(nout, 256), pl.ksplit), dim3(256), 0, L.s, p);
    rc = L.done("wgrad_direct_kernel");
  } else if (pl.mode == 33) {
    bg::Launch L(stream, "conv_wgrad_mfma", flops, abytes);
    const int lgwo = p.Wo == 32 ? 5 : (p.Wo == 16 ? 4 : 3);
    const int spi = p.Ho / (64 / p.Wo), nstrips = B * spi, ntile = pl.tiles_m * pl.tiles_n;
    BG_LDS_ATTR_ONCE_V(conv_wgrad_strip_kernel<5>, StripGeom<5>::lds_bytes);
    BG_LDS_ATTR_ONCE_V(conv_wgrad_strip_kernel<4>, StripGeom<4>::lds_bytes);
    BG_LDS_ATTR_ONCE_V(conv_wgrad_strip_kernel<3>, StripGeom<3>::lds_bytes);
    const dim3 grid((unsigned)(ntile * pl.ksplit));
    if (lgwo == 5) bg::launch((conv_wgrad_strip_kernel<5>), grid, dim3(256), StripGeom<5>::lds_bytes, L.s, p, nstrips, spi, pl.chunk, ntile);
    else if (lgwo == 4) bg::launch((conv_wgrad_strip_kernel<4>), grid, dim3(256), StripGeom<4>::lds_bytes, L.s, p, nstrips, spi, pl.chunk, ntile);
    else bg::launch((conv_wgrad_strip_kernel<3>), grid, dim3(256), StripGeom<3>::lds_bytes, L.s, p, nstrips, spi, pl.chunk, ntile);
    rc = L.done("conv_wgrad_strip_kernel");
  } else if (pl.mode == 32) {
    bg::Launch L(stream, "conv_wgrad_c16", flops, abytes);
    const int spi = p.Ho / 2, nstrips = B * spi;
    const size_t lds = std::max((size_t)2 * kC16Rows * 2 * (p.Wo + 2) * 16, (size_t)25 * 2 * 4 * 64) * sizeof(float);
    BG_LDS_ATTR_ONCE_V(conv_wgrad_c16_kernel<64>, 150 * 1024);
    BG_LDS_ATTR_ONCE_V(conv_wgrad_c16_kernel<32>, 150 * 1024);
    if (p.Wo == 64) bg::launch((conv_wgrad_c16_kernel<64>), dim3(pl.ksplit), dim3(256), lds, L.s, p, nstrips, spi);
    else bg::launch((conv_wgrad_c16_kernel<32>), dim3(pl.ksplit), dim3(256), lds, L.s, p, nstrips, spi);
    rc = L.done("conv_wgrad_c16_kernel");
  } else if (pl.mode == 31) {
    bg::Launch L(stream, "conv_wgrad_mfma_thin_ci", flops, abytes);
    const int rb = stride == 2 ? 8 : 16;
    const int bpi = (int)bg::cdiv(p.Ho, rb), nblocks = B * bpi;
    const int nt = Cout / 16;
    const size_t lds = std::max((size_t)((rb - 1) * stride + ksize) * (W * Cin + 2 * kTiHalo), (size_t)ksize * nt * 4 * 64) * sizeof(float);
    BG_REQUIRE(lds <= 64 * 1024, BG_ERR_UNSUPPORTED, "bg_conv2d_bwd_filter: thin-Ci row kernel needs %zu bytes of LDS", lds);
#define BG_TI(NTv, Kv) bg::launch((conv_wgrad_thin_ci_kernel<NTv, Kv>), dim3(pl.ksplit), dim3(256), lds, L.s, p, nblocks, bpi, rb)
    if (ksize == 5) { if (nt == 1) BG_TI(1, 5); else if (nt == 2) BG_TI(2, 5); else BG_TI(4, 5); }
    else { if (nt == 1) BG_TI(1, 3); else if (nt == 2) BG_TI(2, 3); else BG_TI(4, 3); }
#undef BG_TI
    rc = L.done("conv_wgrad_thin_ci_kernel");
  } else if (pl.mode == 30) {
    bg::Launch L(stream, "conv_wgrad_mfma_thin_co", flops, abytes);
    const int bpi = (int)bg::cdiv(H, kTcRows), nblocks = B * bpi;
    const int mt = Cin / 16;
    const size_t lds = std::max((size_t)(kTcRows + ksize - 1) * (W * Cout + 2 * kTcHalo), (size_t)4 * ksize * mt * 4 * 64) * sizeof(float);
    BG_REQUIRE(lds <= 64 * 1024, BG_ERR_UNSUPPORTED, "bg_conv2d_bwd_filter: thin-Co row kernel needs %zu bytes of LDS", lds);
    if (ksize == 5 && mt == 2) bg::launch((conv_wgrad_thin_co_kernel<2, 5>), dim3(pl.ksplit), dim3(256), lds, L.s, p, nblocks, bpi);
    else if (ksize == 5) bg::launch((conv_wgrad_thin_co_kernel<1, 5>), dim3(pl.ksplit), dim3(256), lds, L.s, p, nblocks, bpi);
    else if (mt == 2) bg::launch((conv_wgrad_thin_co_kernel<2, 3>), dim3(pl.ksplit), dim3(256), lds, L.s, p, nblocks, bpi);
    else bg::launch((conv_wgrad_thin_co_kernel<1, 3>), dim3(pl.ksplit), dim3(256), lds, L.s, p, nblocks, bpi);
    rc = L.done("conv_wgrad_thin_co_kernel");
  } else {
    dim3 grid(pl.tiles_m * pl.tiles_n, pl.taps_in_grid == 1 ? ksize * ksize : (pl.taps_in_grid == 2 ? ksize : 1), pl.ksplit);
    bg::Launch L(stream, pl.mode >= 20 ? "conv_wgrad_mfma_thin_co" : (pl.mode >= 10 ? "conv_wgrad_mfma_thin_ci" : "conv_wgrad_mfma"), flops, abytes);
    const dim3 grid1((unsigned)(grid.x * grid.y * grid.z));      // v3 / tg kernels decode (tile, tap, split) themselves
    switch (pl.mode) {
#define BG_V3(BMv, BNv, BKv, WMv, WNv, WKv)                                                                               \
  do {                                                                                                                   \
    if (p.pow2 == 2) bg::launch((conv_wgrad_v3_kernel<BMv, BNv, BKv, WMv, WNv, WKv, 2>), grid1, dim3(256), 0, L.s, p);     \
    else if (p.pow2 == 1) bg::launch((conv_wgrad_v3_kernel<BMv, BNv, BKv, WMv, WNv, WKv, 1>), grid1, dim3(256), 0, L.s, p); \
    else bg::launch((conv_wgrad_v3_kernel<BMv, BNv, BKv, WMv, WNv, WKv, 0>), grid1, dim3(256), 0, L.s, p);        \
  } while (0)
      case 1: BG_V3(128, 128, 32, 2, 2, 1); break;
      case 2: BG_V3(64, 64, 32, 2, 2, 1); break;
      case 3: BG_V3(64, 32, 64, 2, 1, 2); break;
      case 4: BG_V3(32, 64, 64, 1, 2, 2); break;
      case 5: BG_V3(32, 32, 128, 1, 1, 4); break;
#undef BG_V3
      case 6: bg::launch((conv_wgrad_tg_kernel<5, 64, 2, 2>), grid1, dim3(256), 0, L.s, p); break;
      case 7: bg::launch((conv_wgrad_tg_kernel<5, 32, 1, 4>), grid1, dim3(256), 0, L.s, p); break;
      case 10: bg::launch((conv_wgrad_kernel<32, 64, 64, 1, 2, 2, 1>), grid, dim3(256), 0, L.s, p); break;
      case 11: bg::launch((conv_wgrad_kernel<96, 32, 64, 1, 1, 4, 1>), grid, dim3(256), 0, L.s, p); break;
      case 12: bg::launch((conv_wgrad_kernel<128, 32, 64, 1, 1, 4, 1>), grid, dim3(256), 0, L.s, p); break;
      case 20: bg::launch((conv_wgrad_kernel<64, 32, 64, 2, 1, 2, 2>), grid, dim3(256), 0, L.s, p); break;
      case 21: bg::launch((conv_wgrad_kernel<32, 96, 64, 1, 1, 4, 2>), grid, dim3(256), 0, L.s, p); break;
      default: bg::launch((conv_wgrad_kernel<32, 128, 64, 1, 1, 4, 2>), grid, dim3(256), 0, L.s, p); break;
    }
    rc = L.done("conv_wgrad_kernel");
  }
  if (rc) return rc;
  if (slabs) {
    bg::Launch L(stream, "conv_wgrad_reduce", 0, (double)(pl.ksplit + 1) * nout * 4);
    if (nout % 4 == 0 && bg::aligned16(ws_d) && pl.ksplit >= 64 && nout <= 65536)
      bg::launch(wgrad_reduce_tall_kernel, dim3(bg::cdiv(nout / 4, 16)), dim3(1024), 0, L.s, static_cast<const float*>(ws_d), dw,
                         (int)nout, pl.ksplit, beta, scale);
    else if (nout % 4 == 0 && bg::aligned16(ws_d))
      bg::launch(wgrad_reduce_kernel, dim3(bg::cdiv(nout / 4, 64)), dim3(256), 0, L.s, static_cast<const float*>(ws_d), dw,
                         (int)nout, pl.ksplit, beta, scale);
    else
      bg::launch(wgrad_reduce_scalar_kernel, dim3(bg::cdiv(nout, 256)), dim3(256), 0, L.s, static_cast<const float*>(ws_d), dw,
                         (int)nout, pl.ksplit, beta, scale);
    rc = L.done("wgrad_reduce_kernel");
  }
  return rc;
}

}  // extern "C"
