// Conv2D forward and data-gradient (== Conv2DTranspose forward) as implicit GEMM on fp32 MFMA.
// Replaces layers.Conv2D / layers.Conv2DTranspose of reference demo_celeba.py:62-119 and the tape
// gradients w.r.t. activations of wgan.py:140,166,244.
//
//   conv_igemm_kernel   MFMA path (v_mfma_f32_32x32x2_f32, exact fp32): C[M,N] = A_gather[M,K] * Wt[N,K]^T,
//                       K = taps x channels, BK channels per step, A/B tiles staged through LDS with a
//                       register-prefetched double buffer; epilogue fuses bias / LeakyReLU / dropout mask /
//                       LeakyReLU-gradient mask / tanh.  MFMA-bound: 2*M*N*K flop.
//   conv_thin_n_kernel  N <= 4 output channels (generator's last conv, MNIST ConvT->1): one thread per
//                       output pixel, weights through the scalar cache.  VALU/HBM-bound.
//   conv_thin_k_kernel  <= 4 input channels (critic's first conv, data-grad of the RGB conv): one thread
//                       per output pixel x 32 output channels.  HBM-bound on the output write.
//   conv_direct_kernel  catch-all for shapes the MFMA tiling cannot take (channel count not a multiple
//                       of 16); used by small test geometries only.
#include "conv_common.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace bg {
}

#if defined(BG_DIAG) && defined(IGEMM_CLOCK)
// diagnostic build (tools/igemm_clock.py): every workgroup leaves the shader ticks (s_memtime) of its K loop, the 100 MHz times
// (s_memrealtime) of its entry, loop start and loop end, and where it ran (HW_ID, XCC_ID).  ticks / loop time x 100 MHz is the
// shader clock the kernel really ran at (cdna_hip_programming.md section 7, in-kernel stamps).
__device__ unsigned long long bg_diag_clock[4 * 8192];
__device__ unsigned bg_diag_hw[2 * 8192];
// raw records of the last launch: out[6 i ..] = shader ticks, entry, loop start, loop end, HW_ID, XCC_ID; clears the table
extern "C" int bg_diag_clock_dump(unsigned long long* out, int n_wg) {
  static unsigned long long h[4 * 8192];
  static unsigned hw[2 * 8192];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(bg_diag_clock), sizeof h) != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(hw, HIP_SYMBOL(bg_diag_hw), sizeof hw) != hipSuccess) return -1;
  int n = 0;
  for (int i = 0; i < 8192 && i < n_wg; ++i) {
    if (!h[4 * i + 3]) continue;
    for (int k = 0; k < 4; ++k) out[6 * n + k] = h[4 * i + k];
    out[6 * n + 4] = hw[2 * i];
    out[6 * n + 5] = hw[2 * i + 1];
    ++n;
  }
  for (int i = 0; i < 4 * 8192; ++i) h[i] = 0;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(bg_diag_clock), h, sizeof h);
  return n;
}
#endif

namespace {

using bg::GatherParams;
using bg::GatherPhase;
using bg::RowAnchor;

typedef float floatx16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------------
// MFMA implicit GEMM
// ------------------------------------------------------------------------------------------------
// STATS: the variant that also leaves BatchNorm statistics of its tile (bg_epilogue.stats).  A template flag, not a run-time
// one: the two running sums per accumulator column cost the 64x64 tile its fourth wave per SIMD (103 -> 119 + 16 registers),
// and the plain variant is the dominant kernel of the step.
template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, bool STATS = false>
#ifndef IGEMM_BK16_WAVES
#define IGEMM_BK16_WAVES 4
#endif
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64, (BM == 64 && BN == 64 && !STATS) ? (BK == 16 ? IGEMM_BK16_WAVES : 4) : 1) void conv_igemm_kernel(const GatherParams p) {
  constexpr int NT = WAVES_M * WAVES_N * 64;     // 4 or 8 waves per workgroup
  static_assert(NT == 256 || NT == 512, "4 or 8 waves per workgroup");
  static_assert(BK == 16 || BK == 32, "BK");
  constexpr int LD = BK + 4;                  // row stride (floats): 16-B aligned, conflict-free b128 reads
  constexpr int TPR = BK / 4;                 // loader threads per row (one float4 each)
  constexpr int RPP = NT / TPR;               // rows per loader pass
  constexpr int AP = (BM + RPP - 1) / RPP;    // loader passes for A
  constexpr int BP = (BN + RPP - 1) / RPP;    // loader passes for B
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int MI = WTM / 32, NI = WTN / 32;
  static_assert(MI >= 1 && NI >= 1 && WTM % 32 == 0 && WTN % 32 == 0, "wave tile");
  static_assert(BM % RPP == 0, "A loader passes must tile BM exactly (branch-free stores)");
  constexpr int BROWS = BP * RPP;             // B rows held in LDS (>= BN) so every loader thread stores unconditionally
  constexpr int STAGE = (BM + BROWS) * LD;

  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];
  __shared__ int rowdst[BM];
  __shared__ int taplist[bg::kMaxTaps];
  __shared__ int phase_steps[bg::kMaxPhases], phase_dd[bg::kMaxPhases];
  // Accumulators TRANSPOSED (the weights are the MFMA's row operand) in every variant but the statistics one: a lane then ends
  // with 4 consecutive output channels of one pixel in 4 consecutive registers and the epilogue moves float4 -- 4 stores per
  // 32 x 32 block instead of 16, and for the epilogues that read per element one float4 + one mask word instead of 4 + 4 loads
  // (round 3: every vector-memory instruction costs the matrix pipe ~58 cycles wherever it is issued, and the LeakyReLU-gradient
  // epilogue cost the critic's 32 -> 64 data gradient +24 %).  Needs N % 4 == 0 and 16-byte aligned pointers (host-checked).
  constexpr bool TR = !STATS;
  __shared__ __attribute__((aligned(16))) float s_epi[2][BN];      // TR: bias and folded-BatchNorm scale of the tile's columns

#if defined(BG_DIAG) && defined(IGEMM_CLOCK)
  const unsigned long long dg_re = __builtin_amdgcn_s_memrealtime();
#endif
  // A workgroup owns one output tile of `pm` sub-pixel phases (same anchors, different tap sets and destination offsets).
  // pm = 2 pairs the 9-tap with the 4-tap phase and the two 6-tap phases.
  const int pm = p.pmerge, ngroups = p.nphase / pm;
  int pgrp = p.grp_order[blockIdx.z % ngroups], split = blockIdx.z / ngroups;
  int n_tile, m_tile;
  const int mt = p.mtiles;
  if (p.order_n > 0) {
    // cost-sorted snake (GatherParams::pp_order): 1-D grid, workgroup w sits on CU slot w % 256 in residency round w / 256
    const int w = blockIdx.x, r = w >> 8;
    // odd rounds walk the CUs backwards but keep the XCD (w % 8): the N tile below is u % nt, so an XCD sees the same weight panel
    // in every round -- with the plain reversal 255 - (w & 255) it saw panels x and 7 - x, 6.6 MB of weights on the 512-channel
    // layers against 4 MB of L2, and re-fetched them ten times over (round 3 counters)
    const int u = ((r & 1) && ((r + 1) << 8) <= (int)gridDim.x) ? (r << 8) + ((31 - ((w & 255) >> 3)) << 3) + (w & 7) : w;
    const int pair = u / p.per_pair;
    int rest = u - pair * p.per_pair;
    const int code = p.pp_order[pair];
    const int nt = (p.N + BN - 1) / BN, bch = p.B / BM;
    pgrp = code >> 6;
    if (p.m_fast) {
      const int msib = rest % bch;
      rest /= bch;
      n_tile = rest % nt;
      split = rest / nt;
      m_tile = (code & 63) * bch + msib;
    } else {
      n_tile = rest % nt;           // N tile fastest: with 8 tiles a weight panel stays on one XCD (w % 8) in every round
      rest /= nt;
      split = rest / bch;
      m_tile = (code & 63) * bch + (rest - split * bch);
    }
  } else {
    // logical tile list is n-major (all M tiles of one weight panel, then the next panel): with the XCD remap each
    // XCD's L2 holds only its share of the weight panels while the activations stream through
    const int L = p.xcd_swizzle ? bg::xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    n_tile = L / mt;
    m_tile = L - n_tile * mt;
  }
  auto phase_of = [&](int q) { return pm == 2 ? (q == 0 ? pgrp : p.nphase - 1 - pgrp) : pgrp * pm + q; };
  const GatherPhase& g = p.ph[phase_of(0)];
  const int Mph = p.B * g.Ha * g.Wa;
  const int m0 = m_tile * BM;
  const int n0 = n_tile * BN;
  const int tid = threadIdx.x;
  // BatchNorm statistics of the stored tile (bg_epilogue.stats): one partial row per workgroup, [row][2][N]
  constexpr bool do_stats = STATS;
  float* stats_row = do_stats ? p.stats + ((size_t)pgrp * mt + m_tile) * 2 * p.N : nullptr;
  if (m0 >= Mph) {
    if (do_stats && tid < BN && n0 + tid < p.N) { stats_row[n0 + tid] = 0.f; stats_row[p.N + n0 + tid] = 0.f; }
    return;
  }
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // ---- loader bookkeeping.  Operands come through buffer descriptors: 32-bit byte offsets (cheap address
  // maths) and the hardware range check returns 0 for the offsets we poison (zero padding, rows past M,
  // channels past N) -- no clamps, no selects, no branches in the loop body.
  constexpr unsigned kOob = 0x80000000u;                  // >= num_records: tensors are < 2 GiB (checked on the host)
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Wt), 0, (int)p.w_bytes, 0x00020000);
  const int lrow = tid / TPR, lq = tid % TPR;
  int a_y[AP], a_x[AP];
  unsigned a_off[AP], b_off[BP];
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    RowAnchor ra;
    int dst;
    bg::decode_row(p, g, m0 + lrow + i * RPP, Mph, ra, dst);
    a_y[i] = ra.ay;
    a_x[i] = ra.ax;
    a_off[i] = (unsigned)(((ra.b * p.Hs + ra.ay) * p.Ws + ra.ax) * p.Ck + lq * 4) * 4u;   // garbage for invalid rows: never used
  }
#pragma unroll
  for (int i = 0; i < BP; ++i) {
    const int n = n0 + lrow + i * RPP;
    b_off[i] = n < p.N ? (unsigned)(n * p.Ck + lq * 4) * 4u : kOob;
  }
  if (tid < BM) {
    RowAnchor tmp;
    int dst;
    bg::decode_row(p, g, m0 + tid, Mph, tmp, dst);
    rowdst[tid] = dst;
  }
  if (TR && tid < BN) {
    const int n = n0 + tid;
    s_epi[0][tid] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
    s_epi[1][tid] = (p.epi_mode == BG_EPI_AFFINE_LRELU && n < p.N) ? p.ref[n] : 1.f;
  }
  // Taps whose source pixel is zero padding for EVERY row of the tile are dropped from the K loop (no loads, no MFMAs).
  // Only position-major tiles can lose taps (all their rows sit at ONE output position; a pixel-major tile spans whole image
  // rows, where every tap is in range for some pixel), and there the mask follows from any one row: no exchange needed.
  const int kchunks = p.Ck / BK;
  int tbase = 0;
  for (int q = 0; q < pm; ++q) {
    const GatherPhase& gq = p.ph[phase_of(q)];
    unsigned tmask = gq.ntaps >= 32 ? 0xffffffffu : ((1u << gq.ntaps) - 1u);
    if (p.pos_major) {
      unsigned bits = 0u;
      for (int t = 0; t < gq.ntaps; ++t) {
        const int dy = bg::tap_dy(gq.tap[t]), dx = bg::tap_dx(gq.tap[t]);
        bits |= ((unsigned)(a_y[0] + dy) < (unsigned)p.Hs && (unsigned)(a_x[0] + dx) < (unsigned)p.Ws) ? (1u << t) : 0u;
      }
      tmask = (unsigned)__builtin_amdgcn_readfirstlane((int)bits);
    }
    if (tid < gq.ntaps && ((tmask >> tid) & 1u)) taplist[tbase + __popc(tmask & ((1u << tid) - 1u))] = gq.tap[tid];
    if (tid == 0) {
      phase_steps[q] = __popc(tmask) * kchunks;
      phase_dd[q] = (gq.py - g.py) * p.Wd + (gq.px - g.px);
    }
    tbase += __popc(tmask);
  }
  const int ntaps_c = tbase;
  __syncthreads();

  const int nsteps_all = ntaps_c * kchunks;
  const int s_begin = (int)((long)nsteps_all * split / p.ksplit);          // this workgroup's slice of the K steps
  const int nsteps = (int)((long)nsteps_all * (split + 1) / p.ksplit) - s_begin;
  float4 regA0[AP], regB0[BP], regA1[AP], regB1[BP];     // two staging register sets (global loads run 2 steps ahead of their LDS write)

  // Software pipeline, one barrier per step, loop body branch-free (the scheduler slots the loader between MFMAs):
  //   step s:  MFMA(k-octets 0..KO/2-1 of tile s) | ds_write tile s+1 (loaded during step s-2) | issue loads of tile s+3
  //            | barrier | MFMA(remaining octets of tile s), with the first fragments of tile s+1 fetched underneath
  // The barrier sits in the MIDDLE of the step: every fragment read of tile s has been issued before it (fragments run one
  // octet ahead of their MFMAs), so after it tile s+1 is complete in LDS and its first fragments are read while the last
  // MFMAs of tile s execute -- no LDS round trip is exposed after the barrier.
  // Loads have two full steps (~2 x 1024 MFMA cycles) to return: enough for an L2 miss served by the Infinity Cache /
  // HBM, which the weight-streaming small-M layers (K = 6400) hit on most steps.
  // ---- VALU-lean loader (round 5).  On gfx950 the fp32 MFMA runs on the SIMD's vector datapath: tools/probes/mfma_valu_dual.hip --
  // two v_add_u32 (or one v_cmp + v_cndmask pair, or two v_pk_fma_f32) issued per v_mfma_f32_32x32x2_f32 take the matrix rate from
  // 152 to 134 TFLOP/s, four take it to 118: every vector instruction in the K loop is matrix time lost one for one, whichever
  // wave issues it.  The loader used to spend ~20 of them per step (per row: two adds and two compares for the SAME-padding test, an
  // add and a select for the offset; per weight row an add and a select).  Now
  //  * the padding test of a row is a BIT MASK over the unit's compact tap list, taken once in the prologue;
  //  * a row's offset is rebuilt only when the TAP changes (every Ck / BK steps): bit, add, shift-or = 3 instructions, bit 31 set
  //    = poisoned (the buffer range check returns zeros);
  //  * the channel chunk of a step and the whole weight offset travel in the buffer instruction's SCALAR offset (non-negative by
  //    construction), so the weight rows need no vector instruction at all and the activation rows none between tap changes;
  //  * steps past the unit's end (the pipeline's look-ahead, the padded odd step) read through a descriptor of zero records.
  // Same addresses, same zeros: results are bit-identical.
  int g_tq = s_begin / kchunks;                          // compact tap index / channel chunk of the NEXT gload (wave-uniform)
  int g_kc = s_begin - g_tq * kchunks;
  int g_left = nsteps;                                   // steps still to load; <= 0: padded step, the null descriptor loads zeros
  unsigned a_bad[AP];                                    // bit t: compact tap t falls on zero padding for this row (or the row is past M)
#pragma unroll
  for (int i = 0; i < AP; ++i) a_bad[i] = 0u;
  for (int t = 0; t < ntaps_c; ++t) {
    const int tp = taplist[t];
    const int dy = bg::tap_dy(tp), dx = bg::tap_dx(tp);
#pragma unroll
    for (int i = 0; i < AP; ++i)
      a_bad[i] |= ((unsigned)(a_y[i] + dy) < (unsigned)p.Hs && (unsigned)(a_x[i] + dx) < (unsigned)p.Ws) ? 0u : (1u << t);
  }
  const __amdgpu_buffer_rsrc_t rsNull = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, 0, 0x00020000);
  unsigned a_voff[AP];                                   // byte offset of the row's quad at the current tap (bit 31: poisoned)
  unsigned g_woff = 0;                                   // weight offset of the current tap, without the channel chunk (scalar)
  int g_tp_v = taplist[min(g_tq, ntaps_c - 1)];          // the NEXT tap's packed entry, read one tap ahead of its use (no LDS wait then)
  auto new_tap = [&]() {                                 // wave-uniform: runs when the load stream moves to another tap
    const int tq = min(g_tq, ntaps_c - 1);
    const int tp = __builtin_amdgcn_readfirstlane(g_tp_v);
    g_tp_v = taplist[min(g_tq + 1, ntaps_c - 1)];
    const int dy = bg::tap_dy(tp), dx = bg::tap_dx(tp);
    const unsigned tapoff = (unsigned)(((dy * p.Ws + dx) * p.Ck) * 4);
    g_woff = (unsigned)((bg::tap_wi(tp) * p.N * p.Ck) * 4);
#pragma unroll
    for (int i = 0; i < AP; ++i) a_voff[i] = (((a_bad[i] >> tq) & 1u) << 31) | (a_off[i] + tapoff);
  };
  new_tap();
  auto gload = [&](float4 (&rA)[AP], float4 (&rB)[BP]) {
    const bool live = g_left > 0;
    const int c0b = g_kc * BK * 4;                       // this step's channel chunk, in bytes: the scalar offset
    const __amdgpu_buffer_rsrc_t ra = live ? rsA : rsNull, rb = live ? rsB : rsNull;
#pragma unroll
    for (int i = 0; i < AP; ++i) {
#if defined(BG_DIAG) && defined(IGEMM_NO_A)
      rA[i] = make_float4((live && !(a_voff[i] >> 31)) ? 1.f : 0.f, 0.5f, (float)c0b, 0.25f);      // knock-out: no gather traffic, same data flow
#else
      rA[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ra, a_voff[i], c0b, 0));
#endif
    }
#pragma unroll
    for (int i = 0; i < BP; ++i)
#if defined(BG_DIAG) && defined(IGEMM_NO_B)
      rB[i] = make_float4(live ? 1.f : 0.f, 0.5f, (float)g_woff, 0.25f);
#else
      rB[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rb, b_off[i], (int)g_woff + c0b, 0));
#endif
    --g_left;
    if (++g_kc == kchunks) {
      g_kc = 0;
      ++g_tq;
      new_tap();
    }
  };
  auto lstore = [&](int buf, const float4 (&rA)[AP], const float4 (&rB)[BP]) {
    float* sa = smem + buf * STAGE;
    float* sb = sa + BM * LD;
#if defined(BG_DIAG) && defined(IGEMM_NO_LSTORE)
    // knock-out: the staged tiles are consumed (the loads stay live and waited for) but never written to LDS
#pragma unroll
    for (int i = 0; i < AP; ++i) asm volatile("" ::"v"(rA[i].x), "v"(rA[i].y), "v"(rA[i].z), "v"(rA[i].w));
#pragma unroll
    for (int i = 0; i < BP; ++i) asm volatile("" ::"v"(rB[i].x), "v"(rB[i].y), "v"(rB[i].z), "v"(rB[i].w));
    (void)sa; (void)sb;
#else
#pragma unroll
    for (int i = 0; i < AP; ++i) *reinterpret_cast<float4*>(sa + (lrow + i * RPP) * LD + lq * 4) = rA[i];
#pragma unroll
    for (int i = 0; i < BP; ++i) *reinterpret_cast<float4*>(sb + (lrow + i * RPP) * LD + lq * 4) = rB[i];
#endif
  };

  floatx16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  gload(regA0, regB0);
  gload(regA1, regB1);
  // per-column epilogue operands are fetched now, under the first tile loads, not after the last MFMA
  float e_bias[NI], e_mul[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int n = n0 + wn * WTN + j * 32 + (lane & 31);
    e_bias[j] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
    e_mul[j] = (p.epi_mode == BG_EPI_AFFINE_LRELU && n < p.N) ? p.ref[n] : 1.f;
  }
  lstore(0, regA0, regB0);
  gload(regA0, regB0);
  __syncthreads();

  const int frow = lane & 31, fk = (lane >> 5) * 4;
  const float* sa0 = smem + (wm * WTM + frow) * LD + fk;
  const float* sb0 = smem + BM * LD + (wn * WTN + frow) * LD + fk;
  constexpr int KO = BK / 8;                  // k-octets per step (2 or 4: even, so the fragment slots line up across steps)
  // fragments: 4 k's per b128 read, fetched one k-octet ahead of the MFMAs that consume them, across step boundaries
  float4 af[2][MI], bf[2][NI];
  auto fetch = [&](int slot, int buf, int ko) {
#if defined(BG_DIAG) && defined(IGEMM_NO_FETCH)
    if (ko >= 0) {                           // knock-out: no fragment reads in the loop (registers keep their first contents)
      asm volatile("" : "+v"(af[slot][0].x), "+v"(bf[slot][0].x));
      return;
    }
#endif
    const float* sa = sa0 + buf * STAGE + ko * 8;
    const float* sb = sb0 + buf * STAGE + ko * 8;
#pragma unroll
    for (int i = 0; i < MI; ++i) af[slot][i] = *reinterpret_cast<const float4*>(sa + i * 32 * LD);
#pragma unroll
    for (int j = 0; j < NI; ++j) bf[slot][j] = *reinterpret_cast<const float4*>(sb + j * 32 * LD);
  };
#if defined(BG_DIAG) && defined(IGEMM_NO_FETCH)
#pragma unroll
  for (int sl = 0; sl < 2; ++sl) {
#pragma unroll
    for (int i = 0; i < MI; ++i) af[sl][i] = make_float4(1.f, 0.5f, 0.25f, 2.f);
#pragma unroll
    for (int j = 0; j < NI; ++j) bf[sl][j] = make_float4(0.5f, 1.f, 2.f, 0.25f);
  }
#endif
  fetch(0, 0, 0);
  // one pipeline step on LDS buffer `cur`; `mid` (ds_write of the next tile, loads of a later one) runs before the barrier
  auto step_body = [&](int cur, auto mid) {
#pragma unroll
    for (int ko = 0; ko < KO; ++ko) {
      const int c = ko & 1, n = c ^ 1;
      if (ko + 1 < KO) fetch(n, cur, ko + 1);
      if (ko == KO / 2) {
        mid();
#if !(defined(BG_DIAG) && defined(IGEMM_NO_BARRIER))
        __syncthreads();                      // tile `cur^1` complete; every read of tile `cur` was issued (and drained) before it
#endif
      }
      if (ko + 1 == KO) fetch(n, cur ^ 1, 0);
      __builtin_amdgcn_sched_barrier(0);      // keep the prefetch ahead of the MFMAs (see conv_wgrad.hip)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          if (TR) {                                           // C^T = W^T A^T: same products, same order of the K sum
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[c][j].x, af[c][i].x, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[c][j].y, af[c][i].y, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[c][j].z, af[c][i].z, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[c][j].w, af[c][i].w, acc[i][j], 0, 0, 0);
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][i].x, bf[c][j].x, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][i].y, bf[c][j].y, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][i].z, bf[c][j].z, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][i].w, bf[c][j].w, acc[i][j], 0, 0, 0);
          }
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // ---- epilogue of one phase: acc reg r of lane l holds C[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]; the accumulators
  // restart from zero for the next phase of the group.  It runs between two K steps: the loads of the next phase's tiles are
  // already in flight and its first fragments already in registers, so only the stores themselves sit between the MFMAs.
  const int col = lane & 31, rhalf = (lane >> 5) * 4;
  float st_sum[NI], st_sq[NI];                                // column sums of this lane's accumulator columns (do_stats)
#pragma unroll
  for (int j = 0; j < NI; ++j) { st_sum[j] = 0.f; st_sq[j] = 0.f; }
  // TR: acc reg r of lane l holds C[row = l & 31][col = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)] of the 32 x 32 block
  auto epilogue_tr = [&](int q) {
    const int dd = phase_dd[q];
    const size_t slab_off = (size_t)split * ((size_t)p.B * p.Hd * p.Wd * p.N);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int dst = rowdst[wm * WTM + i * 32 + col];
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int nl = wn * WTN + j * 32 + 8 * g4 + rhalf;     // column within the tile: rhalf = 4 * (lane >> 5)
          const float4 v = make_float4(acc[i][j][4 * g4], acc[i][j][4 * g4 + 1], acc[i][j][4 * g4 + 2], acc[i][j][4 * g4 + 3]);
          if (dst >= 0 && n0 + nl < p.N) {
            const size_t idx = (size_t)(dst + dd) * p.N + n0 + nl;
            if (p.ksplit > 1) {
              *reinterpret_cast<float4*>(p.slab + slab_off + idx) = v;
            } else {
              *reinterpret_cast<float4*>(p.C + idx) = bg::apply_epilogue4(p, v, idx, &s_epi[0][nl], &s_epi[1][nl]);
            }
          }
          acc[i][j][4 * g4] = 0.f; acc[i][j][4 * g4 + 1] = 0.f; acc[i][j][4 * g4 + 2] = 0.f; acc[i][j][4 * g4 + 3] = 0.f;
        }
    }
  };
  auto epilogue = [&](int q) {
#if defined(BG_DIAG) && defined(IGEMM_NO_EPI)
    {                                          // knock-out: nothing stored, nothing read; the accumulators stay live and restart
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          asm volatile("" : "+v"(acc[i][j]));
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
      return;
    }
#endif
    if (TR) {
      epilogue_tr(q);
      return;
    }
    const int dd = phase_dd[q];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int n = n0 + wn * WTN + j * 32 + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + rhalf;
          const int dst = rowdst[row];
          if (dst >= 0 && n < p.N) {
            const size_t idx = (size_t)(dst + dd) * p.N + n;
            if (p.ksplit > 1) p.slab[(size_t)split * ((size_t)p.B * p.Hd * p.Wd * p.N) + idx] = acc[i][j][r];
            else p.C[idx] = bg::apply_epilogue_pre(p, acc[i][j][r], idx, e_bias[j], e_mul[j]);
          }
          if (do_stats) {                                      // rows past M and columns past N hold exact zeros
            st_sum[j] += acc[i][j][r];
            st_sq[j] = fmaf(acc[i][j][r], acc[i][j][r], st_sq[j]);
          }
          acc[i][j][r] = 0.f;
        }
      }
  };
  // steps left in the phase being accumulated (pm == 1: the workgroup's whole K slice); phases without a live tap store zeros
  int c_q = 0, c_left = pm == 1 ? nsteps : __builtin_amdgcn_readfirstlane(phase_steps[0]);       // scalar: a compare in the K loop, not a vector one
  auto phase_done = [&]() {
    do {
      epilogue(c_q);
      ++c_q;
      c_left = c_q < pm ? __builtin_amdgcn_readfirstlane(phase_steps[c_q < pm ? c_q : 0]) : -1;
    } while (c_left == 0);
  };
  if (c_left == 0) phase_done();
#if defined(BG_DIAG) && defined(IGEMM_CLOCK)
  const unsigned long long dg_t0 = __builtin_amdgcn_s_memtime(), dg_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  const int nsteps2 = (nsteps + 1) & ~1;                       // steps come in (even, odd) pairs; a padded step adds zeros
#if defined(IGEMM_PRIO_ROT) || defined(IGEMM_PRIO_INV)
  // experiment (profiles/r05_a_gather_gemm_limits.md): the four workgroups of a CU do not share its matrix pipes evenly -- they end
  // one after the other, and the CU's last quarter runs with one to three of them.  Wave priority by residency slot, fixed
  // (youngest first) or rotating every eight steps.
  const int prio_slot = (int)((((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) >> 8) & 3u);
#endif
#if defined(IGEMM_PRIO_INV)
  switch (prio_slot) { case 0: __builtin_amdgcn_s_setprio(0); break; case 1: __builtin_amdgcn_s_setprio(1); break; case 2: __builtin_amdgcn_s_setprio(2); break; default: __builtin_amdgcn_s_setprio(3); break; }
#endif
  for (int step = 0; step < nsteps2; step += 2) {
#if defined(IGEMM_PRIO_ROT)
    if ((step & 7) == 0) {
      switch (((step >> 3) + prio_slot) & 3) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
      }
    }
#endif
    step_body(0, [&]() { lstore(1, regA1, regB1); gload(regA1, regB1); });
    if (--c_left == 0) phase_done();
    step_body(1, [&]() { lstore(0, regA0, regB0); gload(regA0, regB0); });
    if (--c_left == 0) phase_done();
  }
#if defined(BG_DIAG) && defined(IGEMM_CLOCK)
  {
    const unsigned long long dg_t1 = __builtin_amdgcn_s_memtime(), dg_r1 = __builtin_amdgcn_s_memrealtime();
    const unsigned w = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (tid == 0 && w < 8192) {
      bg_diag_clock[4 * w] = dg_t1 - dg_t0; bg_diag_clock[4 * w + 1] = dg_re; bg_diag_clock[4 * w + 2] = dg_r0; bg_diag_clock[4 * w + 3] = dg_r1;
      bg_diag_hw[2 * w] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
      bg_diag_hw[2 * w + 1] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));
    }
  }
#endif
  if (do_stats) {
    // lane halves hold different rows of the same column; waves along M share columns: fixed-order sums through LDS
    __syncthreads();
    float* red = smem;                                         // [WAVES_M][BN][2]
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const float a = st_sum[j] + __shfl_xor(st_sum[j], 32, 64), b = st_sq[j] + __shfl_xor(st_sq[j], 32, 64);
      if (lane < 32) {
        red[(wm * BN + wn * WTN + j * 32 + col) * 2 + 0] = a;
        red[(wm * BN + wn * WTN + j * 32 + col) * 2 + 1] = b;
      }
    }
    __syncthreads();
    if (tid < BN && n0 + tid < p.N) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES_M; ++w) { a += red[(w * BN + tid) * 2]; b += red[(w * BN + tid) * 2 + 1]; }
      stats_row[n0 + tid] = a;
      stats_row[p.N + n0 + tid] = b;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// thin-N (N <= 4 output channels): one thread per output pixel of an 8 x 32 anchor tile.  The input patch
// (tile + tap halo, CK channels) is staged once through LDS with coalesced float4 loads and re-used by
// every tap; weights come through the scalar cache (wave-uniform addresses).  Pixel stride in LDS is
// CK+4 floats so the per-lane ds_read_b128 of neighbouring pixels is bank-conflict free.
// ------------------------------------------------------------------------------------------------
// TW = 32 (8 rows) or 16 (16 rows): phases no wider than 16 anchors (MNIST's ConvT 64 -> 1 on 14 x 14 anchors) take the square
// tile -- one workgroup per (image, phase) with 77 % of its threads on real pixels instead of two half-empty 8 x 32 tiles
// (38 %), i.e. one round of workgroups instead of two (round 3: 90 -> see DESIGN.md section 7).
template <int CK, int kThinTW>
__global__ __launch_bounds__(256) void conv_thin_n_patch_kernel(const GatherParams p, int tiles_x, int tiles_y) {
  extern __shared__ __attribute__((aligned(16))) float patch[];
  constexpr int kThinTH = 256 / kThinTW;
  constexpr int CS = CK + 4, Q = CK / 4;
  const int phase = blockIdx.z % p.nphase, b = blockIdx.z / p.nphase;
  const GatherPhase& g = p.ph[phase];
  const int a0 = blockIdx.y * kThinTH, x0 = blockIdx.x * kThinTW;
  if (a0 >= g.Ha || x0 >= g.Wa) return;
  // halo of this phase's tap set
  int dmin_y = 127, dmax_y = -127, dmin_x = 127, dmax_x = -127;
  for (int t = 0; t < g.ntaps; ++t) {
    const int dy = bg::tap_dy(g.tap[t]), dx = bg::tap_dx(g.tap[t]);
    dmin_y = min(dmin_y, dy); dmax_y = max(dmax_y, dy);
    dmin_x = min(dmin_x, dx); dmax_x = max(dmax_x, dx);
  }
  const int PH = (kThinTH - 1) * p.ss + (dmax_y - dmin_y) + 1;
  const int PW = (kThinTW - 1) * p.ss + (dmax_x - dmin_x) + 1;
  const int sy0 = a0 * p.ss + dmin_y, sx0 = x0 * p.ss + dmin_x;
  const float* src = p.A + (size_t)b * p.Hs * p.Ws * CK;
  for (int idx = threadIdx.x; idx < PH * PW * Q; idx += 256) {
    const int q = idx % Q, pix = idx / Q;
    const int py = pix / PW, px = pix - py * PW;
    const int sy = sy0 + py, sx = sx0 + px;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws)
      v = *reinterpret_cast<const float4*>(src + ((size_t)sy * p.Ws + sx) * CK + q * 4);
    *reinterpret_cast<float4*>(patch + pix * CS + q * 4) = v;
  }
  __syncthreads();
  const int ty = threadIdx.x / kThinTW, tx = threadIdx.x % kThinTW;
  const int N = p.N;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const float* base = patch + ((ty * p.ss - dmin_y) * PW + tx * p.ss - dmin_x) * CS;
  for (int t = 0; t < g.ntaps; ++t) {
    const int tp = g.tap[t];
    const float* a = base + (bg::tap_dy(tp) * PW + bg::tap_dx(tp)) * CS;
    const float* w = p.Wt + (size_t)bg::tap_wi(tp) * N * CK;     // wave-uniform -> scalar loads
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const float4 av = *reinterpret_cast<const float4*>(a + q * 4);
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        if (n < N) {
          const float4 wv = *reinterpret_cast<const float4*>(w + n * CK + q * 4);
          acc[n] = fmaf(av.x, wv.x, acc[n]);
          acc[n] = fmaf(av.y, wv.y, acc[n]);
          acc[n] = fmaf(av.z, wv.z, acc[n]);
          acc[n] = fmaf(av.w, wv.w, acc[n]);
        }
      }
    }
  }
  const int a_ = a0 + ty, bx = x0 + tx;
  if (a_ < g.Ha && bx < g.Wa) {
    const size_t dst = ((size_t)b * p.Hd + a_ * p.ds + g.py) * p.Wd + bx * p.ds + g.px;
#pragma unroll
    for (int n = 0; n < 4; ++n)
      if (n < N) {
        const size_t idx = dst * N + n;
        p.C[idx] = bg::apply_epilogue(p, acc[n], idx, n);
      }
  }
}

// The same for ALL sub-pixel phases of a transposed conv from ONE staged patch (MNIST's ConvT 64 -> 1 and the data gradient of its
// Conv 1 -> 64: 14 x 14 anchors, four phases).  The per-phase form above stages the same 14 x 14 x 64 source four times, once per
// (image, phase) workgroup, each with an 18 x 18 pixel patch of 88 KB -- one workgroup per CU, four rounds at batch 256, 81 us for
// 13.6 MB of traffic -- and reads its weights with wave-uniform scalar loads INSIDE the channel loop, each waited for before its
// four FMAs.  Here a workgroup owns 64 anchors of one image (4 x 16 or 2 x 32) and wave w computes phase w of them: four times the
// workgroups (256 at MNIST's batch of 64: every CU busy), a quarter of the taps per thread, no divergence inside a wave.  The
// patch carries the union of the phases' halos, clipped to the anchors the map has (6 x 18 pixels, 29 KB); it and the weights of
// every tap go to LDS in ONE batch of up to eight loads per thread (a load -> LDS write loop is a chain of global round trips).
// Same taps in the same order per output: bit-identical to the per-phase kernel.
template <int CK, int kThinTW, int N>
__global__ __launch_bounds__(256) void conv_thin_n_patch_all_kernel(const GatherParams p, int ha_max, int wa_max, int ntap_w) {
  extern __shared__ __attribute__((aligned(16))) float patch[];
  constexpr int kThinTH = 64 / kThinTW;
  constexpr int CS = CK + 4, Q = CK / 4;
  const int b = blockIdx.z;
  const int a0 = blockIdx.y * kThinTH, x0 = blockIdx.x * kThinTW;
  const int th = min(kThinTH, ha_max - a0), tw = min(kThinTW, wa_max - x0);      // anchors this tile really has
  int dmin_y = 127, dmax_y = -127, dmin_x = 127, dmax_x = -127;                 // union halo of all phases
  for (int q = 0; q < p.nphase; ++q) {
    const GatherPhase& g = p.ph[q];
    for (int t = 0; t < g.ntaps; ++t) {
      const int dy = bg::tap_dy(g.tap[t]), dx = bg::tap_dx(g.tap[t]);
      dmin_y = min(dmin_y, dy); dmax_y = max(dmax_y, dy);
      dmin_x = min(dmin_x, dx); dmax_x = max(dmax_x, dx);
    }
  }
  const int PH = (th - 1) * p.ss + (dmax_y - dmin_y) + 1;
  const int PW = (tw - 1) * p.ss + (dmax_x - dmin_x) + 1;
  const int sy0 = a0 * p.ss + dmin_y, sx0 = x0 * p.ss + dmin_x;
  const float* src = p.A + (size_t)b * p.Hs * p.Ws * CK;
  float* wl = patch + PH * PW * CS;
  constexpr int SU = 8;
  const int npatch = PH * PW * Q, total = npatch + ntap_w * N * Q;               // float4 items: the patch, then the weights
  for (int i0 = threadIdx.x; i0 < total; i0 += 256 * SU) {
    float4 v[SU];
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int idx = i0 + u * 256;
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < npatch) {
        const int q = idx % Q, pix = idx / Q;
        const int py = pix / PW, px = pix - py * PW;
        const int sy = sy0 + py, sx = sx0 + px;
        if ((unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws)
          v[u] = *reinterpret_cast<const float4*>(src + ((size_t)sy * p.Ws + sx) * CK + q * 4);
      } else if (idx < total) {
        v[u] = *reinterpret_cast<const float4*>(p.Wt + (size_t)(idx - npatch) * 4);
      }
    }
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int idx = i0 + u * 256;
      if (idx < npatch) *reinterpret_cast<float4*>(patch + (idx / Q) * CS + (idx % Q) * 4) = v[u];
      else if (idx < total) *reinterpret_cast<float4*>(wl + (idx - npatch) * 4) = v[u];
    }
  }
  __syncthreads();
  const int ph = threadIdx.x >> 6, l = threadIdx.x & 63;                          // wave = phase
  const int ty = l / kThinTW, tx = l % kThinTW;
  if (ph >= p.nphase || ty >= th || tx >= tw) return;
  const GatherPhase& g = p.ph[ph];
  const int a_ = a0 + ty, bx = x0 + tx;
  if (a_ >= g.Ha || bx >= g.Wa) return;
  const float* base = patch + ((ty * p.ss - dmin_y) * PW + tx * p.ss - dmin_x) * CS;
  float acc[N];
#pragma unroll
  for (int n = 0; n < N; ++n) acc[n] = 0.f;
  for (int t = 0; t < g.ntaps; ++t) {
    const int tp = g.tap[t];
    const float* a = base + (bg::tap_dy(tp) * PW + bg::tap_dx(tp)) * CS;
    const float* w = wl + bg::tap_wi(tp) * N * CK;                  // same address in every lane: LDS broadcast
    float4 av[Q];                                                   // the pixel's channels at this tap, all reads in flight at once
#pragma unroll
    for (int q = 0; q < Q; ++q) av[q] = *reinterpret_cast<const float4*>(a + q * 4);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
#pragma unroll
      for (int n = 0; n < N; ++n) {
        const float4 wv = *reinterpret_cast<const float4*>(w + n * CK + q * 4);
        acc[n] = fmaf(av[q].x, wv.x, acc[n]);
        acc[n] = fmaf(av[q].y, wv.y, acc[n]);
        acc[n] = fmaf(av[q].z, wv.z, acc[n]);
        acc[n] = fmaf(av[q].w, wv.w, acc[n]);
      }
    }
  }
  const size_t dst = ((size_t)b * p.Hd + a_ * p.ds + g.py) * p.Wd + bx * p.ds + g.px;
#pragma unroll
  for (int n = 0; n < N; ++n) {
    const size_t idx = dst * N + n;
    p.C[idx] = bg::apply_epilogue(p, acc[n], idx, n);
  }
}

// ------------------------------------------------------------------------------------------------
// thin-N forward on MFMA (stride 1, N*k <= 16; the generator's last conv 32->3 / 16->3):
// the channel contraction runs on v_mfma_f32_16x16x4_f32 with the (kw, n) pairs as the 16 MFMA columns,
//   P[x'][kw*N + n] = sum_{kh, c} in[oy + kh - pt][x'][c] * w[kh][kw][n][c]        (x' = input column)
// and the kw shift is a 5-term add afterwards:  y[oy][ox][n] = sum_kw P[ox + kw - pl][kw*N + n].
// One wave per output row of a 64-column tile, 5 M-tiles of 16 input columns, channels staged in chunks of 16
// (pixel stride 17 floats -> conflict-free b32 fragment reads).
// ------------------------------------------------------------------------------------------------
typedef float floatx4 __attribute__((ext_vector_type(4)));
constexpr int kTnRows = 4, kTnCols = 64, kTnPx = 80, kTnCc = 16, kTnCS = kTnCc + 1;

__global__ __launch_bounds__(256) void conv_thin_n_mfma_kernel(const GatherParams p, int k, int pt, int pl) {
  extern __shared__ __attribute__((aligned(16))) float tn_lds[];
  const int b = blockIdx.z, oy0 = blockIdx.y * kTnRows, ox0 = blockIdx.x * kTnCols;
  const int N = p.N, Ck = p.Ck, prow = kTnRows + k - 1;
  float* patch = tn_lds;                                   // [prow][kTnPx][kTnCS]
  float* Ws = patch + prow * kTnPx * kTnCS;                // [k][kTnCc][16]
  float* Pb = Ws + k * kTnCc * 16;                         // [4 waves][kTnPx][17]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* src = p.A + (size_t)b * p.Hs * p.Ws * Ck;
  floatx4 acc[5];
#pragma unroll
  for (int t = 0; t < 5; ++t) acc[t] = floatx4{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fk = lane >> 4;
  for (int c0 = 0; c0 < Ck; c0 += kTnCc) {
    __syncthreads();                                        // previous chunk fully consumed
    // patch chunk: input rows oy0-pt .., columns ox0-pl .. (zero outside the image / past the tile)
    for (int idx = tid; idx < prow * kTnPx * (kTnCc / 4); idx += 256) {
      const int q = idx % (kTnCc / 4), pix = idx / (kTnCc / 4);
      const int py = pix / kTnPx, px = pix - py * kTnPx;
      const int sy = oy0 - pt + py, sx = ox0 - pl + px;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws && c0 + q * 4 < Ck)
        v = *reinterpret_cast<const float4*>(src + ((size_t)sy * p.Ws + sx) * Ck + c0 + q * 4);
      float* d = patch + pix * kTnCS + q * 4;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    // weights chunk: Ws[kh][c][kw*N + n] = Wt[kh*k + kw][n][c0 + c]
    for (int idx = tid; idx < k * kTnCc * 16; idx += 256) {
      const int j = idx & 15, c = (idx >> 4) % kTnCc, kh = idx / (16 * kTnCc);
      const int kw = j / N, n = j - kw * N;
      float v = 0.f;
      if (kw < k && c0 + c < Ck) v = p.Wt[((size_t)(kh * k + kw) * N + n) * Ck + c0 + c];
      Ws[idx] = v;
    }
    __syncthreads();
    const float* prow0 = patch + (wave * kTnPx + frow) * kTnCS + fk;     // this wave's output row, kh = 0
    for (int kh = 0; kh < k; ++kh) {
      const float* pa = prow0 + kh * kTnPx * kTnCS;
      const float* pw = Ws + (kh * kTnCc + fk) * 16 + frow;
#pragma unroll
      for (int ks = 0; ks < kTnCc / 4; ++ks) {
        const float bv = pw[ks * 4 * 16];
#pragma unroll
        for (int t = 0; t < 5; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[t * 16 * kTnCS + ks * 4], bv, acc[t], 0, 0, 0);
      }
    }
  }
  // P tile -> LDS: reg r of lane l holds P[x' = t*16 + 4*(l>>4) + r][col = l&15]
  float* P = Pb + wave * kTnPx * 17;
#pragma unroll
  for (int t = 0; t < 5; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) P[(t * 16 + 4 * fk + r) * 17 + frow] = acc[t][r];
  __syncthreads();
  const int oy = oy0 + wave, ox = ox0 + lane;
  if (oy < p.Hd && ox < p.Wd) {
    const size_t dst = ((size_t)b * p.Hd + oy) * p.Wd + ox;
    for (int n = 0; n < N; ++n) {
      float v = 0.f;
      for (int kw = 0; kw < k; ++kw) v += P[(lane + kw) * 17 + kw * N + n];
      const size_t idx = dst * N + n;
      p.C[idx] = bg::apply_epilogue(p, v, idx, n);
    }
  }
}

// generic thin-N fallback (any CK % 4 == 0): one thread per output pixel, operands straight from L1/L2
__global__ __launch_bounds__(256) void conv_thin_n_kernel(const GatherParams p) {
  const GatherPhase& g = p.ph[blockIdx.z];
  const int Mph = p.B * g.Ha * g.Wa;
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= Mph) return;
  RowAnchor ra;
  int dst;
  bg::decode_row(p, g, m, Mph, ra, dst);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const int N = p.N, Ck = p.Ck;
  for (int t = 0; t < g.ntaps; ++t) {
    const int tp = g.tap[t];
    const int sy = ra.ay + bg::tap_dy(tp), sx = ra.ax + bg::tap_dx(tp);
    if ((unsigned)sy >= (unsigned)p.Hs || (unsigned)sx >= (unsigned)p.Ws) continue;
    const float* a = p.A + ((size_t)(ra.b * p.Hs + sy) * p.Ws + sx) * Ck;
    const float* w = p.Wt + (size_t)bg::tap_wi(tp) * N * Ck;   // wave-uniform -> scalar loads
    for (int c = 0; c < Ck; c += 4) {
      const float4 av = *reinterpret_cast<const float4*>(a + c);
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        if (n < N) {
          const float4 wv = *reinterpret_cast<const float4*>(w + n * Ck + c);
          acc[n] = fmaf(av.x, wv.x, acc[n]);
          acc[n] = fmaf(av.y, wv.y, acc[n]);
          acc[n] = fmaf(av.z, wv.z, acc[n]);
          acc[n] = fmaf(av.w, wv.w, acc[n]);
        }
      }
    }
  }
#pragma unroll
  for (int n = 0; n < 4; ++n)
    if (n < N) {
      const size_t idx = (size_t)dst * N + n;
      p.C[idx] = bg::apply_epilogue(p, acc[n], idx, n);
    }
}

// ------------------------------------------------------------------------------------------------
// thin-K on MFMA (Ck <= 4 input channels: critic's first conv, data-grad of the RGB conv, MNIST 1->64).
// The tap is folded into the contraction index, kf = tap*Ck + c (<= 100), so one 8 x 16 anchor tile is
// C[128 px, N] = A[128, KF] * W[KF, N] with A read straight out of the LDS-resident input patch:
// lane address = pixel base + koff[kf] (per-k offset table in LDS).  Weights for the whole N tile sit in LDS.
// ------------------------------------------------------------------------------------------------
constexpr int kTkTH = 8, kTkTW = 16, kTkMaxKF = 104;

template <int NT>
__global__ __launch_bounds__(256) void conv_thin_k_mfma_kernel(const GatherParams p) {
  extern __shared__ __attribute__((aligned(16))) float tk_lds[];
  constexpr int NP = 32 * NT;
  __shared__ int koff[kTkMaxKF];
  const int phase = blockIdx.z % p.nphase, b = blockIdx.z / p.nphase;
  const GatherPhase& g = p.ph[phase];
  const int a0 = blockIdx.y * kTkTH, x0 = blockIdx.x * kTkTW;
  if (a0 >= g.Ha || x0 >= g.Wa) return;
  const int n0 = 0;   // grid covers one N tile (N <= 32*NT), see dispatch
  const int Ck = p.Ck;
  int dmin_y = 127, dmax_y = -127, dmin_x = 127, dmax_x = -127;
  for (int t = 0; t < g.ntaps; ++t) {
    const int dy = bg::tap_dy(g.tap[t]), dx = bg::tap_dx(g.tap[t]);
    dmin_y = min(dmin_y, dy); dmax_y = max(dmax_y, dy);
    dmin_x = min(dmin_x, dx); dmax_x = max(dmax_x, dx);
  }
  const int PH = (kTkTH - 1) * p.ss + (dmax_y - dmin_y) + 1;
  const int PW = (kTkTW - 1) * p.ss + (dmax_x - dmin_x) + 1;
  const int KF = g.ntaps * Ck, KFP = (KF + 1) & ~1;
  float* Bs = tk_lds;                       // [KFP][NP]
  float* patch = tk_lds + kTkMaxKF * NP;    // [PH*PW][Ck]
  const int tid = threadIdx.x;
  // weights: Bs[kf][n] = Wt[wi(t)][n][c]
  for (int idx = tid; idx < KFP * NP; idx += 256) {
    const int kf = idx / NP, n = idx - kf * NP;
    float v = 0.f;
    if (kf < KF && n0 + n < p.N) {
      const int t = kf / Ck, c = kf - t * Ck;
      v = p.Wt[((size_t)bg::tap_wi(g.tap[t]) * p.N + n0 + n) * Ck + c];
    }
    Bs[idx] = v;
  }
  if (tid < KFP) {
    int o = 0;
    if (tid < KF) {
      const int t = tid / Ck, c = tid - t * Ck;
      o = ((bg::tap_dy(g.tap[t]) - dmin_y) * PW + bg::tap_dx(g.tap[t]) - dmin_x) * Ck + c;
    }
    koff[tid] = o;
  }
  const int sy0 = a0 * p.ss + dmin_y, sx0 = x0 * p.ss + dmin_x;
  const float* src = p.A + (size_t)b * p.Hs * p.Ws * Ck;
  const int rowlen = PW * Ck;
  for (int idx = tid; idx < PH * rowlen; idx += 256) {
    const int py = idx / rowlen, rem = idx - py * rowlen;
    const int px = rem / Ck;
    const int sy = sy0 + py, sx = sx0 + px;
    float v = 0.f;
    if ((unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws) v = src[((size_t)sy * p.Ws + sx0) * Ck + rem];
    patch[idx] = v;
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, half = lane >> 5;
  const int ty = 2 * wave + (i >> 4), tx = i & 15;
  const float* abase = patch + (ty * p.ss * PW + tx * p.ss) * Ck;
  const float* bbase = Bs + i;
  floatx16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll 2
  for (int kk = 0; kk < KFP; kk += 2) {
    const int k = kk + half;
    const float a = abase[koff[k]];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bbase[k * NP + j * 32], acc[j], 0, 0, 0);
  }
  // epilogue: reg r of lane l -> pixel row (r&3) + 8*(r>>2) + 4*(l>>5) of this wave's 32, column l&31
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int pi = (r & 3) + 8 * (r >> 2) + 4 * half;
    const int a_ = a0 + 2 * wave + (pi >> 4), bx = x0 + (pi & 15);
    if (a_ < g.Ha && bx < g.Wa) {
      const size_t dst = ((size_t)b * p.Hd + a_ * p.ds + g.py) * p.Wd + bx * p.ds + g.px;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = n0 + j * 32 + i;
        if (n < p.N) {
          const size_t idx = dst * p.N + n;
          p.C[idx] = bg::apply_epilogue(p, acc[j][r], idx, n);
        }
      }
    }
  }
}

// The same contraction with BOTH operands read straight from global memory (round 5).  The staged form above is a latency chain
// on the layers it serves (MNIST's Conv 1 -> 64: 2 workgroups per image, 26 MFMAs per wave): a gather loop for the weight panel,
// a loop for the patch, a barrier, then a handful of MFMAs and the stores -- 18-24 us per launch for 10 MB of traffic and 0.8 us of
// matrix work.  Here a lane issues its whole operand stream up front -- per k-pair one activation (its pixel at that tap, poisoned
// offset = zero padding) and NT weights (its column) through buffer descriptors, sixteen k-pairs in flight -- and the only LDS use is
// a 104-entry table of unpacked taps.  Same products in the same order: bit-identical to the staged kernel.
template <int NT>
__global__ __launch_bounds__(256) void conv_thin_k_direct_kernel(const GatherParams p) {
  __shared__ int ktab[kTkMaxKF];
  const int phase = blockIdx.z % p.nphase, b = blockIdx.z / p.nphase;
  const GatherPhase& g = p.ph[phase];
  const int a0 = blockIdx.y * kTkTH, x0 = blockIdx.x * kTkTW;
  if (a0 >= g.Ha || x0 >= g.Wa) return;
  const int Ck = p.Ck, N = p.N;
  const int KF = g.ntaps * Ck, KP = (KF + 1) >> 1;
  const int tid = threadIdx.x;
  if (tid < 2 * KP) {                      // entry k = tap * Ck + c: dy + 64 | dx + 64 << 8 | c << 16 | weight tap << 20; -1 past KF
    int e = -1;
    if (tid < KF) {
      const int t = tid / Ck, c = tid - t * Ck;
      const int tp = g.tap[t];
      e = (bg::tap_dy(tp) + 64) | ((bg::tap_dx(tp) + 64) << 8) | (c << 16) | (bg::tap_wi(tp) << 20);
    }
    ktab[tid] = e;
  }
  __syncthreads();
  constexpr unsigned kOob = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Wt), 0, (int)p.w_bytes, 0x00020000);
  const int lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, half = lane >> 5;
  const int ty = 2 * wave + (i >> 4), tx = i & 15;
  const int ay = (a0 + ty) * p.ss, ax = (x0 + tx) * p.ss;
  const bool pix_ok = a0 + ty < g.Ha && x0 + tx < g.Wa;
  const int img = b * p.Hs;
  floatx16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  constexpr int D = 16;                    // k-pairs in flight (MNIST: 13 k-pairs = one round of loads)
  for (int kp0 = 0; kp0 < KP; kp0 += D) {
    unsigned av[D], bv[D][NT];                  // raw bits of the loaded floats
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int kp = kp0 + d;
      const int e = kp < KP ? ktab[2 * kp + half] : -1;
      unsigned aoff = kOob, woff[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) woff[j] = kOob;
      if (e >= 0) {
        const int sy = ay + (e & 0xff) - 64, sx = ax + ((e >> 8) & 0xff) - 64, c = (e >> 16) & 0xf, wi = e >> 20;
        if (pix_ok && (unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws) aoff = (unsigned)(((img + sy) * p.Ws + sx) * Ck + c) * 4u;
#pragma unroll
        for (int j = 0; j < NT; ++j)
          if (j * 32 + i < N) woff[j] = (unsigned)((wi * N + j * 32 + i) * Ck + c) * 4u;
      }
      av[d] = __builtin_amdgcn_raw_buffer_load_b32(rsA, aoff, 0, 0);
#pragma unroll
      for (int j = 0; j < NT; ++j) bv[d][j] = __builtin_amdgcn_raw_buffer_load_b32(rsW, woff[j], 0, 0);
    }
#pragma unroll
    for (int d = 0; d < D; ++d)
      if (kp0 + d < KP) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, av[d]), __builtin_bit_cast(float, bv[d][j]), acc[j], 0, 0, 0);
      }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int pi = (r & 3) + 8 * (r >> 2) + 4 * half;
    const int a_ = a0 + 2 * wave + (pi >> 4), bx = x0 + (pi & 15);
    if (a_ < g.Ha && bx < g.Wa) {
      const size_t dst = ((size_t)b * p.Hd + a_ * p.ds + g.py) * p.Wd + bx * p.ds + g.px;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = j * 32 + i;
        if (n < N) {
          const size_t idx = dst * N + n;
          p.C[idx] = bg::apply_epilogue(p, acc[j][r], idx, n);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// thin-K: Ck <= 4 input channels; one thread per output pixel x 32 consecutive output channels
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_thin_k_kernel(const GatherParams p) {
  const GatherPhase& g = p.ph[blockIdx.z];
  const int Mph = p.B * g.Ha * g.Wa;
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= Mph) return;
  const int n0 = blockIdx.y * 32;
  RowAnchor ra;
  int dst;
  bg::decode_row(p, g, m, Mph, ra, dst);
  float acc[32];
#pragma unroll
  for (int n = 0; n < 32; ++n) acc[n] = 0.f;
  const int N = p.N, Ck = p.Ck;
  for (int t = 0; t < g.ntaps; ++t) {
    const int tp = g.tap[t];
    const int sy = ra.ay + bg::tap_dy(tp), sx = ra.ax + bg::tap_dx(tp);
    if ((unsigned)sy >= (unsigned)p.Hs || (unsigned)sx >= (unsigned)p.Ws) continue;
    const float* a = p.A + ((size_t)(ra.b * p.Hs + sy) * p.Ws + sx) * Ck;
    const float* w = p.Wt + ((size_t)bg::tap_wi(tp) * N + n0) * Ck;   // wave-uniform
    float av[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) av[c] = c < Ck ? a[c] : 0.f;
#pragma unroll
    for (int n = 0; n < 32; ++n) {
      if (n0 + n < N) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (c < Ck) acc[n] = fmaf(av[c], w[n * Ck + c], acc[n]);
      }
    }
  }
#pragma unroll
  for (int n = 0; n < 32; ++n)
    if (n0 + n < N) {
      const size_t idx = (size_t)dst * N + n0 + n;
      p.C[idx] = bg::apply_epilogue(p, acc[n], idx, n0 + n);
    }
}

// ------------------------------------------------------------------------------------------------
// catch-all: one thread per output element
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_direct_kernel(const GatherParams p) {
  const GatherPhase& g = p.ph[blockIdx.z];
  const int Mph = p.B * g.Ha * g.Wa;
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (size_t)Mph * p.N) return;
  const int m = (int)(e / p.N), n = (int)(e - (size_t)m * p.N);
  RowAnchor ra;
  int dst;
  bg::decode_row(p, g, m, Mph, ra, dst);
  float acc = 0.f;
  for (int t = 0; t < g.ntaps; ++t) {
    const int tp = g.tap[t];
    const int sy = ra.ay + bg::tap_dy(tp), sx = ra.ax + bg::tap_dx(tp);
    if ((unsigned)sy >= (unsigned)p.Hs || (unsigned)sx >= (unsigned)p.Ws) continue;
    const float* a = p.A + ((size_t)(ra.b * p.Hs + sy) * p.Ws + sx) * p.Ck;
    const float* w = p.Wt + ((size_t)bg::tap_wi(tp) * p.N + n) * p.Ck;
    for (int c = 0; c < p.Ck; ++c) acc = fmaf(a[c], w[c], acc);
  }
  const size_t idx = (size_t)dst * p.N + n;
  p.C[idx] = bg::apply_epilogue(p, acc, idx, n);
}

// split-K tail: C = epilogue(sum over splits of the partial slabs)
// Four consecutive channels per thread (the MFMA path has N % 4 == 0, 16-byte aligned slabs and output, tensors < 2^31 elements):
// float4 loads of every slab, one epilogue of four, one float4 store -- and 32-bit index arithmetic (the scalar form paid a 64-bit
// modulo per element).  Slabs are summed in split order: deterministic.
__global__ __launch_bounds__(256) void igemm_splitk_reduce_kernel(const GatherParams p, size_t total) {
  const unsigned total4 = (unsigned)(total >> 2);
  for (unsigned q = blockIdx.x * 256u + threadIdx.x; q < total4; q += gridDim.x * 256u) {
    const unsigned e = q * 4u;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    // the slabs of an element four at a time (added in slab order, as before): one load per iteration, waited for before the
    // next, made this 5 us launch a chain of ksplit global round trips
    int z = 0;
    for (; z + 4 <= p.ksplit; z += 4) {
      float4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const float4*>(p.slab + (size_t)(z + i) * total + e);
#pragma unroll
      for (int i = 0; i < 4; ++i) { s.x += v[i].x; s.y += v[i].y; s.z += v[i].z; s.w += v[i].w; }
    }
    if (z < p.ksplit) {
      float4 v[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) v[i] = z + i < p.ksplit ? *reinterpret_cast<const float4*>(p.slab + (size_t)(z + i) * total + e) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int i = 0; i < 3; ++i)
        if (z + i < p.ksplit) { s.x += v[i].x; s.y += v[i].y; s.z += v[i].z; s.w += v[i].w; }
    }
    const unsigned n = e % (unsigned)p.N;
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f), m4 = make_float4(1.f, 1.f, 1.f, 1.f);
    if (p.bias) b4 = *reinterpret_cast<const float4*>(p.bias + n);
    if (p.epi_mode == BG_EPI_AFFINE_LRELU) m4 = *reinterpret_cast<const float4*>(p.ref + n);
    *reinterpret_cast<float4*>(p.C + e) = bg::apply_epilogue4(p, s, e, reinterpret_cast<const float*>(&b4), reinterpret_cast<const float*>(&m4));
  }
}

__global__ __launch_bounds__(256) void transpose_last2_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                              int R, int C) {
  __shared__ float tile[32][33];
  const float* s = src + (size_t)blockIdx.z * R * C;
  float* d = dst + (size_t)blockIdx.z * R * C;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  for (int i = ty; i < 32; i += 8)
    if (r0 + i < R && c0 + tx < C) tile[i][tx] = s[(size_t)(r0 + i) * C + c0 + tx];
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (c0 + i < C && r0 + tx < R) d[(size_t)(c0 + i) * R + r0 + tx] = tile[tx][i];
}

// All kernels of one network in ONE launch: desc[e] = {src_off, dst_off, T, R, C, first_tile} (float offsets into the two base
// pointers; 64 x 64 tiles numbered [t][r-tile][c-tile] from first_tile).  float4 on both sides when R and C are multiples of 4.
struct TransposeDesc { int src_off, dst_off, T, R, C, first_tile; };

__global__ __launch_bounds__(256) void transpose_batched_kernel(const float* __restrict__ src_base, float* __restrict__ dst_base,
                                                                const TransposeDesc* __restrict__ desc, int n) {
  __shared__ float tile[64][65];
  int e = 0;
  while (e + 1 < n && desc[e + 1].first_tile <= (int)blockIdx.x) ++e;      // wave-uniform scan of a short table
  const TransposeDesc d = desc[e];
  const int tr = (d.R + 63) / 64, tc = (d.C + 63) / 64;
  int local = blockIdx.x - d.first_tile;
  const int t = local / (tr * tc);
  local -= t * tr * tc;
  const int r0 = (local / tc) * 64, c0 = (local % tc) * 64;
  const float* s = src_base + d.src_off + (size_t)t * d.R * d.C;
  float* o = dst_base + d.dst_off + (size_t)t * d.R * d.C;
  const int tid = threadIdx.x;
  if (((d.R | d.C) & 3) == 0) {
    const int q = tid & 15, rr = tid >> 4;                    // 16 float4 across 64 columns, 16 rows per pass
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = r0 + rr + 16 * i, c = c0 + 4 * q;
      if (r < d.R && c < d.C) {
        const float4 v = *reinterpret_cast<const float4*>(s + (size_t)r * d.C + c);
        tile[rr + 16 * i][4 * q] = v.x; tile[rr + 16 * i][4 * q + 1] = v.y; tile[rr + 16 * i][4 * q + 2] = v.z; tile[rr + 16 * i][4 * q + 3] = v.w;
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = c0 + rr + 16 * i, r = r0 + 4 * q;
      if (c < d.C && r < d.R)
        *reinterpret_cast<float4*>(o + (size_t)c * d.R + r) =
            make_float4(tile[4 * q][rr + 16 * i], tile[4 * q + 1][rr + 16 * i], tile[4 * q + 2][rr + 16 * i], tile[4 * q + 3][rr + 16 * i]);
    }
  } else {
    const int tx = tid & 63, ty = tid >> 6;
    for (int i = ty; i < 64; i += 4)
      if (r0 + i < d.R && c0 + tx < d.C) tile[i][tx] = s[(size_t)(r0 + i) * d.C + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 64; i += 4)
      if (c0 + i < d.C && r0 + tx < d.R) o[(size_t)(c0 + i) * d.R + r0 + tx] = tile[tx][i];
  }
}

// ------------------------------------------------------------------------------------------------
// host dispatch
// ------------------------------------------------------------------------------------------------

int max_phase_m(const GatherParams& p) {
  int mx = 0;
  for (int i = 0; i < p.nphase; ++i) mx = std::max(mx, p.B * p.ph[i].Ha * p.ph[i].Wa);
  return mx;
}

// algorithmic bytes of one launch: source tensor and the used weight taps read once, result written once
double gather_bytes(const GatherParams& p) {
  bool used[bg::kMaxTaps] = {false};
  int nw = 0;
  for (int i = 0; i < p.nphase; ++i)
    for (int t = 0; t < p.ph[i].ntaps; ++t) {
      const int wi = bg::tap_wi(p.ph[i].tap[t]);
      if (wi < bg::kMaxTaps && !used[wi]) { used[wi] = true; ++nw; }
    }
  return 4.0 * ((double)p.B * p.Hs * p.Ws * p.Ck + (double)nw * p.N * p.Ck + (double)p.B * p.Hd * p.Wd * p.N);
}

double gather_flops(const GatherParams& p) {
  double f = 0;
  for (int i = 0; i < p.nphase; ++i) f += 2.0 * p.B * p.ph[i].Ha * p.ph[i].Wa * (double)p.N * p.Ck * p.ph[i].ntaps;
  return f;
}

// Flops the launch ISSUES on the matrix pipe: every workgroup runs whole BM x BN MFMA tiles (row and column padding
// included) over its compact tap list -- position-major tiles drop the taps that are zero padding at their output position.
double gather_exec_flops(const GatherParams& p, int BM, int BN) {
  const double ntile = (double)bg::cdiv(p.N, BN) * BN;
  double f = 0;
  for (int i = 0; i < p.nphase; ++i) {
    const GatherPhase& g = p.ph[i];
    const long Mph = (long)p.B * g.Ha * g.Wa;
    if (!p.pos_major) {
      f += 2.0 * (double)bg::cdiv(Mph, BM) * BM * ntile * p.Ck * g.ntaps;
      continue;
    }
    for (int ay = 0; ay < g.Ha; ++ay)
      for (int ax = 0; ax < g.Wa; ++ax) {
        const int sy = ay * p.ss, sx = ax * p.ss;                            // the anchor in source space (decode_row)
        int live = 0;
        for (int t = 0; t < g.ntaps; ++t)
          live += ((unsigned)(sy + bg::tap_dy(g.tap[t])) < (unsigned)p.Hs && (unsigned)(sx + bg::tap_dx(g.tap[t])) < (unsigned)p.Ws) ? 1 : 0;
        f += 2.0 * (double)p.B * ntile * p.Ck * live;
      }
  }
  return f;
}

// split-K plan for small-M layers: fewer than 2 workgroups per CU and a long contraction
int plan_splitk(const GatherParams& p, int bm, int bn, int /*bk*/) {
  const long wgs = (long)bg::cdiv(max_phase_m(p), bm) * bg::cdiv(p.N, bn) * p.nphase;
  // the K loop's length in the units the thresholds below were tuned in (steps of 32 channels; of 16 where 32 does not divide the
  // channel count) -- NOT in the launch's own BK: the plan, and with it the workspace bg_conv2d_splitk_workspace_bytes asks for,
  // must not depend on which K step the dispatcher picks for the layer
  const int plan_bk = p.Ck % 32 == 0 ? 32 : 16;
  int min_steps = 1 << 30;
  for (int i = 0; i < p.nphase; ++i) min_steps = std::min(min_steps, p.ph[i].ntaps * (p.Ck / plan_bk));
  static const int min_wgs = getenv("BG_SPLITK_MIN_WGS") ? atoi(getenv("BG_SPLITK_MIN_WGS")) : 512;
  static const int tgt_wgs = getenv("BG_SPLITK_TARGET") ? atoi(getenv("BG_SPLITK_TARGET")) : 768;
  static const int force_ks = getenv("BG_SPLITK_FORCE") ? atoi(getenv("BG_SPLITK_FORCE")) : 0;   // tuning aid
  if (force_ks > 0 && min_steps >= 16 * force_ks) return force_ks;
  // position-major layers skip their padding taps, so workgroup lengths differ (9 to 16 live taps on a 4x4 map): two rounds of
  // shorter workgroups balance better than one round of long ones (G2: 0.225 -> 0.205 ms data gradient, 0.202 -> 0.189 forward)
  int maxpos = 0;
  for (int i = 0; i < p.nphase; ++i) maxpos = std::max(maxpos, p.ph[i].Ha * p.ph[i].Wa);
  const bool skipping = maxpos <= 16 && p.B >= bm && p.B % bm == 0;      // 8x8 maps lose a quarter of their taps at most: not worth it (G3: 0.22 -> 0.26 ms)
  if (skipping && wgs >= min_wgs && wgs <= 2 * min_wgs && min_steps >= 64) {
    // round 3: the cost-sorted snake order balances the CUs of ONE round of resident workgroups, so the split only has to fill
    // that round (1024 slots), not make short workgroups: 768 tiles run unsplit (D4 forward of the merged critic pass
    // 0.188 -> 0.159 ms), 1024 unsplit (G2 data gradient 0.185 -> 0.173), 512 in two halves as before.  BG_SPLITK_OLD=1: round-2 rule
    static const int old_rule = getenv("BG_SPLITK_OLD") ? 1 : 0;
    const int ks = old_rule ? (int)std::min<long>(4, 4L * min_wgs / wgs) : (int)std::max<long>(1, 2L * min_wgs / wgs);
    return std::max(1, std::min(ks, min_steps / 32));
  }
  if (wgs >= 2 * min_wgs || min_steps < 32) return 1;
  if (wgs >= min_wgs) return min_steps >= 200 ? 2 : 1;      // half a round of workgroups: worth a reduce pass only on very long K loops (G1: -5 %)
  int ks = (int)std::min<long>(8, (tgt_wgs + wgs - 1) / wgs);
  ks = std::min(ks, min_steps / 16);
  return std::max(ks, 1);
}

template <int BM, int BN, int BK, int WMv, int WNv>
int launch_igemm(GatherParams& p, const bg_epilogue* epi, void* stream, const char* name) {
  const int Mmax = max_phase_m(p);
  const size_t total = (size_t)p.B * p.Hd * p.Wd * p.N;
  int ks = plan_splitk(p, BM, BN, BK);
  if (ks > 1 && !(epi && epi->splitk_ws && epi->splitk_ws_bytes >= ks * total * sizeof(float))) ks = 1;
  p.ksplit = ks;
  p.slab = ks > 1 ? static_cast<float*>(epi->splitk_ws) : nullptr;
  {
    static const int no_pm = getenv("BG_NO_POS_MAJOR") ? 1 : 0;
    static const int pm_max = getenv("BG_POS_MAJOR_MAX") ? atoi(getenv("BG_POS_MAJOR_MAX")) : 64;
    int maxpos = 0;
    for (int i = 0; i < p.nphase; ++i) maxpos = std::max(maxpos, p.ph[i].Ha * p.ph[i].Wa);
    p.pos_major = (!no_pm && maxpos <= pm_max && p.B >= BM && p.B % BM == 0) ? 1 : 0;
  }
  p.mtiles = (int)bg::cdiv(Mmax, BM);
  static const int no_swz = getenv("BG_NO_XCD_SWIZZLE") ? 1 : 0;
  p.xcd_swizzle = !no_swz;
  // Phases per workgroup: short-K transposed convs (2-18 K steps per phase) spend a quarter of their time in prologue and
  // epilogue; merging the 4 (or 2) phases of a tile into one workgroup pays the prologue once and hides the epilogues, as long
  // as the grid still fills the chip 4 workgroups per CU.
  p.pmerge = 1;
  if (p.nphase == 4 && ks == 1) {
    bool same = true;
    for (int i = 1; i < 4; ++i) same = same && p.ph[i].Ha == p.ph[0].Ha && p.ph[i].Wa == p.ph[0].Wa;
    int maxsteps = 0;
    for (int i = 0; i < 4; ++i) maxsteps = std::max(maxsteps, p.ph[i].ntaps * (p.Ck / BK));
    const long wgs4 = (long)p.mtiles * bg::cdiv(p.N, BN);
    static const int force_pm = getenv("BG_PMERGE") ? atoi(getenv("BG_PMERGE")) : 0;   // tuning aid
    if (same) {
      if (force_pm) p.pmerge = force_pm;
      else {
        // rounds of resident workgroups x (K steps of the longest workgroup + prologue/epilogue), as in plan_wgrad
        const int lds = 2 * (BM + BN) * (BK + 4) * 4;
        const long slots = 256L * std::max(1, std::min(160 * 1024 / lds, 4));
        const int kc = p.Ck / BK;
        int tot = 0, t03 = p.ph[0].ntaps + p.ph[3].ntaps, t12 = p.ph[1].ntaps + p.ph[2].ntaps;
        for (int i = 0; i < 4; ++i) tot += p.ph[i].ntaps;
        const double longest[3] = {(double)maxsteps, (double)std::max(t03, t12) * kc, (double)tot * kc};   // pm = 1, 2, 4
        double best = 1e30;
        for (int i = 0; i < 3; ++i) {
          const int pmv = 1 << i;
          const double rounds = std::ceil((double)wgs4 * (4 / pmv) / slots);
          const double cost = rounds * (longest[i] + 2.5);
          if (cost < best * 0.97) { best = cost; p.pmerge = pmv; }      // prefer fewer merges on a tie
        }
      }
    }
  }
  dim3 grid(p.mtiles * bg::cdiv(p.N, BN), 1, (p.nphase / p.pmerge) * ks);
  p.order_n = 0;
  p.m_fast = 0;
  static const int no_sort = getenv("BG_NO_TILE_SORT") ? 1 : 0;
  {
    const int pm = p.pmerge, ngroups = p.nphase / pm;
    int cost[4] = {0, 0, 0, 0}, idx[4] = {0, 1, 2, 3};
    for (int g = 0; g < ngroups; ++g)
      for (int q = 0; q < pm; ++q) cost[g] += p.ph[pm == 2 ? (q == 0 ? g : p.nphase - 1 - g) : g * pm + q].ntaps;
    if (!no_sort) std::stable_sort(idx, idx + ngroups, [&](int a, int b) { return cost[a] > cost[b]; });
    for (int g = 0; g < 4; ++g) p.grp_order[g] = (unsigned char)idx[g];
  }
  bool uniform = true;          // every phase has the same anchor grid (else the sorted order leaves some (group, M tile) rows of the
  for (int i = 1; i < p.nphase; ++i) uniform = uniform && p.ph[i].Ha * p.ph[i].Wa == p.ph[0].Ha * p.ph[0].Wa;   // statistics partials unwritten)
  if (p.pos_major && !no_sort && (uniform || !(epi && epi->stats))) {
    // cost of a (phase group, position) pair = live taps of the group's phases there (the K steps its workgroups run)
    struct PP { int cost, code; };
    PP v[256];
    int n = 0;
    const int pm = p.pmerge, ngroups = p.nphase / pm;
    bool fits = true;
    for (int g = 0; g < ngroups && fits; ++g) {
      auto phase_of = [&](int q) { return pm == 2 ? (q == 0 ? g : p.nphase - 1 - g) : g * pm + q; };
      const GatherPhase& g0 = p.ph[phase_of(0)];
      for (int pos = 0; pos < g0.Ha * g0.Wa; ++pos) {
        int live = 0;
        for (int q = 0; q < pm; ++q) {
          const GatherPhase& gq = p.ph[phase_of(q)];
          const int sy = (pos / g0.Wa) * p.ss, sx = (pos % g0.Wa) * p.ss;
          for (int t = 0; t < gq.ntaps; ++t)
            live += ((unsigned)(sy + bg::tap_dy(gq.tap[t])) < (unsigned)p.Hs && (unsigned)(sx + bg::tap_dx(gq.tap[t])) < (unsigned)p.Ws) ? 1 : 0;
        }
        if (n == 256 || pos >= 64 || g >= 4) { fits = false; break; }
        v[n++] = PP{live, (g << 6) | pos};
      }
    }
    if (fits && n > 0) {
      std::stable_sort(v, v + n, [](const PP& a, const PP& b) { return a.cost > b.cost; });
      for (int i = 0; i < n; ++i) p.pp_order[i] = (unsigned char)v[i].code;
      p.order_n = n;
      p.per_pair = ks * (int)bg::cdiv(p.N, BN) * (p.B / BM);
      static const int mfast_env = getenv("BG_MFAST") ? atoi(getenv("BG_MFAST")) : -1;
      const size_t l2 = (size_t)4 << 20;
      p.m_fast = bg::cdiv(p.N, BN) > 1 && p.w_bytes <= l2 && (size_t)p.a_bytes >= 2 * (size_t)p.w_bytes;
      if (mfast_env >= 0) p.m_fast = mfast_env;
      grid = dim3((unsigned)(n * p.per_pair), 1, 1);
    }
  }
  {
    // BatchNorm statistics in the epilogue: plain stores only (no split-K slabs, no bias / activation), one row per workgroup
    const size_t srows = (size_t)(p.nphase / p.pmerge) * p.mtiles;
    constexpr bool has_stats_variant = (BM == 64 && BN == 64) || (BM == 128 && BN == 32);     // the tiles the default plan picks
    const bool st_ok = has_stats_variant && epi && epi->stats && epi->stats_rows && ks == 1 && p.epi_mode == BG_EPI_NONE && !p.bias &&
                       epi->stats_capacity >= srows * 2 * (size_t)p.N;
    p.stats = st_ok ? epi->stats : nullptr;
    if (st_ok) *epi->stats_rows = (int)srows;
    bg::Launch L(stream, name, gather_flops(p), gather_bytes(p));
    if (L.prof) L.exec_flops(gather_exec_flops(p, BM, BN));
    if constexpr (has_stats_variant) {
      if (st_ok) bg::launch((conv_igemm_kernel<BM, BN, BK, WMv, WNv, true>), grid, dim3(WMv * WNv * 64), 0, L.s, p);
      else bg::launch((conv_igemm_kernel<BM, BN, BK, WMv, WNv, false>), grid, dim3(WMv * WNv * 64), 0, L.s, p);
    } else {
      bg::launch((conv_igemm_kernel<BM, BN, BK, WMv, WNv, false>), grid, dim3(WMv * WNv * 64), 0, L.s, p);
    }
    int rc = L.done(name);
    if (rc || ks == 1) return rc;
  }
  bg::Launch L(stream, "conv_igemm_splitk_reduce", 0, (double)(ks + 1) * total * 4);
  bg::launch(igemm_splitk_reduce_kernel, dim3((unsigned)std::min<size_t>(bg::cdiv(total / 4, 256), 2048)), dim3(256), 0, L.s, p, total);
  return L.done("igemm_splitk_reduce_kernel");
}

template <int BK>
int dispatch_igemm(GatherParams& p, const bg_epilogue* epi, void* stream, const char* tag) {
  const int Mmax = max_phase_m(p);
  // Pick the tile that keeps >= 2 workgroups per CU resident (two waves per SIMD: one issues MFMAs while the
  // other runs its loader segment); bigger tiles only when the grid still fills the chip twice over.
  auto wgs = [&](int bm, int bn) { return (long)bg::cdiv(Mmax, bm) * bg::cdiv(p.N, bn) * p.nphase; };
  const long kFull = 2 * 256;
  static const int force = getenv("BG_IGEMM_TILE") ? atoi(getenv("BG_IGEMM_TILE")) : 0;   // tuning aid: 1..4
  int pick;
  if (force) pick = force;
  else if (p.N <= 32) pick = 3;
  else if (wgs(64, 64) <= 4 * kFull || p.N < 128) pick = 4;      // measured: 64x64 at 4 WGs/CU beats the larger tiles
  else pick = 4;
  if (pick == 3 && p.N > 32) pick = 2;
  switch (pick) {
    case 1: return launch_igemm<128, 128, BK, 2, 2>(p, epi, stream, tag);
    case 5: return launch_igemm<128, 128, BK, 2, 4>(p, epi, stream, tag);    // 8 waves, 64x32 per wave
    case 6: return launch_igemm<128, 64, BK, 4, 2>(p, epi, stream, tag);     // 8 waves, 32x32 per wave
    case 2: return launch_igemm<128, 64, BK, 2, 2>(p, epi, stream, tag);
    case 3: return launch_igemm<128, 32, BK, 4, 1>(p, epi, stream, tag);
    default: return launch_igemm<64, 64, BK, 2, 2>(p, epi, stream, tag);
  }
}

int run_gather(GatherParams& p, const bg_epilogue* epi, void* stream, const char* tag) {
  p.stats = nullptr;
  p.epi_mode = BG_EPI_NONE; p.bias = nullptr; p.ref = nullptr; p.keep = nullptr; p.keep_elems = 0; p.alpha = 0.3f; p.scale = 1.f;
  p.ksplit = 1; p.slab = nullptr;
  if (epi) {
    p.epi_mode = epi->mode; p.bias = epi->bias; p.ref = epi->ref; p.keep = epi->keep; p.keep_elems = epi->keep_elems;
    p.alpha = epi->alpha; p.scale = epi->scale;
    BG_REQUIRE(epi->mode >= BG_EPI_NONE && epi->mode <= BG_EPI_AFFINE_LRELU, BG_ERR_UNSUPPORTED, "%s: epilogue mode %d", tag, epi->mode);
    BG_REQUIRE(epi->mode != BG_EPI_MUL_GRAD || epi->ref, BG_ERR_NULL, "%s: BG_EPI_MUL_GRAD needs ref", tag);
    BG_REQUIRE(epi->mode != BG_EPI_AFFINE_LRELU || (epi->ref && epi->bias), BG_ERR_NULL, "%s: BG_EPI_AFFINE_LRELU needs ref (scale) and bias (shift)", tag);
  }
  const int Mmax = max_phase_m(p);
  BG_REQUIRE((size_t)p.B * p.Hd * p.Wd * (size_t)p.N < (1ull << 31) && (size_t)p.B * p.Hs * p.Ws * (size_t)p.Ck < (1ull << 31),
             BG_ERR_UNSUPPORTED, "%s: tensor exceeds 2^31 elements", tag);
  char name[96];
  const size_t a_bytes = (size_t)p.B * p.Hs * p.Ws * p.Ck * sizeof(float);
  int ntap_w = 0;
  for (int i = 0; i < p.nphase; ++i)
    for (int t = 0; t < p.ph[i].ntaps; ++t) ntap_w = std::max(ntap_w, bg::tap_wi(p.ph[i].tap[t]) + 1);
  const size_t w_bytes = (size_t)ntap_w * p.N * p.Ck * sizeof(float);
  // the MFMA kernel moves its output as float4: N % 4 == 0 and 16-byte aligned output / reference / split-K scratch, 4-byte mask
  const auto al = [](const void* q, size_t a) { return (reinterpret_cast<uintptr_t>(q) & (a - 1)) == 0; };
  const bool vec_ok = p.N % 4 == 0 && al(p.C, 16) && al(p.bias, 16) &&
                      ((p.epi_mode != BG_EPI_MUL_GRAD && p.epi_mode != BG_EPI_AFFINE_LRELU) || al(p.ref, 16)) && al(p.keep, 4) &&
                      (p.keep_elems % 4 == 0) && (!epi || al(epi->splitk_ws, 16));
  if (p.Ck % 16 == 0 && p.N > 4 && vec_ok && a_bytes < (1ull << 31) && w_bytes < (1ull << 31)) {
    p.a_bytes = (unsigned)a_bytes;
    p.w_bytes = (unsigned)w_bytes;
    snprintf(name, sizeof name, "conv_igemm_%s", tag);
    static const int force_bk = getenv("BG_IGEMM_BK") ? atoi(getenv("BG_IGEMM_BK")) : 0;   // tuning aid: 16 / 32 for every layer
    // K steps of 16 channels for the layers with 32 or 64 channels per tap (measured per layer on the VALU-lean loader, round 5:
    // 92 / 120 registers and half the LDS put a fifth workgroup on the CU for the 64 x 64 tile and a fourth for the 128 x 32 tile --
    // D2 fwd -4 %, D2 dgrad -10 %, G5 fwd -6 %, G5 dgrad -3 %; from 128 channels per tap on it is level, from 256 on 2-4 % slower)
    const bool bk32 = p.Ck % 32 == 0 && (force_bk == 32 || (force_bk != 16 && p.Ck > 64));
    return bk32 ? dispatch_igemm<32>(p, epi, stream, name) : dispatch_igemm<16>(p, epi, stream, name);
  }
  if (p.nphase == 1 && p.ss == 1 && p.ds == 1 && p.Ck % 4 == 0 && p.Ck >= 16 && p.B <= 65535) {
    // full k x k tap rectangle of a stride-1 forward conv with N*k <= 16 -> MFMA thin-N kernel
    int k = 1;
    while (k * k < p.ph[0].ntaps) ++k;
    const int pt = -bg::tap_dy(p.ph[0].tap[0]), pl = -bg::tap_dx(p.ph[0].tap[0]);
    bool rect = k * k == p.ph[0].ntaps && k <= 5 && p.N * k <= 16;
    for (int t = 0; rect && t < p.ph[0].ntaps; ++t)
      rect = bg::tap_dy(p.ph[0].tap[t]) == t / k - pt && bg::tap_dx(p.ph[0].tap[t]) == t % k - pl && bg::tap_wi(p.ph[0].tap[t]) == t;
    static const int no_tn = getenv("BG_NO_THIN_N_MFMA") ? 1 : 0;
    if (rect && pl + kTnCols + (k - 1 - pl) <= kTnPx && !no_tn) {
      const size_t lds = ((size_t)(kTnRows + k - 1) * kTnPx * kTnCS + (size_t)k * kTnCc * 16 + 4 * kTnPx * 17) * sizeof(float);
      BG_LDS_ATTR_ONCE_V(conv_thin_n_mfma_kernel, 100 * 1024);
      snprintf(name, sizeof name, "conv_thin_n_mfma_%s", tag);
      bg::Launch L(stream, name, gather_flops(p), gather_bytes(p));
      bg::launch(conv_thin_n_mfma_kernel, dim3(bg::cdiv(p.Wd, kTnCols), bg::cdiv(p.Hd, kTnRows), p.B), dim3(256), lds, L.s, p, k, pt, pl);
      return L.done(name);
    }
  }
  if (p.N <= 4 && (p.Ck == 16 || p.Ck == 32 || p.Ck == 64)) {
    // LDS patch kernel: patch = anchor tile + tap halo of the widest phase
    size_t lds = 0;
    int tiles_x = 0, tiles_y = 0, wmax = 0;
    for (int i = 0; i < p.nphase; ++i) wmax = std::max(wmax, p.ph[i].Wa);
    const int kThinTW = wmax <= 16 ? 16 : 32, kThinTH = 256 / kThinTW;
    for (int i = 0; i < p.nphase; ++i) {
      int mny = 127, mxy = -127, mnx = 127, mxx = -127;
      for (int t = 0; t < p.ph[i].ntaps; ++t) {
        const int dy = bg::tap_dy(p.ph[i].tap[t]), dx = bg::tap_dx(p.ph[i].tap[t]);
        mny = std::min(mny, dy); mxy = std::max(mxy, dy); mnx = std::min(mnx, dx); mxx = std::max(mxx, dx);
      }
      const size_t PH = (kThinTH - 1) * p.ss + (mxy - mny) + 1, PW = (kThinTW - 1) * p.ss + (mxx - mnx) + 1;
      lds = std::max(lds, PH * PW * (size_t)(p.Ck + 4) * sizeof(float));
      tiles_x = std::max(tiles_x, (int)bg::cdiv(p.ph[i].Wa, kThinTW));
      tiles_y = std::max(tiles_y, (int)bg::cdiv(p.ph[i].Ha, kThinTH));
    }
    // all phases from one staged patch (transposed convs): union halo, the tile clipped to the anchors the map has
    static const int no_all = getenv("BG_NO_THIN_N_ALL") ? 1 : 0;
    if (p.nphase > 1 && p.nphase <= 4 && !no_all && p.B <= 65535) {          // one wave per phase, 64 anchors per workgroup
      const int kAllTH = 64 / kThinTW;
      int mny = 127, mxy = -127, mnx = 127, mxx = -127, hamax = 0, wamax = 0;
      for (int i = 0; i < p.nphase; ++i) {
        hamax = std::max(hamax, p.ph[i].Ha); wamax = std::max(wamax, p.ph[i].Wa);
        for (int t = 0; t < p.ph[i].ntaps; ++t) {
          const int dy = bg::tap_dy(p.ph[i].tap[t]), dx = bg::tap_dx(p.ph[i].tap[t]);
          mny = std::min(mny, dy); mxy = std::max(mxy, dy); mnx = std::min(mnx, dx); mxx = std::max(mxx, dx);
        }
      }
      const size_t PHu = (size_t)(std::min(kAllTH, hamax) - 1) * p.ss + (mxy - mny) + 1;
      const size_t PWu = (size_t)(std::min(kThinTW, wamax) - 1) * p.ss + (mxx - mnx) + 1;
      const size_t lds_all = (PHu * PWu * (size_t)(p.Ck + 4) + (size_t)ntap_w * p.N * p.Ck) * sizeof(float);
      if (lds_all <= 150 * 1024) {
#define BG_TNA1(CKv, TWv, Nv)                                                                                     \
  do {                                                                                                            \
    BG_LDS_ATTR_ONCE_V((conv_thin_n_patch_all_kernel<CKv, TWv, Nv>), 150 * 1024);                               \
    bg::launch((conv_thin_n_patch_all_kernel<CKv, TWv, Nv>), grid, dim3(256), lds_all, L.s, p, hamax, wamax, ntap_w); \
  } while (0)
#define BG_TNA2(CKv, TWv)                                                                                         \
  do {                                                                                                            \
    if (p.N == 1) BG_TNA1(CKv, TWv, 1);                                                                           \
    else if (p.N == 2) BG_TNA1(CKv, TWv, 2);                                                                      \
    else if (p.N == 3) BG_TNA1(CKv, TWv, 3);                                                                      \
    else BG_TNA1(CKv, TWv, 4);                                                                                    \
  } while (0)
#define BG_TNA(CKv)                                                                                               \
  do {                                                                                                            \
    if (kThinTW == 16) BG_TNA2(CKv, 16);                                                                          \
    else BG_TNA2(CKv, 32);                                                                                        \
  } while (0)
        snprintf(name, sizeof name, "conv_thin_n_patch_all_%s", tag);
        dim3 grid((unsigned)bg::cdiv(wamax, kThinTW), (unsigned)bg::cdiv(hamax, kAllTH), (unsigned)p.B);
        bg::Launch L(stream, name, gather_flops(p), gather_bytes(p));
        if (p.Ck == 16) BG_TNA(16);
        else if (p.Ck == 32) BG_TNA(32);
        else BG_TNA(64);
#undef BG_TNA
#undef BG_TNA2
#undef BG_TNA1
        return L.done(name);
      }
    }
    if (lds <= 150 * 1024 && (size_t)p.B * p.nphase <= 65535) {
#define BG_TNP_ATTR(CKv, TWv) BG_LDS_ATTR_ONCE_V((conv_thin_n_patch_kernel<CKv, TWv>), 150 * 1024)
      BG_TNP_ATTR(16, 32); BG_TNP_ATTR(32, 32); BG_TNP_ATTR(64, 32); BG_TNP_ATTR(16, 16); BG_TNP_ATTR(32, 16); BG_TNP_ATTR(64, 16);
#undef BG_TNP_ATTR
      snprintf(name, sizeof name, "conv_thin_n_patch_%s", tag);
      dim3 grid(tiles_x, tiles_y, p.B * p.nphase);
      bg::Launch L(stream, name, gather_flops(p), gather_bytes(p));
#define BG_TNP(CKv)                                                                                                        \
  do {                                                                                                                     \
    if (kThinTW == 16) bg::launch((conv_thin_n_patch_kernel<CKv, 16>), grid, dim3(256), lds, L.s, p, tiles_x, tiles_y); \
    else bg::launch((conv_thin_n_patch_kernel<CKv, 32>), grid, dim3(256), lds, L.s, p, tiles_x, tiles_y);          \
  } while (0)
      if (p.Ck == 16) BG_TNP(16);
      else if (p.Ck == 32) BG_TNP(32);
      else BG_TNP(64);
#undef BG_TNP
      return L.done(name);
    }
  }
  if (p.N <= 4 && p.Ck % 4 == 0) {
    snprintf(name, sizeof name, "conv_thin_n_%s", tag);
    bg::Launch L(stream, name, gather_flops(p), gather_bytes(p));
    bg::launch(conv_thin_n_kernel, dim3(bg::cdiv(Mmax, 256), 1, p.nphase), dim3(256), 0, L.s, p);
    return L.done(name);
  }
  if (p.Ck <= 4 && p.N <= 64 && (size_t)p.B * p.nphase <= 65535) {
    const int NT = p.N <= 32 ? 1 : 2;
    size_t lds = 0;
    int tiles_x = 0, tiles_y = 0, kfmax = 0;
    for (int i = 0; i < p.nphase; ++i) {
      int mny = 127, mxy = -127, mnx = 127, mxx = -127;
      for (int t = 0; t < p.ph[i].ntaps; ++t) {
        const int dy = bg::tap_dy(p.ph[i].tap[t]), dx = bg::tap_dx(p.ph[i].tap[t]);
        mny = std::min(mny, dy); mxy = std::max(mxy, dy); mnx = std::min(mnx, dx); mxx = std::max(mxx, dx);
      }
      const size_t PH = (kTkTH - 1) * p.ss + (mxy - mny) + 1, PW = (kTkTW - 1) * p.ss + (mxx - mnx) + 1;
      lds = std::max(lds, ((size_t)kTkMaxKF * 32 * NT + PH * PW * p.Ck) * sizeof(float));
      tiles_x = std::max(tiles_x, (int)bg::cdiv(p.ph[i].Wa, kTkTW));
      tiles_y = std::max(tiles_y, (int)bg::cdiv(p.ph[i].Ha, kTkTH));
      kfmax = std::max(kfmax, p.ph[i].ntaps * p.Ck);
    }
    if (kfmax + 1 <= kTkMaxKF && lds <= 60 * 1024) {
      snprintf(name, sizeof name, "conv_thin_k_mfma_%s", tag);
      dim3 grid(tiles_x, tiles_y, p.B * p.nphase);
      bg::Launch L(stream, name, gather_flops(p), gather_bytes(p));
      static const int staged = getenv("BG_THIN_K_STAGED") ? 1 : 0;       // the LDS-staged form (see conv_thin_k_direct_kernel)
      const size_t a_bytes_k = (size_t)p.B * p.Hs * p.Ws * p.Ck * sizeof(float);
      if (!staged && a_bytes_k < (1ull << 31) && w_bytes < (1ull << 31)) {
        p.a_bytes = (unsigned)a_bytes_k;
        p.w_bytes = (unsigned)w_bytes;
        if (NT == 1) bg::launch(conv_thin_k_direct_kernel<1>, grid, dim3(256), 0, L.s, p);
        else bg::launch(conv_thin_k_direct_kernel<2>, grid, dim3(256), 0, L.s, p);
      } else if (NT == 1) bg::launch(conv_thin_k_mfma_kernel<1>, grid, dim3(256), lds, L.s, p);
      else bg::launch(conv_thin_k_mfma_kernel<2>, grid, dim3(256), lds, L.s, p);
      return L.done(name);
    }
  }
  if (p.Ck <= 4) {
    snprintf(name, sizeof name, "conv_thin_k_%s", tag);
    bg::Launch L(stream, name, gather_flops(p), gather_bytes(p));
    bg::launch(conv_thin_k_kernel, dim3(bg::cdiv(Mmax, 256), bg::cdiv(p.N, 32), p.nphase), dim3(256), 0, L.s, p);
    return L.done(name);
  }
  snprintf(name, sizeof name, "conv_direct_%s", tag);
  bg::Launch L(stream, name, gather_flops(p), gather_bytes(p));
  bg::launch(conv_direct_kernel, dim3(bg::cdiv((size_t)Mmax * p.N, 256), 1, p.nphase), dim3(256), 0, L.s, p);
  return L.done(name);
}

int check_conv_args(const char* fn, const void* a, const void* w, const void* c, int B, int H, int W, int Cin, int Cout,
                    int k, int s) {
  BG_REQUIRE(a && w && c, BG_ERR_NULL, "%s: null pointer", fn);
  BG_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, BG_ERR_BAD_SHAPE, "%s: B=%d H=%d W=%d Cin=%d Cout=%d", fn, B, H, W, Cin, Cout);
  BG_REQUIRE(k >= 1 && (k & 1) && k * k <= bg::kMaxTaps, BG_ERR_UNSUPPORTED, "%s: kernel size %d (odd, <= 5 supported)", fn, k);
  BG_REQUIRE(s == 1 || s == 2, BG_ERR_UNSUPPORTED, "%s: stride %d (1 or 2 supported)", fn, s);
  BG_REQUIRE(bg::aligned16(a) && bg::aligned16(w) && bg::aligned16(c), BG_ERR_BAD_ALIGNMENT, "%s: pointers must be 16-byte aligned", fn);
  return BG_OK;
}

}  // namespace

extern "C" {

size_t bg_conv2d_splitk_workspace_bytes(int bwd_data, int B, int H, int W, int Cin, int Cout, int ksize, int stride) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || ksize < 1 || !(ksize & 1) || ksize * ksize > bg::kMaxTaps || (stride != 1 && stride != 2)) return 0;
  GatherParams p;
  memset(&p, 0, sizeof p);
  if (bwd_data) bg::make_bwd_data_params(p, B, H, W, Cin, Cout, ksize, stride);
  else bg::make_fwd_params(p, B, H, W, Cin, Cout, ksize, stride);
  if (p.Ck % 16 != 0 || p.N <= 4) return 0;
  const int bk = p.Ck % 32 == 0 ? 32 : 16;
  const int ks = p.N <= 32 ? plan_splitk(p, 128, 32, bk) : plan_splitk(p, 64, 64, bk);
  return ks > 1 ? (size_t)ks * p.B * p.Hd * p.Wd * p.N * sizeof(float) : 0;
}

int bg_conv2d_fwd(const float* x, const float* wT_d, float* y, int B, int H, int W, int Cin, int Cout, int ksize,
                  int stride, const bg_epilogue* epi, void* stream) {
  if (epi && epi->stats_rows) *epi->stats_rows = 0;
  int rc = check_conv_args("bg_conv2d_fwd", x, wT_d, y, B, H, W, Cin, Cout, ksize, stride);
  if (rc) return rc;
  bg::UsefulScope useful(bg::conv_useful_flops(B, H, W, Cin, Cout, ksize, stride));
  int taken = 0;
  rc = bg::try_conv_c16(0, x, wT_d, y, B, H, W, Cin, Cout, ksize, stride, epi, stream, &taken);
  if (rc || taken) return rc;
  rc = bg::try_conv_rows(0, x, wT_d, y, B, H, W, Cin, Cout, ksize, stride, epi, stream, &taken);
  if (rc || taken) return rc;
  rc = bg::try_conv_rows_gather(0, x, wT_d, y, B, H, W, Cin, Cout, ksize, stride, epi, stream, &taken);
  if (rc || taken) return rc;
  GatherParams p;
  memset(&p, 0, sizeof p);
  bg::make_fwd_params(p, B, H, W, Cin, Cout, ksize, stride);
  p.A = x; p.Wt = wT_d; p.C = y;
  return run_gather(p, epi, stream, "fwd");
}

int bg_conv2d_bwd_data(const float* dy, const float* w_d, float* dx, int B, int H, int W, int Cin, int Cout, int ksize,
                       int stride, const bg_epilogue* epi, void* stream) {
  if (epi && epi->stats_rows) *epi->stats_rows = 0;
  int rc = check_conv_args("bg_conv2d_bwd_data", dy, w_d, dx, B, H, W, Cin, Cout, ksize, stride);
  if (rc) return rc;
  bg::UsefulScope useful(bg::conv_useful_flops(B, H, W, Cin, Cout, ksize, stride));
  int taken = 0;
  rc = bg::try_conv_c16(1, dy, w_d, dx, B, H, W, Cin, Cout, ksize, stride, epi, stream, &taken);
  if (rc || taken) return rc;
  rc = bg::try_conv_rows(1, dy, w_d, dx, B, H, W, Cin, Cout, ksize, stride, epi, stream, &taken);
  if (rc || taken) return rc;
  rc = bg::try_conv_rows_gather(1, dy, w_d, dx, B, H, W, Cin, Cout, ksize, stride, epi, stream, &taken);
  if (rc || taken) return rc;
  GatherParams p;
  memset(&p, 0, sizeof p);
  bg::make_bwd_data_params(p, B, H, W, Cin, Cout, ksize, stride);
  p.A = dy; p.Wt = w_d; p.C = dx;
  return run_gather(p, epi, stream, "dgrad");
}

int bg_transpose_last2(const float* src, float* dst, int T, int R, int C, void* stream) {
  BG_REQUIRE(src && dst, BG_ERR_NULL, "bg_transpose_last2: null pointer");
  BG_REQUIRE(T > 0 && R > 0 && C > 0 && T <= 65535, BG_ERR_BAD_SHAPE, "bg_transpose_last2: T=%d R=%d C=%d", T, R, C);
  bg::Launch L(stream, "transpose_last2", 0, 8.0 * T * R * C);
  bg::launch(transpose_last2_kernel, dim3(bg::cdiv(C, 32), bg::cdiv(R, 32), T), dim3(256), 0, L.s, src, dst, R, C);
  return L.done("transpose_last2_kernel");
}

int bg_transpose_last2_batched(const float* src_base, float* dst_base, const int* desc_d, int n, int total_tiles, void* stream) {
  BG_REQUIRE(src_base && dst_base && desc_d, BG_ERR_NULL, "bg_transpose_last2_batched: null pointer");
  BG_REQUIRE(n > 0 && total_tiles > 0, BG_ERR_BAD_SHAPE, "bg_transpose_last2_batched: n=%d tiles=%d", n, total_tiles);
  BG_REQUIRE(bg::aligned16(src_base) && bg::aligned16(dst_base), BG_ERR_BAD_ALIGNMENT, "bg_transpose_last2_batched: bases must be 16-byte aligned");
  bg::Launch L(stream, "transpose_last2", 0, 0);
  bg::launch(transpose_batched_kernel, dim3((unsigned)total_tiles), dim3(256), 0, L.s, src_base, dst_base,
                     reinterpret_cast<const TransposeDesc*>(desc_d), n);
  return L.done("transpose_batched_kernel");
}

}  // extern "C"
