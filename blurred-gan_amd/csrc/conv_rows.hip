// Row-MFMA kernels for the thin ends of the networks (3-channel images on one side of the conv):
// the generator's last conv 32->3 (forward) and the data gradient of the critic's first conv RGB->32
// (reference demo_celeba.py:117-119 / :62-66; wgan.py:140,166,244 for the gradients).  SURVEY.md 8a rows T1, T2.
//
// Both are the same contraction in "scatter form": every input row iy (pixels x Ck channels, Ck = 16/32/64) is multiplied
// ONCE by the k weight panels  B_kh[c][(kw, n)]  (n = the <= 3 thin channels, k*n <= 16 MFMA columns),
//      P_kh[ix][(kw, n)] = sum_c in[iy][ix][c] * w(kh, kw)[n][c]
// and P_kh lands in output row  y = iy*s + kh - pt;  along the row  out[y][x][n] = sum_{kw : x = ix*s + kw - pl} P[ix][(kw, n)].
//   * data gradient of a stride-s conv: exactly this with the kernel array as it lies in memory ([tap][ci][co]);
//   * stride-1 forward conv: the same with the taps flipped (kh -> k-1-kh, kw -> k-1-kw) and the pads mirrored.
// A wave owns 16 pixels of the row; the input row is read once (coalesced float4 -> LDS -> A fragments of
// v_mfma_f32_16x16x4_f32, four rows in flight), the k*Ck/4 weight fragments stay in registers for the whole kernel, and the k
// output rows in flight live in k accumulators that shift down by s places after every input row.
// A finished row goes through LDS once for the kw shift-add and leaves as contiguous stores with the fused epilogue.
// MFMA-bound (no padding of the thin dimension beyond 15 -> 16 columns): 2 * rows * W * Ck * k * 16 flop.
#include "conv_common.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));

struct RowParams {
  const float* A;    // input rows [B][Hi][Wi][Ck]
  const float* Wt;   // weights [tap][N][Ck]
  float* C;          // output [B][Ho][Wo][N]
  int B, Hi, Wi, Ck, Ho, Wo, N;
  int pt, pl, flip;  // y = iy*S + kh - pt, x = ix*S + kw - pl; tap = flip ? (K-1-kh)*K + (K-1-kw) : kh*K + kw
  int R, strips;     // output rows per strip, strips per image
  int wpr, ipw;      // waves per input row (Wi / 16), images per workgroup (4 / wpr)
  int epi_mode;
  const float* bias;
  const float* ref;
  const unsigned char* keep;
  size_t keep_elems;
  float alpha, scale;
};

__device__ inline int floordiv(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

// tanh without a branch (the library's tanhf has two, and a branch ends the basic block the row loop wants to be): the exponential
// form  sign(x) (1 - 2 / (e^(2|x|) + 1))  (absolute error ~ 2e-7, the hardware exp2 and reciprocal), and below |x| = 0.1, where
// that form loses RELATIVE accuracy, the odd series through x^7 (next term 2e-11).  Selected, not branched.
__device__ inline float tanh_branchless(float x) {
  float ax = fabsf(x);
  ax = ax > 15.f ? 15.f : ax;                                 // a comparison, not fminf: a NaN stays a NaN (tanhf's behaviour)
  const float e = __builtin_amdgcn_exp2f(ax * 2.885390081777927f);          // e^(2 ax)
  const float big = 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
  const float x2 = ax * ax;
  const float small = ax * fmaf(x2, fmaf(x2, fmaf(x2, -17.f / 315.f, 2.f / 15.f), -1.f / 3.f), 1.f);
  return copysignf(ax < 0.1f ? small : big, x);
}

// Ck = 4*KS, K x K taps, stride S of the scatter, PIXW input pixels per workgroup, EP: epilogue as a compile-time constant for the
// two cases the networks use -- 0 = bias only (data gradients), 1 = bias + tanh (the generator's last conv) -- so that the row
// loop is ONE basic block the scheduler can interleave; 2 = any epilogue (uniform branches, per-element loads)
template <int KS, int K, int S, int PIXW, int EP>
__global__ __launch_bounds__(PIXW * 4) void conv_rows_scatter_kernel(const RowParams p) {
  constexpr int NTH = PIXW * 4;                              // one wave per 16 input pixels
  constexpr unsigned kOob = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) float row_lds[];
  constexpr int AS = 4 * KS + 4;                             // A row stride in LDS: conflict-free b32 fragment reads, 16-B aligned rows
  float* abuf = row_lds;                                     // [2][ipw * Wi][AS]   input rows r, r+1
  constexpr int PS = PIXW * 17 + 1;                          // one P slot: PIXW pixels x 17 floats + ONE ZERO the absent kw terms read
  float* pbuf = row_lds + 2 * PIXW * AS;                     // [2][S][PS] finished P tiles (ipw * Wi = PIXW pixels)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int t = wave % p.wpr, img = wave / p.wpr;
  const int strip = blockIdx.x % p.strips, bgrp = blockIdx.x / p.strips;
  const int Ck = 4 * KS, N = p.N;
  const int y0 = strip * p.R, y1 = min(y0 + p.R, p.Ho);
  const int iy_lo = -floordiv(-(y0 + p.pt - (K - 1)), S);          // ceil((y0 + pt - (K-1)) / S)
  const int iy_hi = floordiv(y1 - 1 + p.pt, S);
  const int nrows = iy_hi - iy_lo + 1;
  // The workgroup's images as two buffers: a row that does not exist, an image past the batch or an output element nobody owns is
  // an out-of-range OFFSET (loads return 0, stores are dropped), so the row loop has no branch around a memory instruction and the
  // compiler's wait counts are exact -- with the loads behind `if`s it waited for the rows it had just requested (vmcnt(0) in front
  // of the MFMAs) and the kernel ran at the SUM of its matrix time and its memory latency (round 3: 87 = 47 + 39 us on the
  // generator's last conv at batch 256).
  const int nimg = min(p.ipw, p.B - bgrp * p.ipw);
  const size_t img0 = (size_t)bgrp * p.ipw;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A + img0 * p.Hi * p.Wi * Ck), 0,
                                                                        nimg * p.Hi * p.Wi * Ck * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(p.C + img0 * p.Ho * p.Wo * N, 0, nimg * p.Ho * p.Wo * N * 4, 0x00020000);

  // weight fragments B[k = c][j = kw*N + n], all K kernel rows, resident
  float bw[K][KS];
  {
    const int lc = min(li, K * N - 1);
    const int kw = lc / N, n = lc - kw * N;
#pragma unroll
    for (int kh = 0; kh < K; ++kh) {
      const int tap = p.flip ? (K - 1 - kh) * K + (K - 1 - kw) : kh * K + kw;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const float w = p.Wt[((size_t)tap * N + n) * Ck + 4 * ks + kq];
        bw[kh][ks] = li < K * N ? w : 0.f;
      }
    }
  }
  floatx4 acc[K];
#pragma unroll
  for (int i = 0; i < K; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};

  // Input rows travel global -> registers (coalesced float4) -> LDS -> A fragments; a direct gather of the fragment layout from
  // global (16 cache lines per wave instruction) cost as much as all the MFMAs.  Everything of a row load that does not depend on
  // the row is worked out once.
  constexpr int QPR = KS;                                    // float4 per pixel
  constexpr int NQ = PIXW * QPR / NTH;                       // float4 per thread per row set (PIXW pixels x Ck)
  static_assert(NQ >= 1, "Ck >= 16");
  unsigned g_off[NQ], l_off[NQ];                             // byte offset of (image, row 0, ix, 4 c4) in the workgroup's images
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int f = tid + q * NTH, pix = f / QPR, c4 = f - pix * QPR;          // pix = im * Wi + ix
    const int im = pix / p.Wi, ix = pix - im * p.Wi;
    g_off[q] = im < nimg ? (unsigned)(((im * p.Hi * p.Wi + ix) * Ck + c4 * 4) * 4) : kOob;
    l_off[q] = (unsigned)(pix * AS + c4 * 4);
  }
  const int rowB = p.Wi * Ck * 4;
  auto gload = [&](int r, float4 (&g)[NQ]) {
    const int iy = iy_lo + r;
    const bool row_ok = r < nrows && (unsigned)iy < (unsigned)p.Hi;          // wave-uniform
    const unsigned add = row_ok ? (unsigned)(iy * rowB) : 0u, force = row_ok ? 0u : kOob;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      g[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (g_off[q] + add) | force, 0, 0));
  };
  auto lstore = [&](int buf, const float4 (&g)[NQ]) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) *reinterpret_cast<float4*>(abuf + (size_t)buf * PIXW * AS + l_off[q]) = g[q];
  };
  const float* afrag = abuf + ((size_t)img * p.Wi + 16 * t + li) * AS + kq;
  // Row-independent part of the kw shift-add, per thread: output element e = (image, x, n) of a finished row sums the
  // P columns (kw, n) of the input pixels ix = (x + pl - kw) / S that exist.  Offsets into one P buffer; a term that does not
  // exist reads the slot's zero.
  // output elements per thread: a workgroup's PIXW input pixels become PIXW * S output pixels of N <= 16 / K channels over NTH =
  // 4 * PIXW threads, i.e. ceil(S * N / 4) each (host-checked).  Not a flat 3: an element slot nobody owns still issues its LDS reads
  // and its (dropped) store every trip, and a vector-memory instruction is ~58 cycles of the SIMD's matrix pipe (69 -> 64 us on the
  // 16 -> 3 conv of the 128-pixel generator).
  constexpr int NE = (S * (16 / K) + 3) / 4;
  constexpr int NTERM = (K + S - 1) / S;
  int e_src[NE][NTERM];
  unsigned e_dst[NE];                                        // byte offset of (image, y = 0, x, n) in the workgroup's output, kOob = none
  float e_bias[NE], e_mul[NE];                               // the epilogue's per-channel operands
  int e_n[NE];
  if (tid < 2 * S) pbuf[(size_t)tid * PS + PIXW * 17] = 0.f; // visible after the first barrier below
  {
    const int per_img = p.Wo * N;
#pragma unroll
    for (int q = 0; q < NE; ++q) {
      const int e = tid + q * NTH;
      const int im = e / per_img, rem = e - im * per_img;
      const int x = rem / N, n = rem - x * N;
      const bool ok = e < p.ipw * per_img && im < nimg;
      e_dst[q] = ok ? (unsigned)(((im * p.Ho * p.Wo + x) * N + n) * 4) : kOob;
      e_n[q] = n;
      e_bias[q] = ok && p.bias ? p.bias[n] : 0.f;
      e_mul[q] = ok && p.epi_mode == BG_EPI_AFFINE_LRELU ? p.ref[n] : 0.f;
#pragma unroll
      for (int i = 0; i < NTERM; ++i) {
        const int kw = (x + p.pl) % S + i * S;
        const int num = x + p.pl - kw, ix = num / S;
        e_src[q][i] = (ok && kw < K && num >= 0 && ix < p.Wi) ? (im * p.Wi + ix) * 17 + kw * N + n : PIXW * 17;
      }
    }
  }
  const size_t out0 = img0 * p.Ho * p.Wo * N;               // element index of the workgroup's first output (EP = 1: ref / keep lookups)
  // finished output row y: P tile -> LDS (before the row's barrier), then kw shift-add, epilogue, contiguous stores (after it)
  auto pwrite = [&](const floatx4& pacc, int slot) {
    float* pb = pbuf + (size_t)slot * PS;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) pb[((size_t)img * p.Wi + 16 * t + 4 * kq + rr) * 17 + li] = pacc[rr];
  };
  auto pstore = [&](int slot, int y, bool live) {            // live (wave-uniform): the row belongs to this strip
    const float* pb = pbuf + (size_t)slot * PS;
    float v[NE];
#pragma unroll
    for (int q = 0; q < NE; ++q) {                            // all the reads first: one wait for the lot
      v[q] = pb[e_src[q][0]];
#pragma unroll
      for (int i = 1; i < NTERM; ++i) v[q] += pb[e_src[q][i]];
    }
    const unsigned add = live ? (unsigned)(y * p.Wo * N * 4) : 0u, force = live ? 0u : kOob;
#pragma unroll
    for (int q = 0; q < NE; ++q) {
      const unsigned off = (e_dst[q] + add) | force;
      float o;
      if (EP == 0) {
        o = v[q] + e_bias[q];
      } else if (EP == 1) {
        o = tanh_branchless(v[q] + e_bias[q]);
      } else {
        const bool mine = (off & kOob) == 0;                 // the generic epilogue reads ref / keep at the element's index
        o = mine ? bg::apply_epilogue_pre(p, v[q], out0 + (off >> 2), e_bias[q], e_mul[q]) : 0.f;
      }
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), rsC, off, 0, 0);
    }
  };

  // One input row per trip, one barrier per trip (at its top).  acc[kh] always belongs to output row iy*S + kh - pt of the CURRENT
  // input row: after the row the S finished accumulators go to LDS as P tiles and the rest move down S places.
  //   trip r:  barrier | A fragments of row r <- LDS | MFMAs of row r, and UNDER them: shift-add + epilogue + stores of row r-1's
  //            P tiles, row r+1 registers -> LDS, loads of row r+5 | P tiles of row r -> LDS | accumulators move down
  // Round 3: the kernel used to run [MFMAs] [P tiles] [barrier] [stores] one after the other in every wave, and its time was the
  // SUM of its matrix time and everything else (87 = 47 + 39 us on the generator's last conv at batch 256, whatever the number of
  // waves per SIMD): the waves of a workgroup sit on four SIMDs and meet at the barrier every trip, so a wave's latency chain
  // behind the barrier is idle matrix time on all four.  Now that chain belongs to the PREVIOUS row and sits inside the MFMA block
  // of the same wave (one basic block: no branch, every load and store unconditional with out-of-range offsets).
  // Row k travels in register set k % 4 (four rows in flight), LDS row buffers and P slots alternate by parity: unrolled by four.
  float4 g0[NQ], g1[NQ], g2[NQ], g3[NQ];
  gload(0, g0);
  gload(1, g1);
  gload(2, g2);
  gload(3, g3);
  lstore(0, g0);
  gload(4, g0);
  // Dropped (out-of-range) stores that only exist for the compiler's wait counting: it sizes the vmcnt of a trip's wait by the
  // SHORTEST path that reaches it, which is this prologue (few instructions after a row's loads), and would make every trip wait
  // for the loads of the trip before.  With as many stores here as four trips issue, the loop's waits leave the younger rows alone.
#pragma unroll
  for (int i = 0; i < 4 * NE * S; ++i) __builtin_amdgcn_raw_buffer_store_b32(0u, rsC, kOob + 4u * i, 0, 0);
  int y_prev = 0;                                            // output row of slot 0 of the previous trip (rows y_prev .. y_prev + S - 1)
  bool have_prev = false;
  auto flush_prev = [&](int prev_par) {                      // shift-add, epilogue and stores of the previous trip's P tiles
#pragma unroll
    for (int kh = 0; kh < S; ++kh) {
      const int y = y_prev + kh;
      pstore(prev_par * S + kh, y, have_prev & (y >= y0) & (y < y1));    // & not &&: no branch in the row loop
    }
  };
  auto trip = [&](int r, auto parity, float4 (&g)[NQ]) {    // g: row r+1 on entry, row r+5 in flight on exit
    constexpr int cur = decltype(parity)::value;
    __syncthreads();
    float a[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a[ks] = afrag[(size_t)cur * PIXW * AS + 4 * ks];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)                          // kh innermost: K independent accumulator chains
#pragma unroll
      for (int kh = 0; kh < K; ++kh) acc[kh] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], bw[kh][ks], acc[kh], 0, 0, 0);
    flush_prev(cur ^ 1);
    lstore(cur ^ 1, g);
    gload(r + 5, g);
#pragma unroll
    for (int kh = 0; kh < S; ++kh) pwrite(acc[kh], cur * S + kh);   // rows that just received their last contribution
    y_prev = (iy_lo + r) * S - p.pt;
    have_prev = true;
#pragma unroll
    for (int i = 0; i < K; ++i) acc[i] = i + S < K ? acc[i + S] : floatx4{0.f, 0.f, 0.f, 0.f};
  };
  // whole groups of four trips in the loop (one path through it: exact wait counts), the last one to three after it
  int r = 0;
  for (; r + 4 <= nrows; r += 4) {
    trip(r, std::integral_constant<int, 0>{}, g1);
    trip(r + 1, std::integral_constant<int, 1>{}, g2);
    trip(r + 2, std::integral_constant<int, 0>{}, g3);
    trip(r + 3, std::integral_constant<int, 1>{}, g0);
  }
  int last_par = 1;                                          // parity of the last trip that ran (nrows >= 1)
  if (r < nrows) {
    trip(r, std::integral_constant<int, 0>{}, g1);
    last_par = 0;
    if (r + 1 < nrows) {
      trip(r + 1, std::integral_constant<int, 1>{}, g2);
      last_par = 1;
      if (r + 2 < nrows) {
        trip(r + 2, std::integral_constant<int, 0>{}, g3);
        last_par = 0;
      }
    }
  }
  __syncthreads();
  if (last_par == 0) flush_prev(0); else flush_prev(1);
}

template <int K, int S>
int launch_rows(const RowParams& p, dim3 grid, size_t lds, hipStream_t s) {
  const bool wide = p.Wi == 128;
  const int ep = p.epi_mode == BG_EPI_NONE ? 0 : p.epi_mode == BG_EPI_TANH ? 1 : 2;
#define BG_RS2(KSv, EPv) do { if (wide) bg::launch((conv_rows_scatter_kernel<KSv, K, S, 128, EPv>), grid, dim3(512), lds, s, p); \
                              else bg::launch((conv_rows_scatter_kernel<KSv, K, S, 64, EPv>), grid, dim3(256), lds, s, p); } while (0)
#define BG_RS(KSv) do { if (ep == 0) BG_RS2(KSv, 0); else if (ep == 1) BG_RS2(KSv, 1); else BG_RS2(KSv, 2); } while (0)
  switch (p.Ck) {
    case 16: BG_RS(4); return 1;
    case 32: BG_RS(8); return 1;
    case 64: BG_RS(16); return 1;
    default: return 0;
  }
#undef BG_RS
#undef BG_RS2
}

// ------------------------------------------------------------------------------------------------
// Gather form, thin CONTRACTION side (<= 3 input channels: the critic's first conv RGB -> 32 forward, and the data gradient
// of the generator's last conv): with NHWC and Ct <= 3 channels the k*Ct values under one kernel row are a contiguous window
// of the input row, so the MFMA contraction index is i = kw*Ct + ci (<= 15, one k-step of 4 x 4), per kernel row kh:
//   out[oy][ox][n] = sum_kh sum_i xrow(oy*s + kh - pt)[(ox*s - pl)*Ct + i] * w(kh, i)[n]
// Input rows of a strip sit in LDS with a zero halo (they are tiny: W*Ct floats), the k*4*NT weight fragments stay in
// registers, every wave computes whole output rows (TG tiles of 16 pixels x NT tiles of 16 channels) with no barrier inside
// the strip.  The stride-1 data gradient is the same with flipped taps and mirrored pads.
// MFMA-bound at 2 * B*Ho*Wo * N * 16*k flop, no recompute; HBM-bound on the output for the forward.
// ------------------------------------------------------------------------------------------------
constexpr int kRgHalo = 16;

struct RowGParams {
  const float* A;    // input rows [B][Hi][Wi][Ct]
  const float* Wt;   // weights [tap][N][Ct]
  float* C;          // output [B][Ho][Wo][N]
  int B, Hi, Wi, Ct, Ho, Wo, N;
  int s, pt, pl, flip;
  int R, strips;     // output rows per strip, strips per image
  int epi_mode;
  const float* bias;
  const float* ref;
  const unsigned char* keep;
  size_t keep_elems;
  float alpha, scale;
};

// TG pixel tiles per unit, N = 16 * NT, K x K taps, MASK: the bias + LeakyReLU + dropout-mask epilogue of the critic's first layer
// as a variant of its own (its batched mask loads cost registers the other epilogues' variant should not pay: a wave per SIMD)
template <int TG, int NT, int K, bool MASK>
__global__ __launch_bounds__(256) void conv_rows_gather_kernel(const RowGParams p) {
  extern __shared__ __attribute__((aligned(16))) float xl[];       // [(R-1)*s + K][Wi*Ct + 2*halo]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int Ct = p.Ct, N = 16 * NT, st = p.s;
  const int RS = p.Wi * Ct + 2 * kRgHalo;
  const int strip = blockIdx.x % p.strips, b = blockIdx.x / p.strips;
  const int oy0 = strip * p.R, rows = min(p.R, p.Ho - oy0);
  const int nrows_in = (rows - 1) * st + K;

  // weight fragments B[k = i][j = n]: i = 4*ks + kq = kw*Ct + ci
  float bw[K][4][NT];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int i = 4 * ks + kq, kw = i / Ct, ci = i - kw * Ct;
#pragma unroll
    for (int kh = 0; kh < K; ++kh) {
      const int tap = p.flip ? (K - 1 - kh) * K + (K - 1 - kw) : kh * K + kw;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bw[kh][ks][nt] = i < K * Ct ? p.Wt[((size_t)tap * N + nt * 16 + li) * Ct + ci] : 0.f;
    }
  }
  // stage the strip's input rows (zero outside the image, zero halos: every word the fragments can touch is initialised)
  const int RL = p.Wi * Ct;                                  // floats per input row
  if ((RL & 3) == 0) {
    const int q4 = RS / 4;                                   // float4 per LDS row (halo 16 and RL are multiples of 4)
    bg::stage_rows_f4<4>(xl, nrows_in, q4, RS, tid, [&](int r, int c4) -> const float* {
      const int iy = oy0 * st - p.pt + r, e = c4 * 4 - kRgHalo;
      return ((unsigned)iy < (unsigned)p.Hi && (unsigned)e < (unsigned)RL) ? p.A + ((size_t)b * p.Hi + iy) * RL + e : nullptr;
    });
  } else {
    for (int idx = tid; idx < nrows_in * RS; idx += 256) {
      const int r = idx / RS, c = idx - r * RS;
      const int iy = oy0 * st - p.pt + r, e = c - kRgHalo;
      float v = 0.f;
      if ((unsigned)iy < (unsigned)p.Hi && (unsigned)e < (unsigned)RL) v = p.A[((size_t)b * p.Hi + iy) * RL + e];
      xl[idx] = v;
    }
  }
  __syncthreads();

  // per-column epilogue operands, once per workgroup
  float e_bias[NT], e_mul[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    e_bias[nt] = p.bias ? p.bias[nt * 16 + li] : 0.f;
    e_mul[nt] = p.epi_mode == BG_EPI_AFFINE_LRELU ? p.ref[nt * 16 + li] : 1.f;
  }
  const int groups = p.Wo / (16 * TG);                       // units per output row
  const int nunits = rows * groups;
  for (int u = wave; u < nunits; u += 4) {
    const int r = u / groups, g = u - r * groups;
    const int oy = oy0 + r, ox0 = g * 16 * TG;
    floatx4 acc[TG][NT];
#pragma unroll
    for (int tg = 0; tg < TG; ++tg)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[tg][nt] = floatx4{0.f, 0.f, 0.f, 0.f};
    // A[m = ox][k = i] = xrow[(ox*s - pl)*Ct + i]
    const float* abase = xl + (size_t)(r * st) * RS + kRgHalo + ((ox0 + li) * st - p.pl) * Ct + kq;
#pragma unroll
    for (int kh = 0; kh < K; ++kh) {
      float a[TG][4];
#pragma unroll
      for (int tg = 0; tg < TG; ++tg)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) a[tg][ks] = abase[(size_t)kh * RS + tg * 16 * st * Ct + 4 * ks];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int tg = 0; tg < TG; ++tg)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[tg][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tg][ks], bw[kh][ks][nt], acc[tg][nt], 0, 0, 0);
    }
    // critic forward (bias + LeakyReLU + dropout): the mask bytes of one pixel tile (4 x NT per lane) are requested together, with
    // unconditional loads (an element the mask does not cover reads byte 0 and ignores it).  Inside the per-element epilogue each was
    // a load and a wait of its own -- 16 memory latencies in a row per unit, +30 % on the first critic layer at 3 x 128 samples
    // (round 3).  Tile by tile, not the whole unit at once: the unit's 16 x NT bytes cost the wide variants a wave per SIMD.
    constexpr bool mask_epi = MASK;                           // host: p.epi_mode == BG_EPI_BIAS_LRELU && p.keep
    // reg rr of lane l = out[ox = ox0 + 16*tg + 4*kq + rr][n = 16*nt + li]  (64-B pieces per store; a transpose through LDS to
    // float4 stores measured no faster -- the kernel is bound by its short per-strip life, not by the stores)
#pragma unroll
    for (int tg = 0; tg < TG; ++tg) {
      const size_t pix0 = ((size_t)b * p.Ho + oy) * p.Wo + ox0 + 16 * tg + 4 * kq;
      unsigned char e_keep[4][NT];
      if (mask_epi) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const size_t idx = (pix0 + rr) * N + nt * 16 + li;
            e_keep[rr][nt] = p.keep[(p.keep_elems == 0 || idx < p.keep_elems) ? idx : 0];
          }
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const size_t idx = (pix0 + rr) * N + nt * 16 + li;
          if (mask_epi) {                                      // same arithmetic, same order as apply_epilogue (BG_EPI_BIAS_LRELU)
            float v = acc[tg][nt][rr] + e_bias[nt];
            v = v > 0.f ? v : p.alpha * v;
            if (p.keep_elems == 0 || idx < p.keep_elems) v = e_keep[rr][nt] ? v * p.scale : 0.f;
            p.C[idx] = v;
          } else {
            p.C[idx] = bg::apply_epilogue_pre(p, acc[tg][nt][rr], idx, e_bias[nt], e_mul[nt]);
          }
        }
    }
  }
}

template <int K>
int launch_rows_gather(const RowGParams& p, int tg, dim3 grid, size_t lds, hipStream_t s) {
  const int nt = p.N / 16;
  const bool mask = p.epi_mode == BG_EPI_BIAS_LRELU && p.keep != nullptr;
#define BG_RG(TGv, NTv) do { if (mask) bg::launch((conv_rows_gather_kernel<TGv, NTv, K, true>), grid, dim3(256), lds, s, p); \
                             else bg::launch((conv_rows_gather_kernel<TGv, NTv, K, false>), grid, dim3(256), lds, s, p); } while (0)
  if (tg == 1) { if (nt == 1) BG_RG(1, 1); else if (nt == 2) BG_RG(1, 2); else BG_RG(1, 4); }
  else if (tg == 2) { if (nt == 1) BG_RG(2, 1); else if (nt == 2) BG_RG(2, 2); else BG_RG(2, 4); }
  else { if (nt == 1) BG_RG(4, 1); else if (nt == 2) BG_RG(4, 2); else BG_RG(4, 4); }
#undef BG_RG
  return 1;
}

}  // namespace

namespace bg {

// thin contraction side (<= 3 channels in): forward of any stride, or the stride-1 data gradient
int try_conv_rows_gather(int bwd_data, const float* a, const float* w, float* c, int B, int H, int W, int Cin, int Cout, int k, int s,
                         const bg_epilogue* epi, void* stream, int* taken) {
  *taken = 0;
  static const int off = getenv("BG_NO_ROWS") ? 1 : 0;
  if (off || (k != 5 && k != 3)) return BG_OK;
  RowGParams p;
  memset(&p, 0, sizeof p);
  int Ho, Wo, pt, pl;
  same_pads(H, k, s, &Ho, &pt);
  same_pads(W, k, s, &Wo, &pl);
  if (bwd_data) {      // dy [B,H,W,Cout] (s == 1) -> dx [B,H,W,Cin]
    if (s != 1 || Cout * k > 16 || (Cin != 16 && Cin != 32 && Cin != 64)) return BG_OK;
    p.Hi = H; p.Wi = W; p.Ct = Cout; p.Ho = H; p.Wo = W; p.N = Cin; p.s = 1; p.pt = k - 1 - pt; p.pl = k - 1 - pl; p.flip = 1;
  } else {
    if (Cin * k > 16 || (Cout != 16 && Cout != 32 && Cout != 64)) return BG_OK;
    p.Hi = H; p.Wi = W; p.Ct = Cin; p.Ho = Ho; p.Wo = Wo; p.N = Cout; p.s = s; p.pt = pt; p.pl = pl; p.flip = 0;
  }
  if (p.Wo % 16 != 0) return BG_OK;
  if ((size_t)B * p.Ho * p.Wo * p.N >= (1ull << 31)) return BG_OK;
  const int tiles = p.Wo / 16;
  const int tg = tiles % 4 == 0 ? 4 : (tiles % 2 == 0 ? 2 : 1);
  p.A = a; p.Wt = w; p.C = c; p.B = B;
  static const int rows_r = getenv("BG_ROWSG_R") ? atoi(getenv("BG_ROWSG_R")) : 0;     // tuning aid
  p.R = std::min(rows_r ? rows_r : (s == 2 ? 8 : 16), p.Ho);
  p.strips = (int)cdiv(p.Ho, p.R);
  const size_t lds = ((size_t)((p.R - 1) * p.s + k) * (p.Wi * p.Ct + 2 * kRgHalo)) * sizeof(float);
  if (lds > 64 * 1024) return BG_OK;
  p.epi_mode = BG_EPI_NONE; p.alpha = 0.3f; p.scale = 1.f;
  if (epi) {
    BG_REQUIRE(epi->mode >= BG_EPI_NONE && epi->mode <= BG_EPI_AFFINE_LRELU, BG_ERR_UNSUPPORTED, "conv rows: epilogue mode %d", epi->mode);
    BG_REQUIRE(epi->mode != BG_EPI_MUL_GRAD || epi->ref, BG_ERR_NULL, "conv rows: BG_EPI_MUL_GRAD needs ref");
    BG_REQUIRE(epi->mode != BG_EPI_AFFINE_LRELU || (epi->ref && epi->bias), BG_ERR_NULL, "conv rows: BG_EPI_AFFINE_LRELU needs ref and bias");
    p.epi_mode = epi->mode; p.bias = epi->bias; p.ref = epi->ref; p.keep = epi->keep; p.keep_elems = epi->keep_elems; p.alpha = epi->alpha; p.scale = epi->scale;
  }
  const dim3 grid((unsigned)(B * p.strips));
  const double flops = 2.0 * B * (double)Ho * Wo * Cin * Cout * k * k;
  Launch L(stream, bwd_data ? "conv_rows_thin_k_dgrad" : "conv_rows_thin_k_fwd", flops, 0);
  if (k == 5) launch_rows_gather<5>(p, tg, grid, lds, L.s);
  else launch_rows_gather<3>(p, tg, grid, lds, L.s);
  *taken = 1;
  return L.done("conv_rows_gather_kernel");
}



// Tries the row kernel for a thin-N problem; returns BG_OK + *taken = 1 when it ran, *taken = 0 when the shape is not covered.
//   forward (bwd_data = 0): x [B,H,W,Cin] -> y [B,H,W,Cout], stride 1, weights wT [tap][Cout][Cin]
//   data gradient (bwd_data = 1): dy [B,Ho,Wo,Cout] -> dx [B,H,W,Cin], weights w [tap][Cin][Cout]
int try_conv_rows(int bwd_data, const float* a, const float* w, float* c, int B, int H, int W, int Cin, int Cout, int k, int s,
                  const bg_epilogue* epi, void* stream, int* taken) {
  *taken = 0;
  static const int off = getenv("BG_NO_ROWS") ? 1 : 0;
  if (off || (k != 5 && k != 3)) return BG_OK;
  RowParams p;
  memset(&p, 0, sizeof p);
  int Ho, Wo, pt, pl;
  same_pads(H, k, s, &Ho, &pt);
  same_pads(W, k, s, &Wo, &pl);
  if (bwd_data) {
    if (Cin * k > 16 || (Cout != 16 && Cout != 32 && Cout != 64)) return BG_OK;
    p.Hi = Ho; p.Wi = Wo; p.Ck = Cout; p.Ho = H; p.Wo = W; p.N = Cin; p.pt = pt; p.pl = pl; p.flip = 0;
  } else {
    if (s != 1 || Cout * k > 16 || (Cin != 16 && Cin != 32 && Cin != 64)) return BG_OK;
    p.Hi = H; p.Wi = W; p.Ck = Cin; p.Ho = H; p.Wo = W; p.N = Cout; p.pt = k - 1 - pt; p.pl = k - 1 - pl; p.flip = 1;
  }
  if (p.Wi != 16 && p.Wi != 32 && p.Wi != 64 && p.Wi != 128) return BG_OK;            // a workgroup handles 64 pixels (4 / 2 / 1 images) or one 128-pixel row
  if ((size_t)B * p.Hi * p.Wi * p.Ck >= (1ull << 31) || (size_t)B * p.Ho * p.Wo * p.N >= (1ull << 31)) return BG_OK;
  p.A = a; p.Wt = w; p.C = c; p.B = B;
  static const int rows_r = getenv("BG_ROWS_R") ? atoi(getenv("BG_ROWS_R")) : 0;       // tuning aid
  p.R = std::min(rows_r ? rows_r : (s == 1 ? 32 : 16), p.Ho);                        // measured: halo rows (k-1)/s per strip vs workgroups in flight
  p.strips = (int)cdiv(p.Ho, p.R);
  p.wpr = p.Wi / 16;
  p.ipw = p.Wi == 128 ? 1 : 4 / p.wpr;
  const int pixw = p.Wi == 128 ? 128 : 64;
  if ((long)p.ipw * p.Wo * p.N > (long)((s * (16 / k) + 3) / 4) * pixw * 4) return BG_OK;   // output elements per thread of the shift-add (the kernel's NE)
  if ((size_t)p.ipw * p.Hi * p.Wi * p.Ck >= (1ull << 29) || (size_t)p.ipw * p.Ho * p.Wo * p.N >= (1ull << 29)) return BG_OK;   // a workgroup's images are one buffer: 31-bit byte offsets
  p.epi_mode = BG_EPI_NONE; p.alpha = 0.3f; p.scale = 1.f;
  if (epi) {
    BG_REQUIRE(epi->mode >= BG_EPI_NONE && epi->mode <= BG_EPI_AFFINE_LRELU, BG_ERR_UNSUPPORTED, "conv rows: epilogue mode %d", epi->mode);
    BG_REQUIRE(epi->mode != BG_EPI_MUL_GRAD || epi->ref, BG_ERR_NULL, "conv rows: BG_EPI_MUL_GRAD needs ref");
    BG_REQUIRE(epi->mode != BG_EPI_AFFINE_LRELU || (epi->ref && epi->bias), BG_ERR_NULL, "conv rows: BG_EPI_AFFINE_LRELU needs ref and bias");
    p.epi_mode = epi->mode; p.bias = epi->bias; p.ref = epi->ref; p.keep = epi->keep; p.keep_elems = epi->keep_elems; p.alpha = epi->alpha; p.scale = epi->scale;
  }
  const dim3 grid((unsigned)(cdiv(B, p.ipw) * p.strips));
  const size_t lds = ((size_t)2 * pixw * (p.Ck + 4) + (size_t)2 * s * (pixw * 17 + 1)) * sizeof(float);
  const double flops = 2.0 * B * (double)H * W * Cin * Cout * k * k / (s * s);
  Launch L(stream, bwd_data ? "conv_rows_dgrad" : "conv_rows_fwd", flops, 0);
  int ok;
  if (k == 5) ok = s == 2 ? launch_rows<5, 2>(p, grid, lds, L.s) : launch_rows<5, 1>(p, grid, lds, L.s);
  else ok = s == 2 ? launch_rows<3, 2>(p, grid, lds, L.s) : launch_rows<3, 1>(p, grid, lds, L.s);
  (void)ok;
  *taken = 1;
  return L.done("conv_rows_scatter_kernel");
}

}  // namespace bg
