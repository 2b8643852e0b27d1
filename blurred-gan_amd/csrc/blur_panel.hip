// Wide-tap Gaussian blur, BOTH passes in one launch (gaussian_blur.py:116-130 at the callback's default sigma 23.5 -> 143 taps and at
// maximum_reasonable_std(256) -> 255 taps, callbacks.py:49,74; BASELINE.json configs[4]).
//
// blur_band_t_kernel x 2 (blur.hip) runs one banded Toeplitz product per launch through a scratch image and stores every result
// TRANSPOSED; its notes price the transposed stores at 19 of 101 us and the two launches' memory instructions as almost additive
// to the MFMA time.  Here a workgroup owns a PANEL of 32 output rows of one image and nothing leaves the CU between the passes:
//   pass 1 (H):  Y[32][W*3] = T_H[32 x K] * X[K][W*3],  K = the rows within half a kernel of the panel.  Wave w owns pixels
//                [32w, 32w + 32) x RGB = three 32-column MFMA tiles whose columns are the STRIDED sets {3l + c}: one
//                buffer_load_dwordx3 per lane and k-pair (lanes 0..31 row k, 32..63 row k + 1) IS the B operand of the three
//                MFMAs -- no LDS staging of X, no barrier, nothing shared between waves; eight loads in flight per wave.
//                The Toeplitz fragment comes from a zero-padded tap table in LDS (one ds_read per k-pair, shared by the 3 MFMAs).
//   Y stays in LDS (32 x (W*3 + 1) floats, 98 KB at W = 256; odd pitch: pass 2 reads it with lanes along rows).
//   pass 2 (W):  Z^T[x][r] = sum_x' T_W[x][x'] * Y[r][x', c] per channel, computed TRANSPOSED (A = Toeplitz, B = Y^T from LDS): a
//                lane ends with 4 consecutive pixels x 3 channels of ONE output row = 12 contiguous floats -> three float4 stores,
//                straight into the NHWC result (no transposed store, no scratch image).  Wave w owns the pixel tile [32w, 32w+32).
// Both passes contract over the band only (rows / columns outside the image: the range is clipped; partial pairs: zero taps).
// Workgroups are dealt so that the panels of an image share an XCD (its L2 serves their overlapping row ranges) and the two a CU
// runs (512 for 64 images of 256 rows on 256 CUs) are one long-band and one short-band panel.
// MFMA-bound: 2 * 3 * (K_H + K_W) / 2 MFMAs of 32x32x2 per wave; HBM traffic = the image in (re-read from L2 by the panels that
// share its rows) and out once.
#include "common.h"
#include "blur_panel.h"
#include <algorithm>

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx3 __attribute__((ext_vector_type(3)));
constexpr int kPad = 64;            // zeros on either side of the taps in the LDS table
constexpr int kDepth = 8;           // pass-1 loads in flight per wave
constexpr int kStagePitch = 100;   // floats per staged output row (96 + 4: float4 rows, 16-byte aligned)
#ifndef PANEL_PITCH_PAD
#define PANEL_PITCH_PAD 1
#endif
constexpr int kPitchPad = PANEL_PITCH_PAD;   // Y row pitch = 3 W + kPitchPad

struct PanelParams {
  const float* x;
  float* y;
  const float* taps;
  int B, H, W, T, nrb;
  unsigned char order[16];          // row blocks, longest band first
};

__global__ __launch_bounds__(512) void blur_panel_kernel(const PanelParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kk = lane >> 5;
  const int W = p.W, H = p.H, Q = 3 * W, pitch = Q + kPitchPad, T = p.T, half = T >> 1;
  float* Ys = lds;                                  // [32][pitch]
  float* tz = lds + 32 * pitch;                     // [kPad zeros][T taps][kPad zeros]
  // (image, row block).  The panels of an image re-read its rows (a 32-row panel contracts over up to 32 + T - 1 of them), so
  // they must meet in ONE L2: workgroup w runs on XCD w % 8 (tools/probes/placement.hip), which therefore takes the images
  // xcd, xcd + 8, ... whole, image after image (4 images = 32 workgroups = the XCD's CUs at one panel per CU).  Dealt round-robin
  // the eight panels of an image sat on eight XCDs: L2 hit rate 0.15, 300 MB of HBM-side traffic per launch for a 50 MB image
  // set (3.9 TB/s beside the MFMAs).  Within an image the panels go longest band first, and every second group of four images
  // shortest first, so that the two panels a CU runs are one long and one short.
  int img, rb;
  {
    const int nrb = p.nrb, w = blockIdx.x;
    if ((p.B & 7) == 0) {
      const int xcd = w & 7, j = w >> 3, il = j / nrb, k = j - il * nrb;
      img = il * 8 + xcd;
      rb = p.order[(il & 4) ? nrb - 1 - k : k];
    } else {
      img = w / nrb;
      rb = p.order[w - img * nrb];
    }
  }
  const int r0 = 32 * rb;
  floatx16 acc[3];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;

  // ---------------------------------------------------------------- pass 1: Y = T_H * X, B operand straight from global memory
  {
    const int k0 = max(0, r0 - half) & ~1, k1 = min(H, r0 + 32 + half);
#if defined(BG_DIAG) && defined(PANEL_NO_P1)
    const int K1 = 0, NG = 0;
    (void)k1;
#else
    const int K1 = (k1 - k0 + 1) >> 1, NG = (K1 + kDepth - 1) / kDepth;
#endif
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (size_t)img * H * Q), 0, H * Q * 4, 0x00020000);
    // lane (li, kk): row k0 + 2s + kk, floats 3 * (32 * wave + li) .. + 2;  rows >= H lie past the descriptor's range: zeros
    // the lane's part of the offset stays in a register for the whole pass; the k-pair's part (2 rows per pair) travels in the buffer
    // instruction's SCALAR offset, which the range check includes (tools/probes/buffer_soffset_range.hip: voffset + soffset >=
    // num_records reads zeros) -- the pass had one v_add per load and eight more per group of eight (17 vector instructions per 24
    // MFMAs; on gfx950 every one of them is matrix time, profiles/r05_a_gather_gemm_limits.md section 11)
    const unsigned off = (unsigned)(((k0 + kk) * W + 32 * wave + li) * 12);
    const int dstep = 2 * W * 12;
    int soff = 0;
    const float* ta = tz + kPad + half + k0 + kk - r0 - li;            // + 2s
    floatx3 pf[kDepth];
#pragma unroll
    for (int j = 0; j < kDepth; ++j) {
      pf[j] = __builtin_bit_cast(floatx3, __builtin_amdgcn_raw_buffer_load_b96(rs, off, soff, 0));
      soff += dstep;
    }
    // the tap table goes up while the first eight loads are in flight
    for (int j = tid; j < T + 2 * kPad; j += blockDim.x) tz[j] = (j >= kPad && j < kPad + T) ? p.taps[j - kPad] : 0.f;
    __syncthreads();
    float a_cur = ta[0];
    for (int g = 0; g < NG; ++g) {
#pragma unroll
      for (int j = 0; j < kDepth; ++j) {
        const float a = a_cur;
        a_cur = ta[2 * (g * kDepth + j + 1)];                          // next k-pair's Toeplitz fragment (zero past the band)
        const floatx3 b = pf[j];
        __builtin_amdgcn_sched_barrier(0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b.x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b.y, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b.z, acc[2], 0, 0, 0);
#if !(defined(BG_DIAG) && defined(PANEL_NO_LOADS))
        pf[j] = __builtin_bit_cast(floatx3, __builtin_amdgcn_raw_buffer_load_b96(rs, off, soff, 0));
#endif
        soff += dstep;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // accumulator register i of lane (li, kk): row (i & 3) + 8 (i >> 2) + 4 kk, column (pixel 32 wave + li, channel c)
    float* yo = Ys + (4 * kk) * pitch + (32 * wave + li) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) yo[((i & 3) + 8 * (i >> 2)) * pitch + c] = acc[c][i];
  }
  __syncthreads();

  // ---------------------------------------------------------------- pass 2: Z^T = T_W * Y^T per channel, operands from LDS
  {
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    const int n0 = 32 * wave;
    const int c0 = max(0, n0 - half) & ~1, c1 = min(W, n0 + 32 + half);
#if defined(BG_DIAG) && defined(PANEL_NO_P2)
    const int K2 = 0;
    (void)c1;
#else
    const int K2 = (c1 - c0 + 1) >> 1;
#endif
    const float* ta = tz + kPad + half + c0 + kk - n0 - li;            // + 2s : T_W[x = n0 + li][x' = c0 + 2s + kk]
    const float* yb = Ys + li * pitch + (c0 + kk) * 3;                 // + 6s (+ c): Y[r = li][x', c]
    // two register sets, filled one k-pair ahead of the MFMAs that read them (no rotation moves, no multiply in the loop: a
    // first version copied the prefetched set into the current one behind an lgkmcnt(0) at the end of every iteration and took
    // 34 us for the 26 us of MFMAs this pass issues at the clock it runs at)
    float aA = ta[0], bA0 = yb[0], bA1 = yb[1], bA2 = yb[2], aB = 0.f, bB0 = 0.f, bB1 = 0.f, bB2 = 0.f;
    int s = 0;
    // four k-pairs per trip while they last (the pointer bumps and the tail select are vector instructions: two trips' worth per 12
    // MFMAs instead of per 6), then the two-pair loop below takes what is left with the same register protocol
    for (; s + 4 <= K2; s += 4) {
      aB = ta[2]; bB0 = yb[6]; bB1 = yb[7]; bB2 = yb[8];                 // pair s + 1
      __builtin_amdgcn_sched_barrier(0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(aA, bA0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(aA, bA1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(aA, bA2, acc[2], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      aA = ta[4]; bA0 = yb[12]; bA1 = yb[13]; bA2 = yb[14];              // pair s + 2
      __builtin_amdgcn_sched_barrier(0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(aB, bB0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(aB, bB1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(aB, bB2, acc[2], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      aB = ta[6]; bB0 = yb[18]; bB1 = yb[19]; bB2 = yb[20];              // pair s + 3 (exists: s + 4 <= K2)
      __builtin_amdgcn_sched_barrier(0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(aA, bA0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(aA, bA1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(aA, bA2, acc[2], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      const int adv4 = s + 4 < K2 ? 24 : 18;                             // pair s + 4, or pair s + 3 again when it is the last
      aA = ta[8]; bA0 = yb[adv4]; bA1 = yb[adv4 + 1]; bA2 = yb[adv4 + 2];
      __builtin_amdgcn_sched_barrier(0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(aB, bB0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(aB, bB1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(aB, bB2, acc[2], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      ta += 8;
      yb += 24;
    }
    for (; s + 2 <= K2; s += 2) {
      aB = ta[2]; bB0 = yb[6]; bB1 = yb[7]; bB2 = yb[8];                 // pair s + 1 (exists: s + 2 <= K2)
      __builtin_amdgcn_sched_barrier(0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(aA, bA0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(aA, bA1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(aA, bA2, acc[2], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      const int adv = s + 2 < K2 ? 12 : 6;                              // pair s + 2, or pair s + 1 again when it is the last
      aA = ta[4]; bA0 = yb[adv]; bA1 = yb[adv + 1]; bA2 = yb[adv + 2];  // the tap table reads zeros past the band
      __builtin_amdgcn_sched_barrier(0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(aB, bB0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(aB, bB1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(aB, bB2, acc[2], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      ta += 4;
      yb += 12;
    }
    if (s < K2) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(aA, bA0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(aA, bA1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(aA, bA2, acc[2], 0, 0, 0);
    }
    // register i of lane (li, kk): pixel n0 + (i & 3) + 8 (i >> 2) + 4 kk of output row r0 + li: 4 pixels x RGB = 12 contiguous
    // floats.  Stored straight from there every float4 store instruction touched 64 different cache lines (a lane = a row, rows
    // 3 KB apart): 17 of the kernel's 90 us at 143 taps.  So the wave's 32 x 96 tile goes through a PRIVATE LDS slab, 16 rows at
    // a time, and leaves with consecutive lanes on consecutive 16 bytes of one row (24 lanes = the tile's 384 contiguous bytes).
    float* stage = tz + ((T + 2 * kPad + 3) & ~3) + wave * (16 * kStagePitch);
    float* zt = p.y + ((size_t)img * H + r0) * Q + (size_t)n0 * 3;
#pragma unroll
    for (int hrow = 0; hrow < 2; ++hrow) {
      if ((li >> 4) == hrow) {
        float* so = stage + (li & 15) * kStagePitch + 4 * kk * 3;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float4 v0, v1, v2;
          v0.x = acc[0][4 * g + 0]; v0.y = acc[1][4 * g + 0]; v0.z = acc[2][4 * g + 0]; v0.w = acc[0][4 * g + 1];
          v1.x = acc[1][4 * g + 1]; v1.y = acc[2][4 * g + 1]; v1.z = acc[0][4 * g + 2]; v1.w = acc[1][4 * g + 2];
          v2.x = acc[2][4 * g + 2]; v2.y = acc[0][4 * g + 3]; v2.z = acc[1][4 * g + 3]; v2.w = acc[2][4 * g + 3];
          float4* d = reinterpret_cast<float4*>(so + 8 * g * 3);
          d[0] = v0; d[1] = v1; d[2] = v2;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int it = 0; it < 6; ++it) {                       // 16 rows x 24 float4 = 384 = 6 per lane
        const int e = it * 64 + lane, row = e / 24, f4 = e - row * 24;
        const float4 v = *reinterpret_cast<const float4*>(stage + row * kStagePitch + 4 * f4);
#if defined(BG_DIAG) && defined(PANEL_NO_STORE)
        if (v.x == 123.456f)                                  // never true: keeps the data flow, drops the store traffic
#endif
        *reinterpret_cast<float4*>(zt + (size_t)(16 * hrow + row) * Q + 4 * f4) = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// The same two passes over panels of SIXTEEN rows on v_mfma_f32_16x16x4_f32 (round 5).  A 32-row panel contracts over 32 + T - 1
// rows for 32 outputs -- 18 % of its MFMAs at 143 taps multiply zero taps -- and keeps 98 KB of Y, one workgroup per CU.  Sixteen
// rows contract over 16 + T - 1 (9 % / 6 % fewer MFMA cycles at 143 / 255 taps), keep 49 KB, and with an 8-row output slab per wave
// two workgroups fit a CU: the pass-1 loads, the barrier, and the copy-out of one hide under the other's MFMAs.  The instruction
// mix per MFMA cycle is the 32-row kernel's: per k-QUAD (4 rows) two buffer_load_dwordx3 per lane (pixels 16h + li of the wave's
// 32, lanes 16 kq .. 16 kq + 15 = row kq of the quad) feed six 16x16x4 MFMAs (2 pixel halves x RGB), as one load fed three
// 32x32x2 there; the step's row offset travels in the scalar offset.
//   pass 1:  Y[16][W*3] = T_H[16 x K] * X[K][W*3]            A = Toeplitz (row li, k = kq), B = X straight from global memory
//   pass 2:  Z^T[x][r]  = sum_x' T_W[x][x'] * Y[r][x', c]     A = Toeplitz (pixel li of a 16-pixel tile), B = Y^T from LDS; a lane
//            ends with 4 consecutive pixels x RGB of one output row, the wave's tile leaves through an 8-row LDS slab.
typedef float floatx4 __attribute__((ext_vector_type(4)));
constexpr int kDepth16 = 8;         // pass-1 k-quads in flight per wave (16 loads)

struct Panel16Params {
  const float* x;
  float* y;
  const float* taps;
  int B, H, W, T, nrb;
  unsigned char order[32];          // row blocks, longest band first
};

__global__ __launch_bounds__(512, 2) void blur_panel16_kernel(const Panel16Params p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int W = p.W, H = p.H, Q = 3 * W, pitch = Q + kPitchPad, T = p.T, half = T >> 1;
  float* Ys = lds;                                  // [16][pitch]
  float* tz = lds + 16 * pitch;                     // [kPad zeros][T taps][kPad zeros]
  int img, rb;
  {
    const int nrb = p.nrb, w = blockIdx.x;
    if ((p.B & 7) == 0) {                           // the panels of an image on one XCD (see blur_panel_kernel)
      const int xcd = w & 7, j = w >> 3, il = j / nrb, k = j - il * nrb;
      img = il * 8 + xcd;
      rb = p.order[(il & 4) ? nrb - 1 - k : k];
    } else {
      img = w / nrb;
      rb = p.order[w - img * nrb];
    }
  }
  const int r0 = 16 * rb;
  floatx4 acc[2][3];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[h][c] = floatx4{0.f, 0.f, 0.f, 0.f};

  // ---------------------------------------------------------------- pass 1
  {
    const int k0 = max(0, r0 - half) & ~3, k1 = min(H, r0 + 16 + half);
    const int K1 = (k1 - k0 + 3) >> 2, NG = (K1 + kDepth16 - 1) / kDepth16;
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (size_t)img * H * Q), 0, H * Q * 4, 0x00020000);
    // lane (li, kq): row k0 + 4s + kq, pixels 32 wave + 16 h + li; rows >= H lie past the descriptor's range (voffset + soffset): zeros
    const unsigned off0 = (unsigned)(((k0 + kq) * W + 32 * wave + li) * 12), off1 = off0 + 16 * 12;
    const int dstep = 4 * W * 12;
    int soff = 0;
    const float* ta = tz + kPad + half + k0 + kq - r0 - li;            // + 4s
    floatx3 pf[kDepth16][2];
#pragma unroll
    for (int j = 0; j < kDepth16; ++j) {
      pf[j][0] = __builtin_bit_cast(floatx3, __builtin_amdgcn_raw_buffer_load_b96(rs, off0, soff, 0));
      pf[j][1] = __builtin_bit_cast(floatx3, __builtin_amdgcn_raw_buffer_load_b96(rs, off1, soff, 0));
      soff += dstep;
    }
    for (int j = tid; j < T + 2 * kPad; j += blockDim.x) tz[j] = (j >= kPad && j < kPad + T) ? p.taps[j - kPad] : 0.f;
    __syncthreads();
    float a_cur = ta[0];
    for (int g = 0; g < NG; ++g) {
#pragma unroll
      for (int j = 0; j < kDepth16; ++j) {
        const float a = a_cur;
        a_cur = ta[4 * (g * kDepth16 + j + 1)];                        // next k-quad's Toeplitz fragment (zero past the band)
        const floatx3 b0 = pf[j][0], b1 = pf[j][1];
        __builtin_amdgcn_sched_barrier(0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0.x, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0.y, acc[0][1], 0, 0, 0);
        acc[0][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0.z, acc[0][2], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1.x, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1.y, acc[1][1], 0, 0, 0);
        acc[1][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1.z, acc[1][2], 0, 0, 0);
        pf[j][0] = __builtin_bit_cast(floatx3, __builtin_amdgcn_raw_buffer_load_b96(rs, off0, soff, 0));
        pf[j][1] = __builtin_bit_cast(floatx3, __builtin_amdgcn_raw_buffer_load_b96(rs, off1, soff, 0));
        soff += dstep;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // accumulator register i of lane (li, kq): row 4 kq + i, column (pixel 32 wave + 16 h + li, channel c)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float* yo = Ys + (4 * kq) * pitch + (32 * wave + 16 * h + li) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i) yo[i * pitch + c] = acc[h][c][i];
    }
  }
  __syncthreads();

  // ---------------------------------------------------------------- pass 2: two 16-pixel tiles per wave, one after the other
  float* stage = tz + ((T + 2 * kPad + 3) & ~3) + wave * (8 * kStagePitch);
  float* zt = p.y + ((size_t)img * H + r0) * Q + (size_t)(32 * wave) * 3;
  floatx4 z[2][3];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
#pragma unroll
    for (int c = 0; c < 3; ++c) z[m][c] = floatx4{0.f, 0.f, 0.f, 0.f};
    const int n0 = 32 * wave + 16 * m;
    const int c0 = max(0, n0 - half) & ~3, c1 = min(W, n0 + 16 + half);
    const int K2 = (c1 - c0 + 3) >> 2;                                 // k-quads
    const float* ta = tz + kPad + half + c0 + kq - n0 - li;            // + 4s : T_W[x = n0 + li][x' = c0 + 4s + kq]
    const float* yb = Ys + li * pitch + (c0 + kq) * 3;                 // + 12s (+ c): Y[r = li][x', c]; columns >= W meet zero taps
    float aA = ta[0], bA0 = yb[0], bA1 = yb[1], bA2 = yb[2], aB = 0.f, bB0 = 0.f, bB1 = 0.f, bB2 = 0.f;
    int s = 0;
    for (; s + 2 <= K2; s += 2) {
      aB = ta[4]; bB0 = yb[12]; bB1 = yb[13]; bB2 = yb[14];             // quad s + 1 (exists: s + 2 <= K2)
      __builtin_amdgcn_sched_barrier(0);
      z[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(aA, bA0, z[m][0], 0, 0, 0);
      z[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(aA, bA1, z[m][1], 0, 0, 0);
      z[m][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(aA, bA2, z[m][2], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      const int adv = s + 2 < K2 ? 24 : 12;                              // quad s + 2, or quad s + 1 again when it is the last
      aA = ta[8]; bA0 = yb[adv]; bA1 = yb[adv + 1]; bA2 = yb[adv + 2];
      __builtin_amdgcn_sched_barrier(0);
      z[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(aB, bB0, z[m][0], 0, 0, 0);
      z[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(aB, bB1, z[m][1], 0, 0, 0);
      z[m][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(aB, bB2, z[m][2], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      ta += 8;
      yb += 24;
    }
    if (s < K2) {
      z[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(aA, bA0, z[m][0], 0, 0, 0);
      z[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(aA, bA1, z[m][1], 0, 0, 0);
      z[m][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(aA, bA2, z[m][2], 0, 0, 0);
    }
  }
  // register i of lane (li, kq), tile m: pixel 32 wave + 16 m + 4 kq + i of output row r0 + li: 4 pixels x RGB = 12 contiguous floats.
  // Through the wave's private 8-row slab, then out as consecutive float4 of one row (see blur_panel_kernel).
#pragma unroll
  for (int hrow = 0; hrow < 2; ++hrow) {
    if ((li >> 3) == hrow) {
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        float* so = stage + (li & 7) * kStagePitch + (16 * m + 4 * kq) * 3;
        float4 v0, v1, v2;
        v0.x = z[m][0][0]; v0.y = z[m][1][0]; v0.z = z[m][2][0]; v0.w = z[m][0][1];
        v1.x = z[m][1][1]; v1.y = z[m][2][1]; v1.z = z[m][0][2]; v1.w = z[m][1][2];
        v2.x = z[m][2][2]; v2.y = z[m][0][3]; v2.z = z[m][1][3]; v2.w = z[m][2][3];
        float4* d = reinterpret_cast<float4*>(so);
        d[0] = v0; d[1] = v1; d[2] = v2;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int it = 0; it < 3; ++it) {                         // 8 rows x 24 float4 = 192 = 3 per lane
      const int e = it * 64 + lane, row = e / 24, f4 = e - row * 24;
      const float4 v = *reinterpret_cast<const float4*>(stage + row * kStagePitch + 4 * f4);
      *reinterpret_cast<float4*>(zt + (size_t)(8 * hrow + row) * Q + 4 * f4) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

void band_ranges16(int n, int half, int rb, int* lo, int* hi) {
  *lo = std::max(0, 16 * rb - half) & ~3;
  *hi = std::min(n, 16 * rb + 16 + half);
}

void band_ranges(int n, int half, int rb, int* lo, int* hi) {
  *lo = std::max(0, 32 * rb - half) & ~1;
  *hi = std::min(n, 32 * rb + 32 + half);
}

}  // namespace

namespace bg {

bool blur_panel_ok(int B, int H, int W, int C, int n_taps) {
  if (C != 3 || W % 32 || H % 32 || W > 256 || H > 512 || W < 32 || B <= 0) return false;
  if (n_taps < 3 || !(n_taps & 1) || n_taps > 1023) return false;
  return blur_panel_lds_bytes(W, n_taps) <= 160 * 1024 - 512;
}

static size_t panel16_lds_bytes(int W, int n_taps) {
  return ((size_t)16 * (3 * W + kPitchPad) + ((n_taps + 2 * kPad + 3) & ~3) + (size_t)(W / 32) * 8 * kStagePitch + 4) * sizeof(float);
}
// 16-row panels (two workgroups per CU): BG_BLUR_PANEL16=0 keeps the 32-row kernel.  Twice the panels read 1.8x the rows through
// L2 in pass 1 (16 x (16 + T - 1) against 8 x (32 + T - 1) per 256-row image), and from about 200 taps on that costs what the
// shorter band saves: 64 x 256x256x3 at 73 / 101 / 143 / 163 / 203 / 223 / 255 taps: -7.6 / -5.5 / -1.8 / -2.5 / 0 / +3 / +1 % (same
// box, gpurun_out/r05_aj, r05_ak); 128- and 192-pixel images -3 ... -6 %.  (A third form -- 32-row panels on 16x16x4 tiles whose two
// row halves contract over their own bands, operand quads in groups of four with a wave-uniform kind per group: the shorter band
// WITHOUT the extra L2 traffic -- was built, passed every parity case and ran 3-6 % SLOWER than the 32x32x2 kernel at one
// workgroup per CU (r05_ak): what the 16-row kernel gains it gains through its second resident workgroup.  Removed.)
static bool panel16_on(int H, int n_taps) {
  const char* e = getenv("BG_BLUR_PANEL16");                      // test aid, read per call (part of the step-program key, wgan.py)
  const int sw = e ? atoi(e) : 1;
  static const int max_taps = getenv("BG_BLUR_PANEL16_MAX_TAPS") ? atoi(getenv("BG_BLUR_PANEL16_MAX_TAPS")) : 208;
  return sw != 0 && n_taps <= max_taps && H % 16 == 0 && H / 16 <= 32;
}

size_t blur_panel_lds_bytes(int W, int n_taps) {
  // the pass-1 result, the tap table, one 16-row staging slab per wave for the coalesced copy-out
  return ((size_t)32 * (3 * W + kPitchPad) + ((n_taps + 2 * kPad + 3) & ~3) + (size_t)(W / 32) * 16 * kStagePitch + 4) * sizeof(float);
}

// MFMA flops the launch issues: 3 tiles x (k-pairs of pass 1, padded to groups of kDepth, + k-pairs of pass 2) per wave
double blur_panel_exec_flops(int B, int H, int W, int n_taps) {
  const int half = n_taps >> 1;
  if (panel16_on(H, n_taps) && panel16_lds_bytes(W, n_taps) <= 160 * 1024 - 512) {
    // k-quads of 16x16x4 MFMAs: pass 1 six per quad and wave (padded to groups of kDepth16), pass 2 three per quad and 16-pixel tile
    double mf = 0;
    for (int rb = 0; rb < H / 16; ++rb) {
      int lo, hi;
      band_ranges16(H, half, rb, &lo, &hi);
      const int K1 = (hi - lo + 3) / 4;
      mf += 6.0 * ((K1 + kDepth16 - 1) / kDepth16 * kDepth16) * (W / 32);
      for (int j = 0; j < W / 16; ++j) {
        band_ranges16(W, half, j, &lo, &hi);
        mf += 3.0 * ((hi - lo + 3) / 4);
      }
    }
    return (double)B * mf * 2.0 * 16 * 16 * 4;
  }
  double pairs = 0;
  for (int rb = 0; rb < H / 32; ++rb) {
    int lo, hi;
    band_ranges(H, half, rb, &lo, &hi);
    const int K1 = (hi - lo + 1) / 2;
    pairs += (double)((K1 + kDepth - 1) / kDepth * kDepth) * (W / 32);
    for (int j = 0; j < W / 32; ++j) {
      band_ranges(W, half, j, &lo, &hi);
      pairs += (hi - lo + 1) / 2;
    }
  }
  return (double)B * pairs * 3.0 * 2.0 * 32 * 32 * 2;
}

int blur_panel_launch(const float* x, float* y, int B, int H, int W, const float* taps_d, int n_taps, hipStream_t s) {
  if (panel16_on(H, n_taps) && panel16_lds_bytes(W, n_taps) <= 160 * 1024 - 512) {
    Panel16Params q;
    memset(&q, 0, sizeof q);
    q.x = x; q.y = y; q.taps = taps_d; q.B = B; q.H = H; q.W = W; q.T = n_taps; q.nrb = H / 16;
    int idx[32], cost[32];
    for (int rb = 0; rb < q.nrb; ++rb) {
      int lo, hi;
      band_ranges16(H, n_taps >> 1, rb, &lo, &hi);
      idx[rb] = rb;
      cost[rb] = hi - lo;
    }
    std::stable_sort(idx, idx + q.nrb, [&](int a, int b) { return cost[a] > cost[b]; });
    for (int i = 0; i < q.nrb; ++i) q.order[i] = (unsigned char)idx[i];
    BG_LDS_ATTR_ONCE(blur_panel16_kernel, 160 * 1024, "blur_panel16");
    bg::launch(blur_panel16_kernel, dim3((unsigned)(B * q.nrb)), dim3((unsigned)(64 * (W / 32))), panel16_lds_bytes(W, n_taps), s, q);
    return BG_OK;
  }
  PanelParams p;
  memset(&p, 0, sizeof p);
  p.x = x; p.y = y; p.taps = taps_d; p.B = B; p.H = H; p.W = W; p.T = n_taps; p.nrb = H / 32;
  int idx[16], cost[16];
  for (int rb = 0; rb < p.nrb; ++rb) {
    int lo, hi;
    band_ranges(H, n_taps >> 1, rb, &lo, &hi);
    idx[rb] = rb;
    cost[rb] = hi - lo;
  }
  std::stable_sort(idx, idx + p.nrb, [&](int a, int b) { return cost[a] > cost[b]; });
  for (int i = 0; i < p.nrb; ++i) p.order[i] = (unsigned char)idx[i];
  const size_t lds = blur_panel_lds_bytes(W, n_taps);
  BG_LDS_ATTR_ONCE(blur_panel_kernel, 160 * 1024, "blur_panel");
  bg::launch(blur_panel_kernel, dim3((unsigned)(B * p.nrb)), dim3((unsigned)(64 * (W / 32))), lds, s, p);
  return BG_OK;
}

}  // namespace bg
