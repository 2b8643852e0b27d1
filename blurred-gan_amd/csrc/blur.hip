// Separable Gaussian blur, NHWC float32 -- replaces the two tf.nn.depthwise_conv2d calls of
// reference gaussian_blur.py:116-130 (pass 1 along H, pass 2 along W, SAME zero padding) and the
// host-side sigma policy of gaussian_blur.py:15-88.
//
// HBM-bound op: algorithmic traffic is 8*H*W*C bytes per image (read once, write once).
//   blur_fused_kernel  one workgroup per image, whole image resident in LDS, both passes fused
//                      (traffic == algorithmic); used when 2 images fit in 160 KiB of LDS.
//   blur_pass_kernel   generic one-axis pass through L1/L2 with a scratch image in HBM
//                      (traffic 2x algorithmic); any size / tap count.
#include "common.h"
#include "blur_panel.h"
#include <mutex>
#include <cmath>
#include <cstdlib>
#include <type_traits>

// Diagnostic instrumentation (in-kernel s_memtime / HW_ID stamps read by tools/blur_stamps.py and tools/band_placement.py) is
// compiled only into BG_DIAG builds (tools/build_variant.sh <name> blur.hip -DBG_DIAG -DBLUR_STRIP_STAMP | -DBLUR_BAND_STAMP); the
// product build sees none of it, whatever else is on the command line.  The knock-out variants of round 2 (no loads / no stores)
// are gone from this file: profiles/r02_b_blur_notes.md and r02_e_band_notes.md keep what they measured.
#ifndef BG_DIAG
#undef BLUR_STRIP_STAMP
#undef BLUR_BAND_STAMP
#endif

namespace {

constexpr int kBlurThreads = 256;
constexpr int kFusedThreads = 1024;   // 16 waves per CU: the fused kernel is LDS-latency bound, not bandwidth bound
constexpr int kR = 4;                 // outputs per thread along the filtered axis (register sliding window)

// Both passes: each thread produces kR consecutive outputs along the filtered axis from one sweep over the
// T + kR - 1 source samples they share -- 2 LDS reads (sample + tap broadcast) per kR FMAs.
__global__ __launch_bounds__(kFusedThreads) void blur_fused_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                   int H, int W, int C,
                                                                   const float* __restrict__ taps, int T) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int WC = W * C, n = H * WC, half = T >> 1;
  const int npad = (n + 3) & ~3;
  float* s0 = lds;
  float* s1 = lds + npad;
  float* tp = lds + 2 * npad;            // taps, followed by kR zeros
  const float* xi = x + (size_t)blockIdx.x * n;
  float* yi = y + (size_t)blockIdx.x * n;
  const int tid = threadIdx.x;
  for (int j = tid; j < T + kR; j += kFusedThreads) tp[j] = j < T ? taps[j] : 0.f;
  if ((n & 3) == 0) {
    for (int e = tid * 4; e < n; e += kFusedThreads * 4) *reinterpret_cast<float4*>(s0 + e) = *reinterpret_cast<const float4*>(xi + e);
  } else {
    for (int e = tid; e < n; e += kFusedThreads) s0[e] = xi[e];
  }
  __syncthreads();
  // pass 1: along H (gaussian_blur.py:116-122); item = (column element e0, group of kR rows)
  const int HQ = (H + kR - 1) / kR;
  for (int item = tid; item < WC * HQ; item += kFusedThreads) {
    const int hq = item / WC, e0 = item - hq * WC;
    const int h0 = hq * kR;
    float acc[kR], tw[kR];
#pragma unroll
    for (int r = 0; r < kR; ++r) { acc[r] = 0.f; tw[r] = 0.f; }
    for (int j = 0; j < T + kR - 1; ++j) {
#pragma unroll
      for (int r = kR - 1; r > 0; --r) tw[r] = tw[r - 1];
      tw[0] = tp[j];
      const int srow = h0 - half + j;
      const float v = (unsigned)srow < (unsigned)H ? s0[srow * WC + e0] : 0.f;
#pragma unroll
      for (int r = 0; r < kR; ++r) acc[r] = fmaf(tw[r], v, acc[r]);   // output h0+r uses tap j-r
    }
#pragma unroll
    for (int r = 0; r < kR; ++r)
      if (h0 + r < H) s1[(h0 + r) * WC + e0] = acc[r];
  }
  __syncthreads();
  // pass 2: along W (gaussian_blur.py:124-130); item = (row h, group of kR columns, channel c)
  const int WQ = (W + kR - 1) / kR;
  for (int item = tid; item < H * WQ * C; item += kFusedThreads) {
    const int c = item % C;
    const int t2 = item / C;
    const int wq = t2 % WQ, h = t2 / WQ;
    const int w0 = wq * kR;
    const float* row = s1 + h * WC + c;
    float acc[kR], tw[kR];
#pragma unroll
    for (int r = 0; r < kR; ++r) { acc[r] = 0.f; tw[r] = 0.f; }
    for (int j = 0; j < T + kR - 1; ++j) {
#pragma unroll
      for (int r = kR - 1; r > 0; --r) tw[r] = tw[r - 1];
      tw[0] = tp[j];
      const int scol = w0 - half + j;
      const float v = (unsigned)scol < (unsigned)W ? row[scol * C] : 0.f;
#pragma unroll
      for (int r = 0; r < kR; ++r) acc[r] = fmaf(tw[r], v, acc[r]);
    }
#pragma unroll
    for (int r = 0; r < kR; ++r)
      if (w0 + r < W) yi[(h * W + w0 + r) * C + c] = acc[r];
  }
}

// ------------------------------------------------------------------------------------------------
// Matrix-core form for images up to 64 x 64 (every blur of the 64-pixel models): a 1-D blur with SAME zero padding is a
// banded Toeplitz matrix,  Y_c = T_H X_c  then  Z_c = Y_c T_W^T  per channel plane, with T[a][b] = taps[b - a + half].
// At 64 pixels the band (31 taps at sigma 5, 143 at sigma 23.5) covers half to all of the matrix, so the dense product on
// v_mfma_f32_32x32x2_f32 (exact fp32 products) costs 2 * 2*64^3 flop per plane REGARDLESS of the tap count and beats the
// sliding-window VALU kernel from ~13 taps up; the cost no longer grows with sigma (the reference starts training at
// sigma 23.5).  One workgroup per image: planes de-interleaved into LDS (row stride odd -> the transposed fragment reads of
// the W pass are conflict-free), Toeplitz fragments read from a zero-padded copy of the taps, result re-interleaved on the
// way out.  HBM traffic == algorithmic (8*H*W*C bytes per image).
// ------------------------------------------------------------------------------------------------
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
constexpr int kTzPad = 64;

constexpr int kMfmaBlurThreads = 1024;   // 16 waves: the load / de-interleave and re-interleave / store phases are latency-bound

struct FastDiv { unsigned mul, sh; };    // floor(v / d) for v < 2^20: (v * mul) >> sh
__device__ inline int fdiv(int v, FastDiv f) { return (int)(((unsigned long long)(unsigned)v * f.mul) >> f.sh); }

__global__ __launch_bounds__(kMfmaBlurThreads) void blur_mfma_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C,
                                                                     int Hp, int Wp, const float* __restrict__ taps, int T, FastDiv dC, FastDiv dW) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NTH = kMfmaBlurThreads;
  const int SP = Wp + 1;                                     // plane row stride (odd)
  const int plane = Hp * SP;
  float* X = lds;                                            // [C][Hp][SP]  input planes, later the output planes
  float* Y = lds + C * plane;                                // [C][Hp][SP]  after the H pass
  float* tz = lds + 2 * C * plane;                           // [kTzPad zeros][T taps][kTzPad zeros]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = T >> 1, n = H * W * C;
  const float* xi = x + (size_t)blockIdx.x * n;
  float* yi = y + (size_t)blockIdx.x * n;
  auto plane_off = [&](int e) {                              // NHWC element -> offset in the de-interleaved planes
    const int pix = fdiv(e, dC), c = e - pix * C;
    const int yy = fdiv(pix, dW), xx = pix - yy * W;
    return c * plane + yy * SP + xx;
  };
  for (int j = tid; j < T + 2 * kTzPad; j += NTH) tz[j] = (j >= kTzPad && j < kTzPad + T) ? taps[j - kTzPad] : 0.f;
  if (Hp != H || Wp != W) {                                  // padding rows / columns must read as zero
    for (int e = tid; e < C * plane; e += NTH) X[e] = 0.f;
    __syncthreads();
  }
  if ((n & 3) == 0) {
    for (int e = tid * 4; e < n; e += NTH * 4) {
      const float4 v = *reinterpret_cast<const float4*>(xi + e);
      X[plane_off(e)] = v.x; X[plane_off(e + 1)] = v.y; X[plane_off(e + 2)] = v.z; X[plane_off(e + 3)] = v.w;
    }
  } else {
    for (int e = tid; e < n; e += NTH) X[plane_off(e)] = xi[e];
  }
  __syncthreads();
  const int mts = Hp / 32, nts = Wp / 32, ntiles = C * mts * nts;
  const int li = lane & 31, kk = lane >> 5;
  // ---- H pass: Y_c[y][x] = sum_y' taps[y' - y + half] * X_c[y'][x];  A = Toeplitz, B = X rows
  for (int tile = wave; tile < ntiles; tile += NTH / 64) {
    const int c = tile / (mts * nts), r = tile - c * mts * nts, mt = r / nts, nt = r - mt * nts;
    floatx16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    const float* ta = tz + kTzPad + half - (32 * mt + li) + kk;     // + k
    const float* xb = X + c * plane + kk * SP + 32 * nt + li;       // + k * SP
#pragma unroll 8
    for (int kp = 0; kp < Hp / 2; ++kp) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[2 * kp], xb[2 * kp * SP], acc, 0, 0, 0);
    float* yo = Y + c * plane + (32 * mt + 4 * kk) * SP + 32 * nt + li;
#pragma unroll
    for (int q = 0; q < 16; ++q) yo[((q & 3) + 8 * (q >> 2)) * SP] = acc[q];
  }
  __syncthreads();
  // ---- W pass: Z_c[y][x] = sum_x' Y_c[y][x'] * taps[x' - x + half];  A = Y rows (transposed read), B = Toeplitz
  for (int tile = wave; tile < ntiles; tile += NTH / 64) {
    const int c = tile / (mts * nts), r = tile - c * mts * nts, mt = r / nts, nt = r - mt * nts;
    floatx16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    const float* ya = Y + c * plane + (32 * mt + li) * SP + kk;     // + k
    const float* tb = tz + kTzPad + half - (32 * nt + li) + kk;     // + k
#pragma unroll 8
    for (int kp = 0; kp < Wp / 2; ++kp) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ya[2 * kp], tb[2 * kp], acc, 0, 0, 0);
    float* zo = X + c * plane + (32 * mt + 4 * kk) * SP + 32 * nt + li;   // X is dead: every wave passed the barrier above
#pragma unroll
    for (int q = 0; q < 16; ++q) zo[((q & 3) + 8 * (q >> 2)) * SP] = acc[q];
  }
  __syncthreads();
  if ((n & 3) == 0) {
    for (int e = tid * 4; e < n; e += NTH * 4)
      *reinterpret_cast<float4*>(yi + e) = make_float4(X[plane_off(e)], X[plane_off(e + 1)], X[plane_off(e + 2)], X[plane_off(e + 3)]);
  } else {
    for (int e = tid; e < n; e += NTH) yi[e] = X[plane_off(e)];
  }
}

// ------------------------------------------------------------------------------------------------
// blur_rows_kernel (round 3): the same two Toeplitz products on images up to 64 x 64 whose rows are float4-addressable
// (W*C % 4 == 0, C <= 4), with MORE THAN ONE workgroup per CU.  blur_mfma_kernel holds a whole image (two de-interleaved copies,
// 100 KB of LDS at 64x64x3) in one 16-wave workgroup: 256 images = one workgroup per CU, whose load, H-pass, W-pass and store
// phases run strictly one after the other (16.3 us for 25 MB, 0.19 of the HBM roof on the headline configuration).  Here a
// workgroup owns ONE BLOCK OF 32 OUTPUT ROWS of an image and everything stays in the NHWC-interleaved layout:
//   * it loads the rows within half a kernel of its block -- a contiguous piece of the image -- as straight float4 copies;
//   * H pass  Y[32][Q] = T_H[32 x rows] * X[rows][Q]  treats a row as Q = W*C independent columns: interleaving is irrelevant;
//   * W pass per channel  Z[y][x, c] = sum_x' Y[y][x', c] * t[x' - x + half]  reads Y with a stride of C floats along k and writes
//     Z with a stride of C floats along n -- both conflict-free for odd C (row pitch of Y odd, lanes of a half-wave on 32 banks);
//   * the 32 result rows leave as straight float4 copies.
// No de-interleaving index arithmetic, 62 KB of LDS at 64x64x3 / 31 taps -> two 8-wave workgroups per CU, 512 workgroups for 256
// images: one's loads and stores run under the other's MFMA chains.
// ------------------------------------------------------------------------------------------------
constexpr int kRowsThreads = 512;
constexpr int kRowsBlock = 32;            // output rows per workgroup (one MFMA row block)

struct RowsGeom { int Q, Wp, Qp, pitchX, pitchY, pitchZ, xfloats, nb; size_t lds; };
// LDS pitches of the three tiles, chosen against the 64 banks for the lane patterns of the 16 x 16 x 4 passes: X is read with
// lanes (k = lane / 16, column = lane % 16) -> pitch = 16 (mod 64) puts the four k rows on four different bank quarters; Y is
// written with (row = 4 * (lane / 16) + i, column = lane % 16) and read with (row = lane % 16, x' = lane / 16, stride C) ->
// pitch = 4 (mod 64) makes both conflict-free for C <= 5; Z likewise and a multiple of 4 for the float4 copy out.
__host__ __device__ inline int rows_pitch(int q, int r) { return q + ((r - q) & 63); }
inline RowsGeom rows_geom(int H, int W, int C, int T) {
  RowsGeom g;
  const int half = T >> 1;
  g.Q = W * C;
  g.Wp = (W + 31) / 32 * 32;
  g.Qp = (g.Wp * C + 31) / 32 * 32;
  g.pitchX = rows_pitch(g.Qp, 16);
  g.pitchY = rows_pitch(g.Qp, 4);
  g.pitchZ = rows_pitch(g.Qp, 4);
  g.nb = (H + kRowsBlock - 1) / kRowsBlock;
  int rows = 0;
  for (int rb = 0; rb < g.nb; ++rb) rows = std::max(rows, std::min(H, rb * kRowsBlock + kRowsBlock + half) - std::max(0, rb * kRowsBlock - half));
  rows = (rows + 3) & ~3;                 // k-steps of 4 source rows (v_mfma_f32_16x16x4_f32)
  g.xfloats = (std::max(rows * g.pitchX, kRowsBlock * g.pitchZ) + 3) & ~3;
  g.lds = ((size_t)g.xfloats + (((size_t)kRowsBlock * g.pitchY + 3) & ~(size_t)3) + T + 2 * kTzPad) * sizeof(float);
  return g;
}

// n (wave-uniform) k-steps of v_mfma_f32_16x16x4_f32 with both operands from LDS: a[4 i sa], b[4 i sb].  The reads of a run are
// issued together and waited for once (a loop of read, wait, MFMA pays the LDS latency per step: 250 cycles per MFMA measured).
template <int N>
__device__ __forceinline__ floatx4 rows_steps(const float* a, int sa, const float* b, int sb, floatx4 acc) {
  float av[N], bv[N];
#pragma unroll
  for (int i = 0; i < N; ++i) { av[i] = a[4 * i * sa]; bv[i] = b[4 * i * sb]; }
#pragma unroll
  for (int i = 0; i < N; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[i], acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ floatx4 rows_chain(const float* a, int sa, const float* b, int sb, int n, floatx4 acc) {
  while (n >= 12) { acc = rows_steps<12>(a, sa, b, sb, acc); a += 48 * sa; b += 48 * sb; n -= 12; }
  if (n >= 8) { acc = rows_steps<8>(a, sa, b, sb, acc); a += 32 * sa; b += 32 * sb; n -= 8; }
  if (n >= 4) { acc = rows_steps<4>(a, sa, b, sb, acc); a += 16 * sa; b += 16 * sb; n -= 4; }
  if (n >= 2) { acc = rows_steps<2>(a, sa, b, sb, acc); a += 8 * sa; b += 8 * sb; n -= 2; }
  if (n >= 1) acc = rows_steps<1>(a, sa, b, sb, acc);
  return acc;
}

// Three-source form (x2 != nullptr; the critic's batch of wgan.py:138-139,239-240 in ONE launch): images
// [0, Bs) are blurred from x, [Bs, 2 Bs) from x2 and [2 Bs, 3 Bs) from x-hat = x2 + alpha[b] * (x - x2), formed while the rows are
// copied into LDS (the expression of lerp_kernel, so the result is bit-identical to lerp followed by blur) -- x-hat never exists
// in memory, and 3 Bs x nb workgroups pipeline through the CUs where three launches of Bs x nb each ran one after the other.
__global__ __launch_bounds__(kRowsThreads) void blur_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C,
                                                                 int nb, int Qp, int Wp, int xfloats, const float* __restrict__ taps, int T,
                                                                 FastDiv dQ4, const float* __restrict__ x2, const float* __restrict__ alpha,
                                                                 int Bs) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NTH = kRowsThreads, NW = NTH / 64;
  const int Q = W * C, q4 = Q >> 2, half = T >> 1;
  const int pitchX = rows_pitch(Qp, 16), pitchY = rows_pitch(Qp, 4), pitchZ = rows_pitch(Qp, 4);
  float* X = lds;                                             // [rows][pitchX] source rows; later Z [32][pitchZ]
  float* Y = lds + xfloats;                                   // [32][pitchY] after the H pass
  float* tz = Y + ((kRowsBlock * pitchY + 3) & ~3);           // [kTzPad zeros][T taps][kTzPad zeros]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int img = blockIdx.x / nb, rb = blockIdx.x - img * nb, r0 = rb * kRowsBlock;
  const int ks = max(0, r0 - half), ke = min(H, r0 + kRowsBlock + half);       // source rows [ks, ke)
  const int nk = ke - ks, nk4 = (nk + 3) & ~3;
  const int grp = x2 ? img / Bs : 0, bimg = x2 ? img - grp * Bs : img;          // source group and image within it
  const float* xi = (grp == 1 ? x2 : x) + (size_t)bimg * H * Q;
  float* yi = y + (size_t)img * H * Q;
  for (int j = tid; j < T + 2 * kTzPad; j += NTH) tz[j] = (j >= kTzPad && j < kTzPad + T) ? taps[j - kTzPad] : 0.f;
  // what the copy below does not write must read as zero: the columns past the row (when W*C is not a multiple of 32) and the
  // odd k of the last MFMA pair (when the block needs an odd number of source rows)
  if (Qp != Q) {
    const int padc = Qp - Q;
    for (int e = tid; e < nk * padc; e += NTH) X[(e / padc) * pitchX + Q + e % padc] = 0.f;
  }
  for (int e = tid; e < (nk4 - nk) * Qp; e += NTH) X[(nk + e / Qp) * pitchX + e % Qp] = 0.f;
  if (grp == 2) {                                              // x-hat rows: r + a * (f - r), r = x2, f = x (workgroup-uniform branch)
    const float4* sf = reinterpret_cast<const float4*>(xi + (size_t)ks * Q);
    const float4* sr = reinterpret_cast<const float4*>(x2 + (size_t)bimg * H * Q + (size_t)ks * Q);
    const float a = alpha[bimg];
    const int total4 = nk * q4;
    for (int i = tid; i < total4; i += NTH) {
      const int row = fdiv(i, dQ4), c4 = i - row * q4;
      const float4 f = sf[i], r = sr[i];
      float4 v;
      v.x = r.x + a * (f.x - r.x); v.y = r.y + a * (f.y - r.y); v.z = r.z + a * (f.z - r.z); v.w = r.w + a * (f.w - r.w);
      *reinterpret_cast<float4*>(X + row * pitchX + 4 * c4) = v;
    }
  } else {
    const float4* src = reinterpret_cast<const float4*>(xi + (size_t)ks * Q);
    const int total4 = nk * q4;
    for (int i = tid; i < total4; i += NTH) {
      const int row = fdiv(i, dQ4), c4 = i - row * q4;
#if defined(BG_DIAG) && defined(ROWS_NO_MEM)
      *reinterpret_cast<float4*>(X + row * pitchX + 4 * c4) = float4{0.5f, 0.25f, (float)i, 1.f};
      (void)src;
#else
      *reinterpret_cast<float4*>(X + row * pitchX + 4 * c4) = src[i];
#endif
    }
  }
  __syncthreads();
  // Both passes on v_mfma_f32_16x16x4_f32 over 16 x 16 tiles restricted to the BAND (round 4).  The 32 x 32 x 2 form gave a
  // workgroup 6 items per pass for its 8 waves (two SIMDs carried twice the chains of the other two) and contracted over all
  // source rows / all 64 columns although a tile only sees taps within half a kernel of it: with every global load and store
  // knocked out the kernel still took 11.5 of its 14.0 us at 256 x 64x64x3 / 31 taps -- the matrix pipe of the busiest SIMD was
  // the floor, not memory and not the launch.  16-wide tiles: 24 items per pass (3 per wave), 12 k-steps of 4 instead of 24 + 32 of 2.
  const int l16 = lane & 15, kq = lane >> 4;
  const int rts = (min(kRowsBlock, H - r0) + 15) >> 4;            // 16-row tiles of this block that hold output rows
  // ---- H pass: Y[m][q] = sum_k t[(ks + k) - (r0 + m) + half] * X[k][q];  A = Toeplitz [16 rows x 4 k], B = source rows [4 k x 16 columns]
  {
    const int nct = Qp >> 4;
#if defined(BG_DIAG) && defined(ROWS_NO_H)
    for (int item = wave; item < (int)(tz[0] != 0.f); item += NW) {
#else
    for (int item = wave; item < rts * nct; item += NW) {
#endif
      const int rt = item / nct, ct = item - rt * nct;
      const int klo = max(0, r0 + 16 * rt - half - ks) & ~3, khi = min(nk4, (r0 + 16 * rt + 16 + half - ks + 3) & ~3);
      floatx4 acc = {0.f, 0.f, 0.f, 0.f};
      const float* ta = tz + kTzPad + half + ks - (r0 + 16 * rt + l16) + kq;      // + k
      const float* xb = X + kq * pitchX + 16 * ct + l16;                            // + k * pitchX
      acc = rows_chain(ta + klo, 1, xb + klo * pitchX, pitchX, (khi - klo) >> 2, acc);
      float* yo = Y + (16 * rt + 4 * kq) * pitchY + 16 * ct + l16;
#pragma unroll
      for (int i = 0; i < 4; ++i) yo[i * pitchY] = acc[i];
    }
  }
  __syncthreads();
  // ---- W pass per channel: Z[m][x, c] = sum_x' Y[m][x', c] * t[x' - x + half];  A = Y (lanes along rows), B = Toeplitz
  float* Z = X;                                                     // X is dead: every wave passed the barrier above
  {
    const int xts = Wp >> 4;
#if defined(BG_DIAG) && defined(ROWS_NO_W)
    for (int item = wave; item < (int)(tz[0] != 0.f); item += NW) {
#else
    for (int item = wave; item < rts * C * xts; item += NW) {
#endif
      const int rt = item / (C * xts), rem = item - rt * (C * xts), c = rem / xts, xt = rem - c * xts;
      const int lo = max(0, 16 * xt - half) & ~3, hi = min(Wp, (16 * xt + 16 + half + 3) & ~3);
      floatx4 acc = {0.f, 0.f, 0.f, 0.f};
      const float* ya = Y + (16 * rt + l16) * pitchY + kq * C + c;  // + x' * C
      const float* tb = tz + kTzPad + half - (16 * xt + l16) + kq;  // + x'
      acc = rows_chain(ya + lo * C, C, tb + lo, 1, (hi - lo) >> 2, acc);
      float* zo = Z + (16 * rt + 4 * kq) * pitchZ + (16 * xt + l16) * C + c;
#pragma unroll
      for (int i = 0; i < 4; ++i) zo[i * pitchZ] = acc[i];
    }
  }
  __syncthreads();
  {
    float4* dst = reinterpret_cast<float4*>(yi + (size_t)r0 * Q);
    const int total4 = min(kRowsBlock, H - r0) * q4;
    for (int i = tid; i < total4; i += NTH) {
      const int row = fdiv(i, dQ4), c4 = i - row * q4;
      const float4 zv = *reinterpret_cast<const float4*>(Z + row * pitchZ + 4 * c4);
#if defined(BG_DIAG) && defined(ROWS_NO_MEM)
      if (zv.x == 123.456f)                                    // never true: keeps the data flow, drops the store traffic
#endif
      dst[i] = zv;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Images larger than 64 x 64: the same Toeplitz product, one direction per launch with a scratch image between them
// (traffic 2x algorithmic), restricted to the BAND: an output block of 32 rows only contracts over the source rows
// within half a kernel of it.
// ------------------------------------------------------------------------------------------------
#ifndef BLUR_ST_AUX          // cache policy of the fused (cols / strip) kernels' float4 stores
#define BLUR_ST_AUX 0
#endif
// cache policy of the band passes' float4 stores: 2 = nt (streamed; 97 against 100 us at 143 taps with the default policy)
#ifndef BAND_ST_AUX
#define BAND_ST_AUX 2
#endif
// One pass of the separable blur as  out^T = (T * in)^T : contraction along the LEADING dimension of in[R][S][C] (rows of
// Q = S*C floats, lanes along Q -> coalesced), result stored TRANSPOSED as out[S][R][C].  Run twice it blurs both directions
// and lands back in NHWC: x[H][W][C] -> tmp[W][H][C] -> y[H][W][C]; the W direction never needs the C-times-sparser
// Toeplitz matrix an interleaved row would ask for.
//
// Workgroup = 128 output rows x NW column tiles of 32 floats; a WAVE owns one column tile for all four 32-row blocks, so the
// source columns it contracts over are its own: every wave streams its 32 columns down the band through a PRIVATE
// double-buffered LDS chunk (32 rows x 32 floats) and transposes its own results -- after the taps table is up there is no
// barrier in the kernel (with a barrier per chunk and a chunk shared by the workgroup the matrix pipe was 65 % busy inside the
// loop, 82 % without: s_memtime stamps, tools/band_placement.py).
// Source rows are addressed relative to k0 = r0 - half in STAGES of 8 rows (4 MFMA k-pairs): row block jb contracts over the
// stages of rows [32 jb, 32 jb + 32 + T - 1), the same for every workgroup, and its Toeplitz fragment of stage s is
// tz[PadLo + 8 s + 2 u + kk - 32 jb - li] from a zero-padded table (partial last stage: zero taps; rows outside the image:
// zeros in the chunk, whole stages outside it skipped).  The operand reads of stage s + 1 (4 source + 16 Toeplitz values,
// ds_read2) are issued before the MFMAs of stage s into the other of two register sets; the next chunk's global loads fly
// for three stages before they are written to LDS.
// A row block leaves one chunk after its band is through (blocks 0..2 beside the MFMAs of the blocks below, only the last one
// in the tail): the wave lays its 32 x 32 tile out as [pixel column][row][channel] in the chunk buffer it is about to leave
// and stores whole pixel columns (32*C contiguous floats of the transposed image) as float4, a quarter per stage; with 3
// channels a 32-float tile starts and ends inside a pixel, and those two partial columns go out as single floats.
constexpr int kBtRows = 128, kBtPadLo = 128, kBtPadHi = 208;
constexpr int kBtWS = 40;             // chunk row stride: 32 x 40 floats also hold the transposed tile of a row block

// Geometry: the workgroup's column span is a whole number of pixels: 4 tiles = 128 / C pixels for 1, 2 and 4 channels; for 3
// channels THREE tiles = 96 floats = 32 pixels (192-thread workgroups) -- a 256-pixel line is 8 groups with no ragged last
// group, and 64 x 256x256x3 is 1024 workgroups = 4 per CU on every CU, 3 waves on every SIMD (placement read back from HW_ID
// in the -DBLUR_BAND_STAMP build: tools/band_placement.py).
template <int C>
struct BandCfg {
  static constexpr int NW = C == 3 ? 3 : 4;                   // waves = MFMA column tiles per workgroup
  static constexpr int NTH = NW * 64;
  static constexpr int COLS = NW * 32;                        // floats per workgroup row
  static constexpr int PXW = COLS / C;                        // whole pixels per workgroup
  static constexpr int RUN = 32 * C + 4;                      // one pixel column of a transposed 32-row block (+ pad)
  static constexpr int SLOTS = (32 + 2 * (C - 1)) / C;        // pixel columns a 32-float tile can touch
  static constexpr int BUF = NW * 2 * 32 * kBtWS;             // per-wave double-buffered source chunks (floats)
  static_assert(COLS % C == 0, "workgroup span must be whole pixels");
  static_assert(SLOTS * RUN <= 32 * kBtWS, "the transposed tile must fit a chunk buffer");
};
inline int band_tz_floats(int T) { return (T + kBtPadLo + kBtPadHi + 3) & ~3; }

#ifdef BLUR_BAND_STAMP          // diagnostic build only: where and when every wave of the last band pass ran
__device__ unsigned long long g_band_dbg[4096 * 4];
#endif

// The four 32 x 32 accumulators of a wave live in a[0:63] BY HAND: the MFMAs, the zeroing and the read-out are asm statements
// naming the registers (the compiler only learns that they are clobbered).  Through the builtin -- or through asm with
// allocated operands -- the compiler gives the arms of the band's control flow their own copies of the accumulators (128 AGPRs
// plus moves; capped at three waves per SIMD it spills them), and the band needs that control flow: stages that all four row
// blocks cover issue k-pair major, consecutive MFMAs going to different accumulators, the others block by block.  (A build
// with every memory instruction knocked out runs the MFMAs of a pass in 36 us that way against 50 with dependent chains of
// four throughout; in the full kernel the difference is 4 % at 255 taps and nothing at 143.)
// The compiler does not see MFMAs here: the wait states between the last MFMA and the VALU read of its result are in band_acc_read.
#define BAND_CL0 "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15"
#define BAND_CL1 "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31"
#define BAND_CL2 "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47"
#define BAND_CL3 "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63"
// EVERY band asm statement clobbers all 64 registers, whichever group it touches: a clobber does not reserve a register, it
// only keeps compiler values that are live ACROSS the statement out of it, so each statement has to fence the whole file.
// A value whose live range sits between two statements could still be placed in a[0:63]; tests/test_build_cpu.py compiles
// this file to ISA and asserts that nothing but these hand-written instructions names a0..a63 in any band kernel.
#define BAND_CL_ALL BAND_CL0, BAND_CL1, BAND_CL2, BAND_CL3
template <int JB>
__device__ __forceinline__ void band_mfma(float a, float b) {
  if constexpr (JB == 0) asm volatile("v_mfma_f32_32x32x2_f32 a[0:15], %0, %1, a[0:15]" ::"v"(a), "v"(b) : BAND_CL_ALL);
  if constexpr (JB == 1) asm volatile("v_mfma_f32_32x32x2_f32 a[16:31], %0, %1, a[16:31]" ::"v"(a), "v"(b) : BAND_CL_ALL);
  if constexpr (JB == 2) asm volatile("v_mfma_f32_32x32x2_f32 a[32:47], %0, %1, a[32:47]" ::"v"(a), "v"(b) : BAND_CL_ALL);
  if constexpr (JB == 3) asm volatile("v_mfma_f32_32x32x2_f32 a[48:63], %0, %1, a[48:63]" ::"v"(a), "v"(b) : BAND_CL_ALL);
}
__device__ __forceinline__ void band_acc_zero() {
  asm volatile(".irp r,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,36,37,38,39,40,41,42,43,44,45,46,47,48,49,50,51,52,53,54,55,56,57,58,59,60,61,62,63\n\tv_accvgpr_write_b32 a\\r, 0\n\t.endr\n\ts_nop 3" ::
               : BAND_CL0, BAND_CL1, BAND_CL2, BAND_CL3);
}
#define BAND_RD16_OUT "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7]), "=v"(v[8]), "=v"(v[9]), \
                      "=v"(v[10]), "=v"(v[11]), "=v"(v[12]), "=v"(v[13]), "=v"(v[14]), "=v"(v[15])
#define BAND_SETTLE "s_nop 15\n\ts_nop 7\n\t"          /* 16-pass MFMA -> VALU read of its result: 18 wait states */
__device__ __forceinline__ void band_acc_read(int jb, float (&v)[16]) {      // jb uniform
  switch (jb) {
    case 0: asm volatile(BAND_SETTLE "v_accvgpr_read_b32 %0, a0\n\t" "v_accvgpr_read_b32 %1, a1\n\t" "v_accvgpr_read_b32 %2, a2\n\t" "v_accvgpr_read_b32 %3, a3\n\t" "v_accvgpr_read_b32 %4, a4\n\t" "v_accvgpr_read_b32 %5, a5\n\t" "v_accvgpr_read_b32 %6, a6\n\t" "v_accvgpr_read_b32 %7, a7\n\t" "v_accvgpr_read_b32 %8, a8\n\t" "v_accvgpr_read_b32 %9, a9\n\t" "v_accvgpr_read_b32 %10, a10\n\t" "v_accvgpr_read_b32 %11, a11\n\t" "v_accvgpr_read_b32 %12, a12\n\t" "v_accvgpr_read_b32 %13, a13\n\t" "v_accvgpr_read_b32 %14, a14\n\t" "v_accvgpr_read_b32 %15, a15\n\t" : BAND_RD16_OUT : : BAND_CL_ALL); break;
    case 1: asm volatile(BAND_SETTLE "v_accvgpr_read_b32 %0, a16\n\t" "v_accvgpr_read_b32 %1, a17\n\t" "v_accvgpr_read_b32 %2, a18\n\t" "v_accvgpr_read_b32 %3, a19\n\t" "v_accvgpr_read_b32 %4, a20\n\t" "v_accvgpr_read_b32 %5, a21\n\t" "v_accvgpr_read_b32 %6, a22\n\t" "v_accvgpr_read_b32 %7, a23\n\t" "v_accvgpr_read_b32 %8, a24\n\t" "v_accvgpr_read_b32 %9, a25\n\t" "v_accvgpr_read_b32 %10, a26\n\t" "v_accvgpr_read_b32 %11, a27\n\t" "v_accvgpr_read_b32 %12, a28\n\t" "v_accvgpr_read_b32 %13, a29\n\t" "v_accvgpr_read_b32 %14, a30\n\t" "v_accvgpr_read_b32 %15, a31\n\t" : BAND_RD16_OUT : : BAND_CL_ALL); break;
    case 2: asm volatile(BAND_SETTLE "v_accvgpr_read_b32 %0, a32\n\t" "v_accvgpr_read_b32 %1, a33\n\t" "v_accvgpr_read_b32 %2, a34\n\t" "v_accvgpr_read_b32 %3, a35\n\t" "v_accvgpr_read_b32 %4, a36\n\t" "v_accvgpr_read_b32 %5, a37\n\t" "v_accvgpr_read_b32 %6, a38\n\t" "v_accvgpr_read_b32 %7, a39\n\t" "v_accvgpr_read_b32 %8, a40\n\t" "v_accvgpr_read_b32 %9, a41\n\t" "v_accvgpr_read_b32 %10, a42\n\t" "v_accvgpr_read_b32 %11, a43\n\t" "v_accvgpr_read_b32 %12, a44\n\t" "v_accvgpr_read_b32 %13, a45\n\t" "v_accvgpr_read_b32 %14, a46\n\t" "v_accvgpr_read_b32 %15, a47\n\t" : BAND_RD16_OUT : : BAND_CL_ALL); break;
    default: asm volatile(BAND_SETTLE "v_accvgpr_read_b32 %0, a48\n\t" "v_accvgpr_read_b32 %1, a49\n\t" "v_accvgpr_read_b32 %2, a50\n\t" "v_accvgpr_read_b32 %3, a51\n\t" "v_accvgpr_read_b32 %4, a52\n\t" "v_accvgpr_read_b32 %5, a53\n\t" "v_accvgpr_read_b32 %6, a54\n\t" "v_accvgpr_read_b32 %7, a55\n\t" "v_accvgpr_read_b32 %8, a56\n\t" "v_accvgpr_read_b32 %9, a57\n\t" "v_accvgpr_read_b32 %10, a58\n\t" "v_accvgpr_read_b32 %11, a59\n\t" "v_accvgpr_read_b32 %12, a60\n\t" "v_accvgpr_read_b32 %13, a61\n\t" "v_accvgpr_read_b32 %14, a62\n\t" "v_accvgpr_read_b32 %15, a63\n\t" : BAND_RD16_OUT : : BAND_CL_ALL); break;
  }
}

struct BandFrag { float a[4][4]; float b[4]; };               // operands of one stage: Toeplitz [row block][k-pair], source [k-pair]

// C: channel count as a compile-time constant (index divisions).  LD: how a chunk is loaded -- 2: float4 buffer loads (rows are
// float4-addressable), 1: dword buffer loads, 0: synchronous 64-bit addressing (images beyond the 32-bit byte offsets of a
// buffer descriptor)
template <int C, int LD>
__global__ __launch_bounds__(BandCfg<C>::NTH, 3) void blur_band_t_kernel(const float* __restrict__ x, float* __restrict__ y, int R, int S,
                                                                        int row_groups, int col_groups, const float* __restrict__ taps, int T) {
  using G = BandCfg<C>;
  constexpr int NTH = G::NTH, PXW = G::PXW, RUN = G::RUN, WS = kBtWS;
  constexpr bool V4 = LD == 2;
  extern __shared__ __attribute__((aligned(16))) float tl[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = T >> 1, Q = S * C;
#ifdef BLUR_BAND_STAMP
  const unsigned long long st0 = __builtin_amdgcn_s_memtime();
  unsigned long long st1 = st0;
#endif
  const int tzn = (T + kBtPadLo + kBtPadHi + 3) & ~3;
  float* tz = tl;                                              // [PadLo zeros][T][PadHi zeros]
  float* wb = tl + tzn + wave * 2 * 32 * WS;                   // this wave's two chunk buffers
  const int per_img = row_groups * col_groups;
  const int b = blockIdx.x / per_img, u_ = blockIdx.x - b * per_img;
  const int rg = u_ / col_groups, cg = u_ - rg * col_groups;
  const int r0 = rg * kBtRows, s0 = cg * PXW;
  const int qw = s0 * C + wave * 32;                           // this wave's first source column
  const int wcols = min(32, Q - qw);                           // <= 0: nothing to contract (ragged last group)
  const int li = lane & 31, kk = lane >> 5;
  const float* xi = x + (size_t)b * R * Q;
  float* yi = y + (size_t)b * R * Q;
  const int k0 = r0 - half, L = 32 + T - 1;
  const int k_end = min(R, r0 + kBtRows + half);               // source rows needed: [max(0, k0), k_end)
  const int ci_lo = k0 < 0 ? (-k0) >> 5 : 0, ci_hi = (k_end - k0 + 31) >> 5;         // chunks of 32 rows from k0
  const int lr = lane >> 3, c4 = lane & 7;                     // loader: 8 float4 per row, rows lr + 8 i
  float4 g[4];
  float gs[LD == 1 ? 16 : 1];                                  // LD == 1: element e = lane + 64 i of the chunk, row e / 32, column e % 32
  constexpr int kOob = (int)0x80000000;
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xi), 0, LD ? R * Q * 4 : 0, 0x00020000);
  auto gload = [&](int ci) {
    if (LD == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int k = k0 + 32 * ci + (lane >> 5) + 2 * i, c = lane & 31;
        gs[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsx, (c < wcols && (unsigned)k < (unsigned)R) ? (k * Q + qw + c) * 4 : kOob, 0, 0));
      }
    }
    if (V4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = k0 + 32 * ci + lr + 8 * i;                  // outside the image: an offset past the descriptor loads zeros
        g[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                                              rsx, (4 * c4 < wcols && (unsigned)k < (unsigned)R) ? (k * Q + qw + 4 * c4) * 4 : kOob, 0, 0));
      }
    }
  };
  auto lstore = [&](int ci) {
    float* d = wb + (ci & 1) * 32 * WS;
    if (V4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(d + (lr + 8 * i) * WS + 4 * c4) = g[i];
    } else if (LD == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) d[((lane >> 5) + 2 * i) * WS + (lane & 31)] = gs[i];
    } else {
      for (int e = lane; e < 32 * 32; e += 64) {
        const int rr = e >> 5, c = e & 31, k = k0 + 32 * ci + rr;
        d[rr * WS + c] = (c < wcols && (unsigned)k < (unsigned)R) ? xi[(size_t)k * Q + qw + c] : 0.f;
      }
    }
  };
  if (wcols > 0) gload(ci_lo);                                 // the first chunk flies while the taps table is built
  for (int j = tid; j < tzn; j += NTH) tz[j] = (j >= kBtPadLo && j < kBtPadLo + T) ? taps[j - kBtPadLo] : 0.f;
  band_acc_zero();
  __syncthreads();                                             // the taps table: the only barrier of the kernel
  if (wcols <= 0) return;

  // transposed store of row block jb from the chunk buffer `tt` (free at the time): tile element (r, q) of pixel column
  // p = (off + q) / C, channel c goes to tt[p][r*C + c]; whole pixel columns leave as float4 out[((sp)*R + rw0 + r)*C + c]
  const int off = qw % C, px0 = qw / C;                        // first (maybe partial) pixel column of the tile
  const bool st4 = LD != 0 && ((R * C) & 3) == 0 && ((uintptr_t)y & 15) == 0;    // float4 runs of the transposed image
  const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(yi, 0, LD ? R * Q * 4 : 0, 0x00020000);
  auto tile_write = [&](int jb, float* tt) {
    float a[16];
    band_acc_read(jb, a);
    int ln = lane;
    asm volatile("" : "+v"(ln));                               // indices are recomputed here, not kept in registers across the band
    const int colp = (ln & 31) + off, px_l = colp / C, c_l = colp - px_l * C, k4 = (ln >> 5) * 4;
    float* td = tt + px_l * RUN + k4 * C + c_l;
#pragma unroll
    for (int q = 0; q < 16; ++q) td[((q & 3) + 8 * (q >> 2)) * C] = a[q];
  };
  // One quarter (part 0..3) of the read-out of row block jb's tile: a row block drains over the four stages after its tile
  // was written, so that the stores of a block are spread over the band instead of leaving in one burst.
  // (Knock-out builds price the stores at 19 of 101 us at 143 taps, 10 of them the float4 store instructions themselves;
  // spreading them changed nothing measurable, the streaming cache policy 3 us.)
  auto drain = [&](int jb, const float* tt, int part) {
    const int rw0 = r0 + 32 * jb;
    const int nrow = min(32, R - rw0);
    int ln = lane;
    asm volatile("" : "+v"(ln));
    __builtin_amdgcn_wave_barrier();
    const int p_full_lo = off ? 1 : 0, p_full_hi = (off + wcols) / C;        // slots [lo, hi) hold all C channels
    if (st4 && nrow == 32) {
      const int obase = (px0 * R + rw0) * C;                   // float offset of (first slot, first row) in the transposed image
      const int f = ln + 64 * part;                            // at most 32 * 8 float4 in the full columns: four rounds
      if (f < (p_full_hi - p_full_lo) * 8 * C) {
        const int p = p_full_lo + f / (8 * C), w4 = f % (8 * C);
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, *reinterpret_cast<const float4*>(tt + p * RUN + 4 * w4)), rsy,
                                               (obase + p * R * C + 4 * w4) * 4, 0, BAND_ST_AUX);
      }
      // the partial columns at either end of the tile: single floats.  Only 3 channels have them (32 % C == 0 otherwise),
      // holding 1 or 2 of the 3 channels: n - 1 is the mask / shift of the (row, channel) split
      if constexpr (C == 3) {
        if (part == 0 && off) {
          const int n = C - off, e = ln;
          if (e < 32 * n) {
            const int r = e >> (n - 1), c = off + (e & (n - 1));
            yi[(size_t)obase + r * C + c] = tt[r * C + c];
          }
        }
        const int tail = (off + wcols) - p_full_hi * C;        // channels of the last, partial column
        if (part == 1 && tail > 0) {
          const int e = ln;
          if (e < 32 * tail) {
            const int r = e >> (tail - 1), c = e & (tail - 1);
            yi[(size_t)obase + (size_t)p_full_hi * R * C + r * C + c] = tt[p_full_hi * RUN + r * C + c];
          }
        }
      }
    } else {
      const int lo = off, hi = off + wcols;                    // tile columns in slot coordinates
      constexpr int per = (G::SLOTS * 32 * C / 64 + 4) / 4;    // rounds of 64 elements per part
#pragma unroll 1
      for (int k = part * per; k < (part + 1) * per; ++k) {
        const int e = ln + 64 * k;
        const int p = e / (32 * C), rem = e - p * 32 * C, r = rem / C, cq = p * C + rem % C;
        if (p < G::SLOTS && cq >= lo && cq < hi && r < nrow) yi[((size_t)(px0 + p) * R + rw0) * C + rem] = tt[p * RUN + rem];
      }
    }
    __builtin_amdgcn_wave_barrier();
  };

  const float* tzl = tz + kBtPadLo + kk - li;
  const float* wbl = wb + kk * WS + li;
  auto fetch = [&](BandFrag& f, int s) {
    const float* bb = wbl + ((s >> 2) & 1) * 32 * WS + (s & 3) * 8 * WS;
    const float* ta = tzl + 8 * s;
#pragma unroll
    for (int u = 0; u < 4; ++u) f.b[u] = bb[2 * u * WS];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
      for (int u = 0; u < 4; ++u) f.a[jb][u] = ta[2 * u - 32 * jb];
  };
  auto compute = [&](const BandFrag& f, int s) {
    const int s8 = 8 * s;
    if (k0 + s8 + 8 <= 0 || k0 + s8 >= R) return;              // the whole stage lies outside the image
    bool on[4];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) on[jb] = s8 + 8 > 32 * jb && s8 < 32 * jb + L && r0 + 32 * jb < R;
#define BAND_U(u) band_mfma<0>(f.a[0][u], f.b[u]); band_mfma<1>(f.a[1][u], f.b[u]); band_mfma<2>(f.a[2][u], f.b[u]); band_mfma<3>(f.a[3][u], f.b[u]);
#define BAND_J(jb) if (on[jb]) { band_mfma<jb>(f.a[jb][0], f.b[0]); band_mfma<jb>(f.a[jb][1], f.b[1]); band_mfma<jb>(f.a[jb][2], f.b[2]); band_mfma<jb>(f.a[jb][3], f.b[3]); }
    if (on[0] && on[1] && on[2] && on[3]) {                    // k-pair major: consecutive MFMAs go to different accumulators
      BAND_U(0) BAND_U(1) BAND_U(2) BAND_U(3)
    } else {
      BAND_J(0) BAND_J(1) BAND_J(2) BAND_J(3)
    }
#undef BAND_U
#undef BAND_J
  };
  lstore(ci_lo);
  if (ci_lo + 1 < ci_hi) gload(ci_lo + 1);
  BandFrag X, Y;
  fetch(X, 4 * ci_lo);
  const int c_done = (L + 7) / 8 - 1;                          // row block jb is complete after stage 4 jb + c_done, i.e. with
  auto done_chunk = [&](int jb) { return min((4 * jb + c_done) >> 2, ci_hi - 1); };   // this chunk (or with the image)
  // A row block's tile is written one chunk after its band ended, behind the loader's LDS write, into the CURRENT chunk's
  // buffer (its last operands are in registers by then, and the loader overwrites it one chunk later), and drains from there
  // a quarter per stage.  The bands end one chunk apart, so at most one block is draining, except in the last chunk,
  // where the blocks the image cut short end together: those leave after the loop.
  int dj = -1;                                                 // the draining row block
  const float* dt = wb;
  for (int ci = ci_lo; ci < ci_hi; ++ci) {
    const int s = 4 * ci;
    fetch(Y, s + 1);
    compute(X, s);
    if (dj >= 0) drain(dj, dt, 2);
    fetch(X, s + 2);
    compute(Y, s + 1);
    if (dj >= 0) { drain(dj, dt, 3); dj = -1; }
    fetch(Y, s + 3);
    compute(X, s + 2);
    if (ci + 1 < ci_hi) lstore(ci + 1);                        // the next chunk lands in the buffer chunk ci - 1 has left
    for (int jb = 0; jb < 4; ++jb)
      if (r0 + 32 * jb < R && done_chunk(jb) == ci - 1) {      // ci - 1 < ci_hi - 1: one block at most
#ifdef BLUR_BAND_STAMP
        if (jb == 0) st1 = __builtin_amdgcn_s_memtime();
#endif
        dj = jb;
        dt = wb + (ci & 1) * 32 * WS;
        tile_write(jb, wb + (ci & 1) * 32 * WS);
        drain(dj, dt, 0);
      }
    if (ci + 2 < ci_hi) gload(ci + 2);
    fetch(X, s + 4);                                           // past the last chunk: stale values nobody multiplies
    compute(Y, s + 3);
    if (dj >= 0) drain(dj, dt, 1);
  }
  if (dj >= 0) { drain(dj, dt, 2); drain(dj, dt, 3); }
  for (int jb = 0; jb < 4; ++jb)
    if (r0 + 32 * jb < R && done_chunk(jb) == ci_hi - 1) {
      float* tt = wb + (jb & 1) * 32 * WS;
      tile_write(jb, tt);
#pragma unroll 1
      for (int part = 0; part < 4; ++part) drain(jb, tt, part);
    }
#ifdef BLUR_BAND_STAMP
  if (lane == 0 && blockIdx.x * G::NW + wave < 4096) {
    unsigned long long* d = g_band_dbg + (size_t)(blockIdx.x * G::NW + wave) * 4;
    d[0] = st0; d[1] = st1; d[2] = __builtin_amdgcn_s_memtime();
    d[3] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |             // HW_ID
           ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32);       // XCC_ID
  }
#endif
}
#ifdef BLUR_BAND_STAMP
extern "C" int bg_dbg_band_read(unsigned long long* host, size_t n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_band_dbg), n * 8); }
#endif

// Flops one band pass issues on the matrix pipe: 4 MFMAs of 32x32x2 for every (row block, 8-row stage) pair the kernel visits
// -- the stages of rows [32 jb, 32 jb + 32 + T - 1) relative to r0 - T/2 that are not wholly outside the image -- for every
// 32-float column tile that holds source columns.  (Same predicates as `compute` in the kernel.)
static double band_exec_flops(int B, int R, int S, int C, int T) {
  const int half = T >> 1, L = 32 + T - 1;
  const long tiles = ((long)S * C + 31) / 32;
  long pairs = 0;
  for (int r0 = 0; r0 < R; r0 += kBtRows) {
    const int k0 = r0 - half, k_end = std::min(R, r0 + kBtRows + half);
    const int ci_lo = k0 < 0 ? (-k0) >> 5 : 0, ci_hi = (k_end - k0 + 31) >> 5;
    for (int st = 4 * ci_lo; st < 4 * ci_hi; ++st) {
      const int s8 = 8 * st;
      if (k0 + s8 + 8 <= 0 || k0 + s8 >= R) continue;
      for (int jb = 0; jb < 4; ++jb) pairs += (s8 + 8 > 32 * jb && s8 < 32 * jb + L && r0 + 32 * jb < R) ? 1 : 0;
    }
  }
  return (double)B * tiles * pairs * 4.0 * (2.0 * 32 * 32 * 2);
}

template <int C>
void launch_band_t(hipStream_t s, const float* src, float* dst, int B, int R, int S, const float* taps, int T) {
  using G = BandCfg<C>;
  const int cgs = (int)bg::cdiv(S, G::PXW), rgs = (int)bg::cdiv(R, kBtRows);
  const size_t lds = ((size_t)band_tz_floats(T) + G::BUF) * sizeof(float);
  const dim3 grid((unsigned)((size_t)B * rgs * cgs));
  const bool off32 = (size_t)R * S * C < ((size_t)1 << 29);    // 32-bit byte offsets per image
  int ld = !off32 ? 0 : (((S * C) & 3) == 0 && ((uintptr_t)src & 15) == 0) ? 2 : 1;
  if (const char* f = getenv("BG_BLUR_BAND_LD")) ld = std::min(ld, std::max(0, atoi(f)));    // test aid: force a slower loader (read per call)
  auto kern = ld == 2 ? blur_band_t_kernel<C, 2> : ld == 1 ? blur_band_t_kernel<C, 1> : blur_band_t_kernel<C, 0>;
  bg::launch(kern, grid, dim3(G::NTH), lds, s, src, dst, R, S, rgs, cgs, taps, T);
}

// ------------------------------------------------------------------------------------------------
// Images larger than 64 x 64 at up to 65 taps: BOTH passes in one launch, no scratch image (traffic == algorithmic + the column
// halo, which the strips of one image share through their XCD's L2).
//
// A workgroup owns a strip of 32 output pixel columns of one image and streams down its rows in chunks of 32.  The two passes
// are the banded Toeplitz products  Z = T_H (X T_W^T)  -- associativity puts the W pass first, so the H pass runs on the
// 32-pixel-wide result instead of the haloed input and no MFMA is spent on halo columns:
//   W pass  a chunk of 32 raw rows x (32 + 32*HB) pixels (LDS, row stride = 2 mod 4 floats with stride/2 odd: the per-channel
//           operand reads at stride C floats are conflict-free) -> 32 rows x 32 pixels per channel, written interleaved (NHWC)
//           into a ring of 16-row blocks;
//   H pass  an output block of 16 rows contracts over the 2*HB + 1 ring blocks around it (ring row stride = 16 mod 32) and is
//           computed TRANSPOSED (data as the MFMA's row operand, Toeplitz as its column operand) so that every lane ends with
//           4 consecutive floats of one row and stores a float4.
// v_mfma_f32_16x16x4_f32 tiles: a 16-wide output block needs 16 + taps - 1 <= 16*(2*HB + 1) source positions, against
// 32 + taps - 1 for the 32-wide instruction (48 vs 64 at 31 taps).  The Toeplitz fragment of a lane depends only on
// (k - n): ONE set of 4*(2*HB+1) registers, loaded once, serves every tile of both passes; each MFMA costs one ds_read_b32.
// SAME zero padding is zeros in LDS (rows / columns outside the image load as 0), never a modified Toeplitz block.
// The next chunk's global loads are issued into registers before the current chunk's MFMAs; two workgroups per CU.
// ------------------------------------------------------------------------------------------------
constexpr int kSP = 32;                                       // output pixel columns per strip

// P = reach of the band on either side of a 16-wide output block, a multiple of 4 >= taps/2: the block contracts over
// 16 + 2P source positions = 4 + P/2 k-steps of 4, every one of which carries taps.
template <int C, int P>
struct StripCfg {
  static constexpr int HB = (P + 15) / 16;                    // halo blocks of 16 (columns either side of the strip, ring depth)
  static constexpr int NQ = kSP * C;                          // output columns (floats) of a strip
  static constexpr int XW = (kSP + 32 * HB) * C;              // input columns per row: halo of 16*HB pixels either side
  static constexpr int XS = XW + 2;                           // XW is a multiple of 32 -> XS/2 odd
  static constexpr int YS = NQ + 16;                          // = 16 mod 32
  static constexpr int NK = 4 + P / 2;                        // k-steps of 4 per 16-wide output block
  static constexpr int RB = 2 * HB + 2;                       // ring blocks of 16 rows
  static constexpr int F4 = XW / 4;                           // float4 per input row
  static constexpr int NLD = 32 * F4 / 256;                   // float4 per thread per chunk
  static constexpr int TZ = 128;                              // zero-padded taps table: index k - n - P + taps/2 + 48 in [0, 128)
  static constexpr size_t lds_bytes = ((size_t)32 * XS + (size_t)RB * 16 * YS + TZ) * sizeof(float);
  static_assert(P % 4 == 0 && P >= 4 && P <= 32, "band reach");     // instantiated for 24 and 32 (35..65 taps)
  static_assert(32 * F4 % 256 == 0, "chunk loads must divide evenly");
};

// Diagnostic build only (-DBLUR_STRIP_STAMP): wave 0 of every workgroup writes s_memtime stamps of each phase into a buffer
// nothing else reads (the scratch pointer of bg_blur_nhwc_f32); the product build compiles none of it.
#ifdef BLUR_STRIP_STAMP
#define STRIP_STAMP(k) do { if (tid == 0 && j < 12) dbg[((size_t)blockIdx.x * 12 + j) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define STRIP_DBG_PARAM , unsigned long long* __restrict__ dbg
#define STRIP_DBG_ARG , dbg
#else
#define STRIP_STAMP(k) do { } while (0)
#define STRIP_DBG_PARAM
#define STRIP_DBG_ARG
#endif

// ------------------------------------------------------------------------------------------------
// Up to 33 taps (band reach P <= 16): every WAVE is an autonomous pipeline -- no workgroup barrier, no shared tile.
// A wave owns a column strip of 16 output pixels (3 channels) or 48 (1 channel) of one image segment and walks down it in
// blocks of 16 rows; per block:
//   W pass   X block (16 rows x (16*TPW + 32) pixels, the wave's PRIVATE slice of LDS) -> Y block, one accumulator per plane
//            (plane = channel for RGB, 16-pixel tile for grey);
//   4x4 transpose between lane groups and registers (a 1 KB wave-private LDS tile per plane): an accumulator holds rows 4g+i of its
//            column in register i of lane group g; as a contraction operand the next pass wants row 4s+g in register s.  After it
//            the last three Y blocks simply stay in registers: no ring in LDS, nothing shared between waves;
//   H pass   output block b-1 from Y blocks b-2, b-1, b, computed transposed (Y as the row operand, Toeplitz as the column
//            operand): a lane ends with 4 consecutive pixels of one row -- all their channels = 12 contiguous floats for RGB;
//   the next X block's global loads are issued before the W pass and written to LDS after the H pass (in-order LDS, same wave).
// With no barriers the waves of a CU drift apart by themselves: one wave's loads, LDS traffic and stores run under another's
// MFMA chains.  Halo rows at a segment boundary cost one extra W pass each side; halo columns are re-read from the XCD's L2
// (48 pixels in for 16 out) -- HBM sees each byte once.
// ------------------------------------------------------------------------------------------------
template <int C, int P>
struct ColsCfg {
  static constexpr int TPW = C == 3 ? 1 : 3;                  // 16-pixel tiles per wave
  static constexpr int PXO = 16 * TPW;                        // output pixels per wave
  static constexpr int XW = (PXO + 32) * C;                   // input floats per row: 144 (RGB), 80 (grey)
  static constexpr int XS = XW + 2;                           // = 2 mod 4, XS/2 odd: operand reads at stride C are conflict-free
  static constexpr int NK = 4 + P / 2;
  static constexpr int F4 = XW / 4;
  static constexpr int NLD = 16 * F4 / 64;                    // float4 per lane per block: 9 (RGB), 5 (grey)
  static constexpr int TS = 17;                               // row stride of the per-plane transpose tile
  static constexpr int XBLK = 16 * XS + 3 * 16 * TS;          // per wave: X block + three 16 x 16 transpose tiles
  static constexpr size_t lds_bytes = ((size_t)4 * XBLK + 128) * sizeof(float);
  static_assert(C == 1 || C == 3, "planes are channels or tiles");
  static_assert(P % 4 == 0 && P >= 4 && P <= 16, "band reach within one block either side");
  static_assert(16 * F4 % 64 == 0, "block loads must divide evenly over a wave");
};

#ifndef BLUR_COLS_WPS
#define BLUR_COLS_WPS 3
#endif
template <int C, int P>
__global__ __launch_bounds__(256, BLUR_COLS_WPS) void blur_cols_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int strips,
                                                        int segs, int seg_rows, const float* __restrict__ taps, int T STRIP_DBG_PARAM) {
#ifdef BLUR_STRIP_STAMP
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#endif
  using K = ColsCfg<C, P>;
  constexpr int TPW = K::TPW, PXO = K::PXO, XS = K::XS, NK = K::NK, F4 = K::F4, NLD = K::NLD;
  extern __shared__ __attribute__((aligned(16))) float sl[];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* tz = sl + 4 * K::XBLK;                               // zero-padded taps (see blur_strip_kernel: never straight from global)
  if (tid < 128) tz[tid] = (tid >= 48 && tid < 48 + T) ? taps[tid - 48] : 0.f;
  __syncthreads();                                             // the only barrier: before any wave can have left
  const int half = T >> 1, WC = W * C;
  float toep[NK];
#pragma unroll
  for (int j = 0; j < NK; ++j) toep[j] = tz[4 * j + g - li - P + half + 48];
  // work unit of this wave: blocks b, b+8, ... share an XCD and walk the units of one image in order (halo columns from L2)
  const int upi = strips * segs, wpi = (upi + 3) >> 2;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int img = (idx / wpi) * 8 + xcd, unit = (idx - (idx / wpi) * wpi) * 4 + wave;
  if (img >= B || unit >= upi) return;
  const int seg = unit / strips, strip = unit - seg * strips;
  const int p0 = strip * PXO;                                 // first output pixel
  const int qin0 = (p0 - 16) * C;                             // first input column (may be negative)
  const int bl = seg * seg_rows >> 4, bh = min((H + 15) >> 4, (seg + 1) * seg_rows >> 4);   // output blocks [bl, bh)
  float* X = sl + wave * K::XBLK;
  float* TT = X + 16 * XS;
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (size_t)img * H * WC), 0, H * WC * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(y + (size_t)img * H * WC, 0, H * WC * 4, 0x00020000);
  constexpr int kOob = (int)0x80000000;
  float4 gq[NLD];
  auto gload = [&](int blk) {                                  // rows / columns outside the image: out-of-range offset -> zeros
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int e = lane + i * 64, rr = e / F4, c4 = e - rr * F4;
      const int r = blk * 16 + rr, q = qin0 + 4 * c4;
      gq[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsx, (r >= 0 && r < H && q >= 0 && q < WC) ? (r * WC + q) * 4 : kOob, 0, 0));
    }
  };
  auto xstore = [&]() {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int e = lane + i * 64, rr = e / F4, c4 = e - rr * F4;
      float2* d = reinterpret_cast<float2*>(X + rr * XS + 4 * c4);
      d[0] = make_float2(gq[i].x, gq[i].y);
      d[1] = make_float2(gq[i].z, gq[i].w);
    }
  };
  floatx4 Ya[3], Yb[3], Yc[3];
#pragma unroll
  for (int p = 0; p < 3; ++p) Ya[p] = Yb[p] = Yc[p] = floatx4{0.f, 0.f, 0.f, 0.f};
  int b = bl - 1;
  const int nblk_img = (H + 15) >> 4;
  // a block wholly outside the image (above the first segment, below the last) is zeros: no loads, no W pass
  auto inside = [&](int blk) { return blk >= 0 && blk < nblk_img; };
  if (inside(b)) { gload(b); xstore(); }
  // one block step; Y2 receives block b, Y0 / Y1 hold blocks b-2 / b-1
  auto step = [&](floatx4 (&Y0)[3], floatx4 (&Y1)[3], floatx4 (&Y2)[3]) {
    const bool next_in = b + 1 <= bh && inside(b + 1);         // wave-uniform
    if (next_in) gload(b + 1);                                 // in flight under this step's MFMAs
    if (!inside(b)) {
#pragma unroll
      for (int p = 0; p < 3; ++p) Y2[p] = floatx4{0.f, 0.f, 0.f, 0.f};
    } else {
      // operand reads in groups of 4 k-steps, one group ahead of the MFMAs that consume them (all at once costs 12 more live
      // registers and a wave per SIMD)
      constexpr int GS = 4, NG = (NK + GS - 1) / GS;
      const float* xa = X + li * XS + (16 - P + g) * C;
      float av[2][GS][3];
      auto rd = [&](int grp, float (&dst)[GS][3]) {
#pragma unroll
        for (int u = 0; u < GS; ++u) {
          const int j = grp * GS + u;
          if (j < NK) {
#pragma unroll
            for (int p = 0; p < 3; ++p) dst[u][p] = C == 3 ? xa[4 * j * C + p] : xa[4 * j + 16 * p];
          }
        }
      };
      rd(0, av[0]);
#pragma unroll
      for (int p = 0; p < 3; ++p) Y2[p] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int grp = 0; grp < NG; ++grp) {
        if (grp + 1 < NG) rd(grp + 1, av[(grp + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < GS; ++u) {
          const int j = grp * GS + u;
          if (j < NK) {
#pragma unroll
            for (int p = 0; p < 3; ++p) Y2[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[grp & 1][u][p], toep[j], Y2[p], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // accumulator layout (row 4g+i in register i) -> contraction-operand layout (row 4s+g in register s) through the wave's own
      // LDS tile: same wave writes and reads, LDS serves a wave's operations in order, no barrier.  (The register-only route,
      // v_permlane16_swap + v_permlane32_swap, is miscompiled by this hipcc: two swaps of different register pairs are merged
      // into one -- tools/probes/permlane_swap_miscompile.hip reproduces it.)  The first H-pass k-steps read the two OLDER blocks, so this round
      // trip hides under them.
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        float* tp = TT + p * 16 * K::TS;
#pragma unroll
        for (int i = 0; i < 4; ++i) tp[(4 * g + i) * K::TS + li] = Y2[p][i];
      }
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const float* tp = TT + p * 16 * K::TS;
#pragma unroll
        for (int sI = 0; sI < 4; ++sI) Y2[p][sI] = tp[(4 * sI + g) * K::TS + li];
      }
    }
    const int o = b - 1;
    if (o >= bl) {                                             // wave-uniform
      floatx4 acc[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) acc[p] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < NK; ++j) {
        constexpr int dummy = 0; (void)dummy;
        const int off = 4 * j - P;                             // first row of this k-step relative to the output block (compile time)
        const int blk = off < 0 ? 0 : (off < 16 ? 1 : 2);      // Y0 / Y1 / Y2
        const int sib = (off + 16) % 16 / 4;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          const float a = blk == 0 ? Y0[p][sib] : (blk == 1 ? Y1[p][sib] : Y2[p][sib]);
          acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, toep[j], acc[p], 0, 0, 0);
        }
      }
      const int row = 16 * o + li;
      typedef unsigned u4 __attribute__((__vector_size__(4 * sizeof(unsigned))));
      if (C == 3) {
        float v[12];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int c = 0; c < 3; ++c) v[3 * i + c] = acc[c][i];
        const int q = (p0 + 4 * g) * 3;                        // 4 consecutive pixels, all channels: 12 contiguous floats
        const int base = (row < H && p0 + 4 * g < W) ? (row * WC + q) * 4 : kOob;
#pragma unroll
        for (int f = 0; f < 3; ++f)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, make_float4(v[4 * f], v[4 * f + 1], v[4 * f + 2], v[4 * f + 3])), rsy,
                                                 base == kOob ? kOob : base + 16 * f, 0, BLUR_ST_AUX);
      } else {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          const int q = p0 + 16 * p + 4 * g;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, acc[p]), rsy, (row < H && q < W) ? (row * WC + q) * 4 : kOob, 0, BLUR_ST_AUX);
        }
      }
    }
    if (next_in) xstore();                                     // block b+1 replaces block b (every read of b has been issued)
    ++b;
  };
  // the two older blocks move down by register copies (24 v_mov per 72 MFMAs): one step body instead of three rotated ones
  // keeps the kernel under the register budget of three waves per SIMD
  for (; b <= bh;) {
    step(Ya, Yb, Yc);
#pragma unroll
    for (int p = 0; p < 3; ++p) { Ya[p] = Yb[p]; Yb[p] = Yc[p]; }
  }
#ifdef BLUR_STRIP_STAMP
  if (lane == 0) {                       // shader cycles and 100 MHz reference ticks of this wave's life
    unsigned long long* d = dbg + ((size_t)blockIdx.x * 4 + wave) * 4;
    d[0] = t0; d[1] = r0; d[2] = __builtin_amdgcn_s_memtime(); d[3] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

template <int C, int P>
int launch_cols(hipStream_t s, const float* x, float* y, int B, int H, int W, const float* taps, int T, void* dbg_) {
#ifdef BLUR_STRIP_STAMP
  unsigned long long* dbg = static_cast<unsigned long long*>(dbg_);
#endif
  using K = ColsCfg<C, P>;
  const int strips = (int)bg::cdiv(W, K::PXO);
  // segments of rows: enough waves for two per SIMD (2048 on 256 CUs; measured best: 1024 / 2048 / 3072 / 4096 waves give
  // 38.6 / 35.9 / 37.0 / 41.1 us on 64 x 256x256x3 -- a segment boundary costs two extra W passes per wave), each segment at
  // least 32 rows
  static const int want = getenv("BG_BLUR_COLS_WAVES") ? atoi(getenv("BG_BLUR_COLS_WAVES")) : 2048;
  int segs = (int)std::min<long>(std::max<long>(1, bg::cdiv(want, (size_t)B * strips)), std::max(1, H / 32));
  const int seg_rows = (int)bg::cdiv(bg::cdiv(H, segs), 16) * 16;
  segs = (int)bg::cdiv(H, seg_rows);
  const int wpi = (strips * segs + 3) / 4;
  const dim3 grid((unsigned)(8 * bg::cdiv(B, 8) * wpi));
  auto kern = blur_cols_kernel<C, P>;
  const size_t lds = K::lds_bytes;
  bg::launch(kern, grid, dim3(256), lds, s, x, y, B, H, W, strips, segs, seg_rows, taps, T STRIP_DBG_ARG);
  return BG_OK;
}

template <int C>
int launch_cols_c(int reach, hipStream_t s, const float* x, float* y, int B, int H, int W, const float* taps, int T, void* dbg) {
  if (reach <= 4) return launch_cols<C, 4>(s, x, y, B, H, W, taps, T, dbg);
  if (reach <= 8) return launch_cols<C, 8>(s, x, y, B, H, W, taps, T, dbg);
  return launch_cols<C, 16>(s, x, y, B, H, W, taps, T, dbg);
}

template <int C, int P>
__global__ __launch_bounds__(256) void blur_strip_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int strips,
                                                         const float* __restrict__ taps, int T STRIP_DBG_PARAM) {
  using K = StripCfg<C, P>;
  constexpr int HB = K::HB, XS = K::XS, YS = K::YS, NK = K::NK, RB = K::RB, F4 = K::F4, NLD = K::NLD;
  constexpr int NCH = C == 1 ? 2 : C;                         // independent accumulator chains (16x16x4 has a 40-cycle dependent latency)
  extern __shared__ __attribute__((aligned(16))) float sl[];
  float* X = sl;                                              // [32][XS]   raw chunk
  float* Y = sl + 32 * XS;                                    // [RB*16][YS] W-passed rows, circular
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, kk = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // (Measured and dropped: a static s_setprio for one of the two co-resident workgroups -- by hardware wave slot or by LDS base --
  // and a start stagger between them: neither moves the kernel time, profiles/r02_b_blur_notes.md.)
  // blocks b, b+8, b+16, ... share an XCD (round-robin dispatch): the strips of one image are consecutive there and re-use each
  // other's halo columns from that L2 (speed only)
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int img = (idx / strips) * 8 + xcd, strip = idx - (idx / strips) * strips;
  if (img >= B) return;
  const int half = T >> 1, WC = W * C;
  const int s0 = strip * kSP;
  const int qin0 = (s0 - 16 * HB) * C;                        // first input column of the strip (may be negative)
  const float* xi = x + (size_t)img * H * WC;
  float* yi = y + (size_t)img * H * WC;
  // Toeplitz fragment: element (k, n) of a band block holds taps[k - n - P + half].  Read through a zero-padded table in LDS,
  // not straight from global memory: registers that a vector-memory load fills before the loop make the compiler wait for
  // vmcnt(0) at their first use INSIDE the loop, i.e. for the next chunk's prefetch it has just issued.
  float* tz = Y + RB * 16 * YS;
  if (tid < K::TZ) tz[tid] = (tid >= 48 && tid < 48 + T) ? taps[tid - 48] : 0.f;
  for (int e = tid; e < RB * 16 * YS; e += 256) Y[e] = 0.f;   // blocks above the image read as zero
  __syncthreads();
  float toep[NK];
#pragma unroll
  for (int s = 0; s < NK; ++s) toep[s] = tz[4 * s + kk - li - P + half + 48];
  float4 g[NLD];
  // chunk loader, one float4 piece at a time so that the pieces can be placed between the MFMAs of the H pass.  Buffer loads
  // with a per-image descriptor: rows / columns outside the image take an out-of-range offset and come back as zeros (SAME
  // padding with no branch and no select)
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xi), 0, H * WC * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(yi, 0, H * WC * 4, 0x00020000);
  constexpr int kOob = (int)0x80000000;
  auto gload1 = [&](int i, int row0, bool live) {
    const int e = tid + i * 256, rr = e / F4, c4 = e - rr * F4;
    const int r = row0 + rr, q = qin0 + 4 * c4;
    g[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsx, (live && r < H && q >= 0 && q < WC) ? (r * WC + q) * 4 : kOob, 0, 0));
  };
  auto xstore1 = [&](int i) {
    const int e = tid + i * 256, rr = e / F4, c4 = e - rr * F4;
    float2* d = reinterpret_cast<float2*>(X + rr * XS + 4 * c4);         // rows are 8-byte aligned (XS even)
    d[0] = make_float2(g[i].x, g[i].y);
    d[1] = make_float2(g[i].z, g[i].w);
  };
  const int nblk = (H + 15) >> 4, n_iter = (nblk + HB + 1) >> 1;
  // prologue: chunk 0 into LDS, chunk 1 in flight
#pragma unroll
  for (int i = 0; i < NLD; ++i) gload1(i, 0, true);
#pragma unroll
  for (int i = 0; i < NLD; ++i) xstore1(i);
#pragma unroll
  for (int i = 0; i < NLD; ++i) gload1(i, 32, true);
  __syncthreads();
  // Iteration j = two MFMA-paced phases with one barrier each:
  //   W phase  chunk j (in X) -> ring blocks 2j, 2j+1
  //   H phase  output blocks 2j-HB, 2j+1-HB from the ring; between its MFMAs the wave also moves chunk j+1 from its registers
  //            into X (free since the barrier) and issues the loads of chunk j+2 -- every memory instruction of the kernel
  //            except the result stores issues in the shadow of an MFMA chain
  constexpr int OPS = (2 * NLD + NK - 1) / NK;                // memory pieces per k-step of the H pass
  for (int j = 0; j < n_iter; ++j) {
    STRIP_STAMP(0);
    if (32 * j < H) {
      // C == 3: wave -> (row block, pixel tile), one accumulator per channel; C == 1: the same tiles, the k-steps alternate
      // between two accumulators
      const int rb = wave >> 1, pt = wave & 1;
      floatx4 acc[NCH];
#pragma unroll
      for (int c = 0; c < NCH; ++c) acc[c] = floatx4{0.f, 0.f, 0.f, 0.f};
      const float* xa = X + (16 * rb + li) * XS + (16 * pt + 16 * HB - P + kk) * C;
      // all operand reads of the tile first, then the MFMA chains: the matrix pipe never waits on an LDS round trip mid-chain
      float av[NK][C];
#pragma unroll
      for (int s = 0; s < NK; ++s)
#pragma unroll
        for (int c = 0; c < C; ++c) av[s][c] = xa[4 * s * C + c];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < NK; ++s) {
        if (C == 1) {
          acc[s & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][0], toep[s], acc[s & 1], 0, 0, 0);
        } else {
#pragma unroll
          for (int c = 0; c < C; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][c], toep[s], acc[c], 0, 0, 0);
        }
      }
      if (C == 1) acc[0] += acc[1];
      const int slot = (2 * j + rb) % RB;
      float* yo = Y + (slot * 16 + 4 * kk) * YS + (16 * pt + li) * C;
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i) yo[i * YS + c] = acc[c][i];
    } else {
      // below the image: the two blocks read as zero
      for (int e = tid; e < 2 * 16 * YS; e += 256) {
        const int blk = e / (16 * YS), r = e - blk * 16 * YS;
        Y[((2 * j + blk) % RB) * 16 * YS + r] = 0.f;
      }
    }
    STRIP_STAMP(1);
    __syncthreads();                                           // ring blocks visible; X is free
    STRIP_STAMP(2);
    {
      constexpr int NT = C == 1 ? 1 : 3;                      // column tiles of 16 per wave: 2 blocks x (NQ/16) tiles over 4 waves
      const int ob = wave >> 1, ct0 = (wave & 1) * NT;
      const int o = 2 * j - HB + ob;
      const bool ld_next = 32 * (j + 2) < H;                  // past the image: out-of-range offsets, the loads return zeros
      // straight-line body in two versions (a wave with no output block this iteration -- first / last iterations -- still moves
      // its share of the next chunk): no branch sits between the MFMAs and the memory pieces, which would also make the
      // compiler drain vmcnt before every piece
      auto h_phase = [&](auto hv_) {
        constexpr bool HV = decltype(hv_)::value;
        floatx4 acc[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) acc[c] = floatx4{0.f, 0.f, 0.f, 0.f};
        float bv[NK][NT];
        if (HV) {
          const int r0 = 16 * o - P + RB * 16;                // first band row in ring coordinates (>= 0, a multiple of 4)
          const float* yb = Y + kk * YS + 16 * ct0 + li;
#pragma unroll
          for (int s = 0; s < NK; ++s) {
            const float* yr = yb + ((r0 + 4 * s) % (RB * 16)) * YS;
#pragma unroll
            for (int t = 0; t < NT; ++t) bv[s][t] = yr[16 * t];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < NK; ++s) {
          if (HV) {
            if (C == 1) {
              acc[s & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[s][0], toep[s], acc[s & 1], 0, 0, 0);
            } else {
#pragma unroll
              for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[s][t], toep[s], acc[t], 0, 0, 0);
            }
          }
#pragma unroll
          for (int u = 0; u < OPS; ++u) {
            const int op = s * OPS + u;                       // pieces 0..NLD-1: registers -> X; NLD..2*NLD-1: next loads
            if (op < NLD) xstore1(op);
            else if (op < 2 * NLD) gload1(op - NLD, 32 * (j + 2), ld_next);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (HV) {
          if (C == 1) acc[0] += acc[1];
          const int row = 16 * o + li;
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const int q = s0 * C + 16 * (ct0 + t) + 4 * kk;   // 4 consecutive floats of one output row
            const bool ok = row < H && q < WC;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, acc[t]), rsy,
                                                   ok ? (row * WC + q) * 4 : kOob, 0, BLUR_ST_AUX);
          }
        }
      };
      if (o >= 0 && o < nblk) h_phase(std::true_type{});
      else h_phase(std::false_type{});
    }
    STRIP_STAMP(3);
    __syncthreads();                                           // chunk j+1 visible in X; every wave is past this H pass
    STRIP_STAMP(4);
  }
}

// Images larger than LDS: two passes through a scratch image in HBM (traffic 2x algorithmic), each pass holding
// whole lines along the filtered axis in LDS so no halo is ever re-read, same register sliding window.
//   pass H: workgroup = one image x a strip of kStripW consecutive (w,c) columns, all H rows resident
//   pass W: workgroup = kRowsW full rows of one image
constexpr int kStripW = 64, kLineThreads = 256;

__global__ __launch_bounds__(kLineThreads) void blur_lines_h_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                    int H, int WC, int strips,
                                                                    const float* __restrict__ taps, int T) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int half = T >> 1;
  float* s0 = lds;                         // [H][kStripW]
  float* tp = lds + H * kStripW;           // taps + kR zeros
  const int img = blockIdx.x / strips, strip = blockIdx.x - img * strips;
  const int c0 = strip * kStripW;
  const int cw = min(kStripW, WC - c0);
  const float* xi = x + (size_t)img * H * WC + c0;
  float* yi = y + (size_t)img * H * WC + c0;
  const int tid = threadIdx.x;
  for (int j = tid; j < T + kR; j += kLineThreads) tp[j] = j < T ? taps[j] : 0.f;
  for (int e = tid; e < H * kStripW; e += kLineThreads) {
    const int h = e / kStripW, c = e - h * kStripW;
    s0[e] = c < cw ? xi[(size_t)h * WC + c] : 0.f;
  }
  __syncthreads();
  const int HQ = (H + kR - 1) / kR;
  for (int item = tid; item < kStripW * HQ; item += kLineThreads) {
    const int hq = item / kStripW, c = item - hq * kStripW;
    if (c >= cw) continue;
    const int h0 = hq * kR;
    float acc[kR], tw[kR];
#pragma unroll
    for (int r = 0; r < kR; ++r) { acc[r] = 0.f; tw[r] = 0.f; }
    // only source rows inside the image contribute: clip the sweep to them (large T, small H)
    const int jlo = max(0, half - h0 - (kR - 1)), jhi = min(T + kR - 1, H + half - h0);
#pragma unroll
    for (int q = 0; q < kR - 1; ++q) tw[q] = (jlo - 1 - q >= 0) ? tp[jlo - 1 - q] : 0.f;   // tap history for the first sample
    for (int j = jlo; j < jhi; ++j) {
#pragma unroll
      for (int r = kR - 1; r > 0; --r) tw[r] = tw[r - 1];
      tw[0] = tp[j];
      const int srow = h0 - half + j;
      const float v = (unsigned)srow < (unsigned)H ? s0[srow * kStripW + c] : 0.f;
#pragma unroll
      for (int r = 0; r < kR; ++r) acc[r] = fmaf(tw[r], v, acc[r]);
    }
#pragma unroll
    for (int r = 0; r < kR; ++r)
      if (h0 + r < H) yi[(size_t)(h0 + r) * WC + c] = acc[r];
  }
}

__global__ __launch_bounds__(kLineThreads) void blur_lines_w_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                    int rows_total, int W, int C, int rows_per,
                                                                    const float* __restrict__ taps, int T) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int half = T >> 1, WC = W * C;
  float* s1 = lds;                         // [rows_per][WC]
  float* tp = lds + rows_per * WC;
  const int r0 = blockIdx.x * rows_per;
  const int nr = min(rows_per, rows_total - r0);
  const float* xi = x + (size_t)r0 * WC;
  float* yi = y + (size_t)r0 * WC;
  const int tid = threadIdx.x;
  for (int j = tid; j < T + kR; j += kLineThreads) tp[j] = j < T ? taps[j] : 0.f;
  for (int e = tid; e < nr * WC; e += kLineThreads) s1[e] = xi[e];
  __syncthreads();
  const int WQ = (W + kR - 1) / kR;
  for (int item = tid; item < nr * WQ * C; item += kLineThreads) {
    const int c = item % C;
    const int t2 = item / C;
    const int wq = t2 % WQ, h = t2 / WQ;
    const int w0 = wq * kR;
    const float* row = s1 + h * WC + c;
    float acc[kR], tw[kR];
#pragma unroll
    for (int r = 0; r < kR; ++r) { acc[r] = 0.f; tw[r] = 0.f; }
    const int jlo = max(0, half - w0 - (kR - 1)), jhi = min(T + kR - 1, W + half - w0);
#pragma unroll
    for (int q = 0; q < kR - 1; ++q) tw[q] = (jlo - 1 - q >= 0) ? tp[jlo - 1 - q] : 0.f;
    for (int j = jlo; j < jhi; ++j) {
#pragma unroll
      for (int r = kR - 1; r > 0; --r) tw[r] = tw[r - 1];
      tw[0] = tp[j];
      const int scol = w0 - half + j;
      const float v = (unsigned)scol < (unsigned)W ? row[scol * C] : 0.f;
#pragma unroll
      for (int r = 0; r < kR; ++r) acc[r] = fmaf(tw[r], v, acc[r]);
    }
#pragma unroll
    for (int r = 0; r < kR; ++r)
      if (w0 + r < W) yi[(size_t)(h * W + w0 + r) * C + c] = acc[r];
  }
}

// last-resort generic pass (lines that do not fit LDS): one thread per output through L1/L2
template <int AXIS>
__global__ __launch_bounds__(kBlurThreads) void blur_pass_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                 size_t total, int H, int W, int C,
                                                                 const float* __restrict__ taps, int T) {
  const int WC = W * C, half = T >> 1;
  for (size_t e = (size_t)blockIdx.x * kBlurThreads + threadIdx.x; e < total; e += (size_t)gridDim.x * kBlurThreads) {
    const size_t row = e / WC;                 // b*H + h
    const int h = (int)(row % H);
    const int w = (int)(e - row * WC) / C;
    const int pos = AXIS == 0 ? h : w;
    const int len = AXIS == 0 ? H : W;
    const int st = AXIS == 0 ? WC : C;
    const int jlo = max(0, half - pos), jhi = min(T, len + half - pos);
    const float* src = x + e - (size_t)half * st;   // only dereferenced for j in [jlo, jhi)
    float acc = 0.f;
    for (int j = jlo; j < jhi; ++j) acc = fmaf(taps[j], src[(size_t)j * st], acc);
    y[e] = acc;
  }
}

template <int C, int P>
int launch_strip(dim3 grid, hipStream_t s, const float* x, float* y, int B, int H, int W, int strips, const float* taps, int T, void* dbg_) {
#ifdef BLUR_STRIP_STAMP
  unsigned long long* dbg = static_cast<unsigned long long*>(dbg_);
#endif
  auto kern = blur_strip_kernel<C, P>;
  size_t lds = StripCfg<C, P>::lds_bytes;
  if (lds > 64 * 1024) BG_LDS_ATTR_ONCE(kern, lds, "blur_strip");      // one flag per instantiation
  bg::launch(kern, grid, dim3(256), lds, s, x, y, B, H, W, strips, taps, T STRIP_DBG_ARG);
  return BG_OK;
}

template <int C>
int launch_strip_c(int reach, dim3 grid, hipStream_t s, const float* x, float* y, int B, int H, int W, int strips, const float* taps, int T, void* dbg) {
  // band reach <= 16 (up to 33 taps) belongs to blur_cols_kernel; the workgroup form serves 35..65 taps
  if (reach <= 24) return launch_strip<C, 24>(grid, s, x, y, B, H, W, strips, taps, T, dbg);
  return launch_strip<C, 32>(grid, s, x, y, B, H, W, strips, taps, T, dbg);
}

size_t fused_lds_bytes(int H, int W, int C) { return (2 * (size_t)(((H * W * C) + 3) & ~3) + 512) * sizeof(float); }
constexpr size_t kFusedLdsCap = 150 * 1024;

}  // namespace

extern "C" {

int bg_blur_policy(float sigma, int H, int W, float* kernel_size, float* sigma_eff, int* n_taps) {
  BG_REQUIRE(H > 0 && W > 0, BG_ERR_BAD_SHAPE, "bg_blur_policy: H=%d W=%d", H, W);
  // gaussian_blur.py:21-26 "(6*std)*2//2+1", :67 clip to [3, max(h,w)], :29-31,71-72 sigma re-derived; float32.
  const float full = (float)(H > W ? H : W);
  float ks = floorf(((6.0f * sigma) * 2.0f) / 2.0f) + 1.0f;
  ks = fminf(fmaxf(ks, 3.0f), full);
  float s = (ks - 1.0f) / 6.0f;
  s = fmaxf(s, 0.01f);
  if (kernel_size) *kernel_size = ks;
  if (sigma_eff) *sigma_eff = s;
  if (n_taps) *n_taps = 2 * (int)floorf(ks / 2.0f) + 1;   // gaussian_blur.py:84 range(-(ks//2), ks//2+1)
  return BG_OK;
}

int bg_gauss_kernel_1d(float sigma_eff, float kernel_size, float* taps_host, int cap, int* n_taps) {
  BG_REQUIRE(taps_host != nullptr, BG_ERR_NULL, "bg_gauss_kernel_1d: taps_host is NULL");
  BG_REQUIRE(sigma_eff > 0.f && kernel_size >= 1.f, BG_ERR_BAD_SHAPE, "bg_gauss_kernel_1d: sigma=%g ks=%g", sigma_eff, kernel_size);
  const int half = (int)floorf(kernel_size / 2.0f);
  const int T = 2 * half + 1;
  BG_REQUIRE(T <= cap, BG_ERR_WORKSPACE, "bg_gauss_kernel_1d: need %d taps, capacity %d", T, cap);
  // gaussian_blur.py:85-87, float32 arithmetic
  const float denom = sqrtf(2.0f * 3.14159265358979323846f) * sigma_eff;
  const float two_s2 = 2.0f * (sigma_eff * sigma_eff);
  float sum = 0.f;
  for (int j = 0; j < T; ++j) {
    const float xv = (float)(j - half);
    taps_host[j] = expf(-((xv * xv) / two_s2)) / denom;
    sum += taps_host[j];
  }
  for (int j = 0; j < T; ++j) taps_host[j] /= sum;
  if (n_taps) *n_taps = T;
  return BG_OK;
}

// Which kernel family a blur call takes (bg_blur_workspace_bytes and bg_blur_nhwc_f32 must agree on the scratch image):
//   0 whole image per workgroup on the matrix cores (<= 64x64, >= 13 taps)     -- no scratch
//   1 whole image per workgroup, register sliding window (fits LDS, < 13 taps)  -- no scratch
//   2 two transposing banded-Toeplitz passes (C <= 4)                           -- scratch
//   3 line kernels / generic passes                                             -- scratch
//   4 fused streaming strips, both passes in one launch (1 or 3 channels, <= 65 taps, larger than 64x64) -- no scratch
//   6 32-row panels, both band passes in one launch (RGB, > 65 taps, up to 256 pixels wide)            -- no scratch
static int blur_path(int B, int H, int W, int C, int n_taps) {
  static const int strip_min = getenv("BG_BLUR_STRIP_MIN_SIZE") ? atoi(getenv("BG_BLUR_STRIP_MIN_SIZE")) : 65;
  static const int strip_max_taps = getenv("BG_BLUR_STRIP_MAX_TAPS") ? atoi(getenv("BG_BLUR_STRIP_MAX_TAPS")) : 65;
  if ((C == 1 || C == 3) && ((W * C) & 3) == 0 && n_taps <= std::min(strip_max_taps, 65) && (H >= strip_min || W >= strip_min) &&
      (size_t)H * W * C < (1u << 29))        // per-image buffer descriptors: 32-bit byte offsets
    return 4;
  static const int mfma_min_taps = getenv("BG_BLUR_MFMA_MIN_TAPS") ? atoi(getenv("BG_BLUR_MFMA_MIN_TAPS")) : 13;
  static const int band_t_min = getenv("BG_BLUR_BANDT_MIN_TAPS") ? atoi(getenv("BG_BLUR_BANDT_MIN_TAPS")) : 13;
  const int Hp = (H + 31) / 32 * 32, Wp = (W + 31) / 32 * 32;
  const size_t lds_m = ((size_t)2 * C * Hp * (Wp + 1) + n_taps + 2 * kTzPad) * sizeof(float);
  const bool no_rows = getenv("BG_BLUR_NO_ROWS") != nullptr;            // test aid, read per call: take the whole-image kernel instead
  if (!no_rows && H <= 64 && W <= 64 && C <= 4 && ((W * C) & 3) == 0 && n_taps >= mfma_min_taps &&
      rows_geom(H, W, C, n_taps).lds <= 80 * 1024)
    return 5;
  if (H <= 64 && W <= 64 && C <= 16 && n_taps >= mfma_min_taps && lds_m <= kFusedLdsCap) return 0;
  const bool fused_fits = fused_lds_bytes(H, W, C) <= kFusedLdsCap && n_taps <= 500;
  const bool band_ok = C <= 4 && (size_t)B * H * W * C < (1ull << 31);
  // both band passes in one launch (32-row panels, pass-1 result in LDS): RGB images of 96 ... 256 pixels a side beyond 65 taps
  static const int panel_min = getenv("BG_BLUR_PANEL_MIN_TAPS") ? atoi(getenv("BG_BLUR_PANEL_MIN_TAPS")) : 67;
  if (band_ok && n_taps >= panel_min && (H > 64 || W > 64) && bg::blur_panel_ok(B, H, W, C, n_taps) && !getenv("BG_BLUR_NO_PANEL")) return 6;
  // measured (tools/blur_sweep.py): the band kernels beat the line kernels at every tap count (128x128x3: 38-43 us against
  // 43-68; 256x256x3: 60-72 against 97-170)
  // ... and the sliding-window fused kernel from about 31 taps up (128x128x1: 28 us both at 31 taps, 37 against 77 at 129)
  if (fused_fits && !(band_ok && n_taps >= std::max(band_t_min, 31) && (H > 64 || W > 64))) return 1;
  if (band_ok) return 2;
  return 3;
}

size_t bg_blur_workspace_bytes(int B, int H, int W, int C, int n_taps) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
  const int path = blur_path(B, H, W, C, n_taps);
  return (path <= 1 || path >= 4) ? 0 : (size_t)B * H * W * C * sizeof(float);
}

int bg_blur_nhwc_f32(const float* x, float* y, int B, int H, int W, int C, const float* taps_d, int n_taps,
                     float* tmp_d, void* stream) {
  BG_REQUIRE(x && y && taps_d, BG_ERR_NULL, "bg_blur_nhwc_f32: null pointer");
  BG_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0, BG_ERR_BAD_SHAPE, "bg_blur_nhwc_f32: B=%d H=%d W=%d C=%d", B, H, W, C);
  BG_REQUIRE(n_taps >= 1 && (n_taps & 1), BG_ERR_BAD_SHAPE, "bg_blur_nhwc_f32: tap count %d must be odd", n_taps);
  BG_REQUIRE(bg::aligned16(x) && bg::aligned16(y), BG_ERR_BAD_ALIGNMENT, "bg_blur_nhwc_f32: x/y must be 16-byte aligned");
  const size_t total = (size_t)B * H * W * C;
  const double flops = 4.0 * n_taps * (double)total, bytes = 8.0 * (double)total;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int path = blur_path(B, H, W, C, n_taps);
  auto magic = [](unsigned d) {           // FastDiv: exact for dividends < 2^20
    FastDiv f;
    unsigned sft = 0;
    while ((1u << sft) < d) ++sft;
    f.sh = 20 + sft;
    f.mul = (unsigned)(((1ull << f.sh) + d - 1) / d);
    return f;
  };
  if (path == 6) {
    bg::Launch L(stream, "blur_panel", flops, bytes);
    const int rc = bg::blur_panel_launch(x, y, B, H, W, taps_d, n_taps, s);
    if (rc) return rc;
    L.exec_flops(bg::blur_panel_exec_flops(B, H, W, n_taps));
    return L.done("blur_panel_kernel");
  }
  if (path == 5) {
    const RowsGeom g = rows_geom(H, W, C, n_taps);
    BG_LDS_ATTR_ONCE(blur_rows_kernel, 80 * 1024, "bg_blur_nhwc_f32");
    bg::Launch L(stream, "blur_rows", flops, bytes);
    bg::launch(blur_rows_kernel, dim3((unsigned)(B * g.nb)), dim3(kRowsThreads), g.lds, s, x, y, H, W, C, g.nb, g.Qp, g.Wp, g.xfloats,
                       taps_d, n_taps, magic((unsigned)(g.Q / 4)), nullptr, nullptr, B);
    return L.done("blur_rows_kernel");
  }
  {
    const int Hp = (H + 31) / 32 * 32, Wp = (W + 31) / 32 * 32;
    const size_t lds_m = ((size_t)2 * C * Hp * (Wp + 1) + n_taps + 2 * kTzPad) * sizeof(float);
    if (path == 0) {
      BG_LDS_ATTR_ONCE(blur_mfma_kernel, kFusedLdsCap, "bg_blur_nhwc_f32");
      bg::Launch L(stream, "blur_mfma", flops, bytes);
      bg::launch(blur_mfma_kernel, dim3(B), dim3(kMfmaBlurThreads), lds_m, s, x, y, H, W, C, Hp, Wp, taps_d, n_taps, magic((unsigned)C),
                         magic((unsigned)W));
      return L.done("blur_mfma_kernel");
    }
  }
  if (path == 4) {
    const int strips = (int)bg::cdiv(W, kSP);
    const dim3 grid((unsigned)(8 * bg::cdiv(B, 8) * strips));
    if ((n_taps >> 1) <= 16) {
      bg::Launch L(stream, "blur_cols", flops, bytes);
      const int rc = C == 3 ? launch_cols_c<3>(n_taps >> 1, s, x, y, B, H, W, taps_d, n_taps, tmp_d) : launch_cols_c<1>(n_taps >> 1, s, x, y, B, H, W, taps_d, n_taps, tmp_d);
      if (rc) return rc;
      return L.done("blur_cols_kernel");
    }
    bg::Launch L(stream, "blur_strip", flops, bytes);
    const int rc = C == 3 ? launch_strip_c<3>(n_taps >> 1, grid, s, x, y, B, H, W, strips, taps_d, n_taps, tmp_d)
                          : launch_strip_c<1>(n_taps >> 1, grid, s, x, y, B, H, W, strips, taps_d, n_taps, tmp_d);
    if (rc) return rc;
    return L.done("blur_strip_kernel");
  }
  const size_t lds = fused_lds_bytes(H, W, C);
  if (path == 1) {
    BG_LDS_ATTR_ONCE(blur_fused_kernel, kFusedLdsCap, "bg_blur_nhwc_f32");
    bg::Launch L(stream, "blur_fused", flops, bytes);
    bg::launch(blur_fused_kernel, dim3(B), dim3(kFusedThreads), lds, s, x, y, H, W, C, taps_d, n_taps);
    return L.done("blur_fused_kernel");
  }
  BG_REQUIRE(tmp_d != nullptr, BG_ERR_WORKSPACE, "bg_blur_nhwc_f32: image of %zu bytes needs tmp_d (see bg_blur_workspace_bytes)",
             (size_t)H * W * C * 4);
  const int WC = W * C;
  {
    // two transposing banded-Toeplitz passes on the matrix cores: x[H][W][C] -> tmp[W][H][C] -> y[H][W][C].
    // measured (64 x 256x256x3): 31 / 143 / 255 taps 0.075 / 0.126 / 0.160 ms against 0.17 / 0.47 / 0.67 for the line kernels;
    // 128 x 128x128x3 at 31 taps 0.043 against 0.068.
    if (path == 2) {
      for (int pass = 0; pass < 2; ++pass) {
        const int R = pass == 0 ? H : W, S = pass == 0 ? W : H;
        bg::Launch L(stream, pass == 0 ? "blur_band_t1" : "blur_band_t2", flops / 2, bytes);
        const float* src = pass == 0 ? x : tmp_d;
        float* dst = pass == 0 ? tmp_d : y;
        switch (C) {
          case 1: launch_band_t<1>(s, src, dst, B, R, S, taps_d, n_taps); break;
          case 2: launch_band_t<2>(s, src, dst, B, R, S, taps_d, n_taps); break;
          case 3: launch_band_t<3>(s, src, dst, B, R, S, taps_d, n_taps); break;
          default: launch_band_t<4>(s, src, dst, B, R, S, taps_d, n_taps); break;
        }
        L.exec_flops(band_exec_flops(B, R, S, C, n_taps));
        int rc = L.done("blur_band_t_kernel");
        if (rc) return rc;
      }
      return BG_OK;
    }
  }
  const size_t lds_h = ((size_t)H * kStripW + n_taps + kR + 4) * sizeof(float);
  int rows_per = (int)std::min<size_t>(8, std::max<size_t>(1, (48 * 1024 / sizeof(float)) / (size_t)WC));
  const size_t lds_w = ((size_t)rows_per * WC + n_taps + kR + 4) * sizeof(float);
  if (lds_h <= 140 * 1024 && lds_w <= 140 * 1024) {
    BG_LDS_ATTR_ONCE(blur_lines_h_kernel, 140 * 1024, "bg_blur_nhwc_f32");
    BG_LDS_ATTR_ONCE(blur_lines_w_kernel, 140 * 1024, "bg_blur_nhwc_f32");
    const int strips = (int)bg::cdiv(WC, kStripW);
    {
      bg::Launch L(stream, "blur_lines_h", flops / 2, bytes);
      bg::launch(blur_lines_h_kernel, dim3((unsigned)B * strips), dim3(kLineThreads), lds_h, s, x, tmp_d, H, WC, strips, taps_d, n_taps);
      int rc = L.done("blur_lines_h_kernel");
      if (rc) return rc;
    }
    const int rows_total = B * H;
    bg::Launch L(stream, "blur_lines_w", flops / 2, bytes);
    bg::launch(blur_lines_w_kernel, dim3(bg::cdiv(rows_total, rows_per)), dim3(kLineThreads), lds_w, s, tmp_d, y, rows_total, W, C,
                       rows_per, taps_d, n_taps);
    return L.done("blur_lines_w_kernel");
  }
  const unsigned grid = (unsigned)std::min<size_t>(bg::cdiv(total, kBlurThreads), 256 * 16);
  {
    bg::Launch L(stream, "blur_pass_h", flops / 2, bytes);
    bg::launch(blur_pass_kernel<0>, dim3(grid), dim3(kBlurThreads), 0, s, x, tmp_d, total, H, W, C, taps_d, n_taps);
    int rc = L.done("blur_pass_kernel<H>");
    if (rc) return rc;
  }
  bg::Launch L(stream, "blur_pass_w", flops / 2, bytes);
  bg::launch(blur_pass_kernel<1>, dim3(grid), dim3(kBlurThreads), 0, s, tmp_d, y, total, H, W, C, taps_d, n_taps);
  return L.done("blur_pass_kernel<W>");
}

int bg_blur3_lerp_supported(int B, int H, int W, int C, int n_taps) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || n_taps < 1 || !(n_taps & 1)) return 0;
  if (blur_path(3 * B, H, W, C, n_taps) == 5) return 1;
  // Below the tap count from which the row-block kernel is the single-source choice (13) the three-source launch still wins:
  // at 28x28x1 / 3-7 taps every blur launch costs its ~5 us floor, and one launch replaces lerp + three of them (MNIST step)
  return !getenv("BG_BLUR_NO_ROWS") && H <= 64 && W <= 64 && C <= 4 && ((W * C) & 3) == 0 && n_taps >= 3 &&
                 rows_geom(H, W, C, n_taps).lds <= 80 * 1024
             ? 1
             : 0;
}

int bg_blur3_lerp_nhwc_f32(const float* f, const float* r, const float* alpha_b, float* y3, int B, int H, int W, int C, const float* taps_d,
                           int n_taps, void* stream) {
  BG_REQUIRE(f && r && alpha_b && y3 && taps_d, BG_ERR_NULL, "bg_blur3_lerp_nhwc_f32: null pointer");
  BG_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0, BG_ERR_BAD_SHAPE, "bg_blur3_lerp_nhwc_f32: B=%d H=%d W=%d C=%d", B, H, W, C);
  BG_REQUIRE(n_taps >= 1 && (n_taps & 1), BG_ERR_BAD_SHAPE, "bg_blur3_lerp_nhwc_f32: tap count %d must be odd", n_taps);
  BG_REQUIRE(bg::aligned16(f) && bg::aligned16(r) && bg::aligned16(y3), BG_ERR_BAD_ALIGNMENT, "bg_blur3_lerp_nhwc_f32: f / r / y3 must be 16-byte aligned");
  BG_REQUIRE(bg_blur3_lerp_supported(B, H, W, C, n_taps), BG_ERR_UNSUPPORTED,
             "bg_blur3_lerp_nhwc_f32: %dx%dx%d at %d taps is not on the row-block kernel (see bg_blur3_lerp_supported): run bg_lerp_f32 and three bg_blur_nhwc_f32",
             H, W, C, n_taps);
  const RowsGeom g = rows_geom(H, W, C, n_taps);
  BG_LDS_ATTR_ONCE(blur_rows_kernel, 80 * 1024, "bg_blur3_lerp_nhwc_f32");
  const size_t total = (size_t)3 * B * H * W * C;
  bg::Launch L(stream, "blur_rows3", 4.0 * n_taps * (double)total, (8.0 / 3.0 + 4.0) * (double)total);     // reads f and r (x-hat rides on them), writes three
  FastDiv dq;
  {
    const unsigned d = (unsigned)(g.Q / 4);
    unsigned sft = 0;
    while ((1u << sft) < d) ++sft;
    dq.sh = 20 + sft;
    dq.mul = (unsigned)(((1ull << dq.sh) + d - 1) / d);
  }
  bg::launch(blur_rows_kernel, dim3((unsigned)(3 * B * g.nb)), dim3(kRowsThreads), g.lds, static_cast<hipStream_t>(stream), f, y3, H, W, C, g.nb,
             g.Qp, g.Wp, g.xfloats, taps_d, n_taps, dq, r, alpha_b, B);
  return L.done("blur_rows_kernel(3 sources)");
}

}  // extern "C"
