// Row-staged MFMA kernels for the 16-channel ends of the 128x128 networks (reference demo_celeba.py:84-90 ConvT 32->16,
// :99-103 Conv 16->32 and their gradients, wgan.py:140,166,244).  SURVEY.md 8a rows T1/T2, config C4.
//
// With 16 channels on one side a gather-GEMM tile is half padding (the 32-wide MFMA needs N >= 32) and, worse, re-reads every
// source pixel once per tap from L2: 25 taps x 128 B per 64 B of output made the per-tap gather the bound (measured 37 TFLOP/s
// on ConvT 32->16 at 128x128).  These kernels stage the source ROWS in LDS once (register-prefetched, double-buffered, zero
// halo columns), keep the whole 5x5 weight set in LDS, and run v_mfma_f32_16x16x4_f32 tiles whose N is exactly 16:
//   conv_c16_dgrad_kernel   data gradient of a stride-2 5x5 conv with Cin = 16 (== forward of ConvT -> 16 channels):
//                           out[b, 2a+py, 2c+px, n] = sum_{kh = py+1 (2), kw = px+1 (2), ck} src[b, a + (py+1-kh)/2, c + (px+1-kw)/2, ck] * w[kh,kw][n][ck]
// Workgroups are persistent (the weights are loaded once) and loop over strips of 2 anchor rows (4 output rows).
#include "conv_common.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOob = 0x80000000u;

struct C16Params {
  const float* A;    // source rows [B][Hs][Ws][Ck]
  const float* Wt;   // [25][16][Ck]
  float* C;          // [B][Hd][Wd][16]
  int B, Hs, Ws, Ck, Hd, Wd, N;
  int nstrips, strips_per_img;
  unsigned a_bytes, w_bytes;
  int epi_mode;
  const float* bias;
  const float* ref;
  const unsigned char* keep;
  size_t keep_elems;
  float alpha, scale;
};

// ------------------------------------------------------------------------------------------------------------------------
// data gradient, N = 16, Ck = 32, k = 5, s = 2, even H and W (pt = pl = 1): phase (py, px) of output pixel (2a+py, 2c+px) takes
// kernel rows kh = py+1 (mod 2) from source row a + (py+1-kh)/2 in {a-1, a, a+1}, same along the row.
// LDS: weights [25*16][36], rows [2 buffers][4 rows: a0-1 .. a0+2][(WS + 2) pixels][36]; pixel stride 36 floats keeps the b128
// fragment reads of 16 lanes on distinct banks.  Wave w: anchor row a0 + (w >> 1), anchor columns (w & 1) * WS/2 ...
// ------------------------------------------------------------------------------------------------------------------------
constexpr int kDgCk = 32, kDgAst = kDgCk + 4, kDgRows = 4;

template <int WS>
__global__ __launch_bounds__(256) void conv_c16_dgrad_kernel(const C16Params p) {
  constexpr int MT = WS / 32;                                 // 16-anchor tiles per wave
  constexpr int RSTR = (WS + 2) * kDgAst;                     // floats per staged row
  constexpr int PF = kDgRows * WS * (kDgCk / 4) / 256;        // float4 prefetched per thread and strip
  extern __shared__ __attribute__((aligned(16))) float c16_lds[];
  float* wl = c16_lds;
  float* rows = c16_lds + 25 * 16 * kDgAst;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int ar = wave >> 1, mh = wave & 1;

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, (int)p.a_bytes, 0x00020000);
  // weights -> LDS (once per workgroup); halo columns of both row buffers -> 0 (never written again)
  {                                                           // 3200 float4 = 12.5 per thread: all requested before the first LDS write
    constexpr int NW = (25 * 16 * (kDgCk / 4) + 255) / 256;
    float4 wv[NW];
#pragma unroll
    for (int u = 0; u < NW; ++u) {
      const int i = tid + u * 256, r = i >> 3, q = i & 7;
      wv[u] = i < 25 * 16 * (kDgCk / 4) ? *reinterpret_cast<const float4*>(p.Wt + (size_t)r * kDgCk + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < NW; ++u) {
      const int i = tid + u * 256, r = i >> 3, q = i & 7;
      if (i < 25 * 16 * (kDgCk / 4)) *reinterpret_cast<float4*>(wl + r * kDgAst + q * 4) = wv[u];
    }
  }
  for (int i = tid; i < 2 * kDgRows * 2 * kDgAst; i += 256) {
    const int e = i % kDgAst, side = (i / kDgAst) & 1, r = i / (2 * kDgAst);
    rows[r * RSTR + (side ? (WS + 1) * kDgAst : 0) + e] = 0.f;
  }

  float4 pf[PF];
  auto prefetch = [&](int strip) {
    const int b = strip / p.strips_per_img, a0 = (strip - b * p.strips_per_img) * 2;
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int item = i * 256 + tid;
      const int r = item / (WS * 8), rem = item - r * (WS * 8);
      const int ys = a0 - 1 + r;
      const bool ok = strip < p.nstrips && (unsigned)ys < (unsigned)p.Hs;
      const unsigned off = (unsigned)(((b * p.Hs + ys) * WS) * (kDgCk * 4) + rem * 16);
      pf[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsA, ok ? off : kOob, 0, 0));
    }
  };
  auto stash = [&](int buf) {
    float* dst = rows + buf * kDgRows * RSTR;
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int item = i * 256 + tid;
      const int r = item / (WS * 8), rem = item - r * (WS * 8);
      const int px = rem >> 3, q = rem & 7;
      *reinterpret_cast<float4*>(dst + r * RSTR + (px + 1) * kDgAst + q * 4) = pf[i];
    }
  };

  float e_bias[4], e_mul[4];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    e_bias[rr] = p.bias ? p.bias[4 * kq + rr] : 0.f;
    e_mul[rr] = p.epi_mode == BG_EPI_AFFINE_LRELU ? p.ref[4 * kq + rr] : 1.f;
  }

  // keep_elems is a multiple of 4 whenever it falls inside the tensor (whole samples of 16-channel pixels)
  const int epi_kind = (p.epi_mode == BG_EPI_NONE && !p.bias) ? 0 : (p.epi_mode == BG_EPI_MUL_GRAD ? 1 : 2);
  int strip = blockIdx.x;
  prefetch(strip);
  stash(0);
  __syncthreads();
  int buf = 0;
  for (; strip < p.nstrips; strip += gridDim.x, buf ^= 1) {
    prefetch(strip + gridDim.x);
    const int b = strip / p.strips_per_img, a0 = (strip - b * p.strips_per_img) * 2;
    const int a = a0 + ar;
    const float* rb = rows + buf * kDgRows * RSTR + (ar + 1) * RSTR + (1 + mh * (WS / 2) + li) * kDgAst + 4 * kq;
    const float* wb = wl + li * kDgAst + 4 * kq;
#pragma unroll
    for (int py = 0; py < 2; ++py) {
      floatx4 acc[2][MT];
#pragma unroll
      for (int px = 0; px < 2; ++px)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[px][mt] = floatx4{0.f, 0.f, 0.f, 0.f};
      // Both column phases in one pass: kernel columns 0, 2, 4 belong to px = 1, columns 1, 3 to px = 0, so taking them in pairs
      // (0, 1), (2, 3), (4) makes consecutive MFMAs write DIFFERENT accumulators -- with the px loop outside, the 32-column
      // variant issued 195 of its 200 MFMAs per strip as one dependent chain, each waiting for the one before (round 3).  Every
      // accumulator still sees its products in the order (kh, kw, channel half, k component): results unchanged.
#pragma unroll
      for (int kh = 0; kh < 5; ++kh) {
        if (((py + 1 - kh) & 1) != 0) continue;
        const int dy = (py + 1 - kh) / 2;
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          const int kwA = 2 * g, kwB = 2 * g + 1;             // px = 1 takes kwA (dx = (2 - kwA) / 2), px = 0 takes kwB (dx = (1 - kwB) / 2)
          const float* apA = rb + dy * RSTR + ((2 - kwA) / 2) * kDgAst;
          const float* bpA = wb + (kh * 5 + kwA) * 16 * kDgAst;
          const float* apB = rb + dy * RSTR + ((1 - kwB) / 2) * kDgAst;
          const float* bpB = wb + (kh * 5 + kwB) * 16 * kDgAst;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const float4 bvA = *reinterpret_cast<const float4*>(bpA + 16 * h);
            float4 avA[MT], avB[MT];
            float4 bvB = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) avA[mt] = *reinterpret_cast<const float4*>(apA + mt * 16 * kDgAst + 16 * h);
            if (kwB < 5) {
              bvB = *reinterpret_cast<const float4*>(bpB + 16 * h);
#pragma unroll
              for (int mt = 0; mt < MT; ++mt) avB[mt] = *reinterpret_cast<const float4*>(apB + mt * 16 * kDgAst + 16 * h);
            }
            // weights are the MFMA's row operand: D[i = channel][j = anchor], so a lane ends up with 4 consecutive channels
#define BG_C16_STEP(c)                                                                                                   \
  do {                                                                                                                    \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                                     \
        acc[1][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bvA.c, avA[mt].c, acc[1][mt], 0, 0, 0);                          \
    if (kwB < 5) {                                                                                                        \
      _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                                   \
          acc[0][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bvB.c, avB[mt].c, acc[0][mt], 0, 0, 0);                        \
    }                                                                                                                     \
  } while (0)
            BG_C16_STEP(x);
            BG_C16_STEP(y);
            BG_C16_STEP(z);
            BG_C16_STEP(w);
#undef BG_C16_STEP
          }
        }
      }
      // reg rr of lane (li, kq) = out[anchor column 16*mt + li][channel 4*kq + rr]: one float4 per lane, the 4 kq lanes of a
      // pixel write its 64 bytes, and the two px phases are neighbouring pixels
      {
        const size_t rowbase = ((size_t)b * p.Hd + 2 * a + py) * p.Wd;
        // critic data gradient: the reference values and mask bytes of ALL the lane's outputs are requested before the first is
        // used (unconditional loads: a mask that does not cover an element is read at element 0 and ignored).  One (mt, px) at a
        // time -- load, wait, load, wait, store -- cost the 16 -> 32 layer of the 128-pixel critic +63 % at 3 x 128 samples.
        float4 e_ref[MT][2];
        uchar4 e_keep[MT][2];
        if (epi_kind == 1) {
          const uint8_t* kptr = p.keep ? p.keep : reinterpret_cast<const uint8_t*>(p.ref);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int px = 0; px < 2; ++px) {
              const size_t idx = (rowbase + 2 * (mh * (WS / 2) + mt * 16 + li) + px) * 16 + 4 * kq;
              e_ref[mt][px] = *reinterpret_cast<const float4*>(p.ref + idx);
              e_keep[mt][px] = *reinterpret_cast<const uchar4*>(kptr + ((p.keep_elems == 0 || idx < p.keep_elems) ? idx : 0));
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int c = mh * (WS / 2) + mt * 16 + li;
#pragma unroll
          for (int px = 0; px < 2; ++px) {
            const size_t idx = (rowbase + 2 * c + px) * 16 + 4 * kq;
            const floatx4 v = acc[px][mt];
            float4 o;
            if (epi_kind == 0) {                               // no epilogue (ConvT -> BatchNorm): straight-line stores
              o = make_float4(v[0], v[1], v[2], v[3]);
            } else if (epi_kind == 1) {                        // critic data gradient: LeakyReLU' of the layer input, dropout mask
              const float4 r = e_ref[mt][px];
              float f0 = r.x > 0.f ? 1.f : p.alpha, f1 = r.y > 0.f ? 1.f : p.alpha, f2 = r.z > 0.f ? 1.f : p.alpha, f3 = r.w > 0.f ? 1.f : p.alpha;
              if (p.keep && (p.keep_elems == 0 || idx < p.keep_elems)) {
                const uchar4 k4 = e_keep[mt][px];
                f0 = k4.x ? f0 * p.scale : 0.f; f1 = k4.y ? f1 * p.scale : 0.f; f2 = k4.z ? f2 * p.scale : 0.f; f3 = k4.w ? f3 * p.scale : 0.f;
              }
              o = make_float4((v[0] + e_bias[0]) * f0, (v[1] + e_bias[1]) * f1, (v[2] + e_bias[2]) * f2, (v[3] + e_bias[3]) * f3);
            } else {
              o.x = bg::apply_epilogue_pre(p, v[0], idx + 0, e_bias[0], e_mul[0]);
              o.y = bg::apply_epilogue_pre(p, v[1], idx + 1, e_bias[1], e_mul[1]);
              o.z = bg::apply_epilogue_pre(p, v[2], idx + 2, e_bias[2], e_mul[2]);
              o.w = bg::apply_epilogue_pre(p, v[3], idx + 3, e_bias[3], e_mul[3]);
            }
            *reinterpret_cast<float4*>(p.C + idx) = o;
          }
        }
      }
    }
    stash(buf ^ 1);
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// forward, Ck = 16 input channels, N = 32, k = 5, s = 2, even H and W:
//   out[b, oy, ox, n] = sum_{kh, kw, c} x[b, 2oy + kh - 1, 2ox + kw - 1, c] * wT[kh*5 + kw][n][c]
// (also the data gradient of ConvT 32 -> 16).  A strip is 2 output rows = 7 input rows, staged de-interleaved by column parity
// (the 16 pixels of an MFMA tile are then 16 consecutive 20-float slots: conflict-free b128 reads).  No weight LDS: wave w works
// on output-channel tile (w & 1) and keeps its 25 weight fragments in registers; it owns output row (w >> 1) of the strip.
// ------------------------------------------------------------------------------------------------------------------------
constexpr int kFwRows = 7, kFwPst = 20;

template <int WO>
__global__ __launch_bounds__(256) void conv_c16_fwd_kernel(const C16Params p) {
  constexpr int W = 2 * WO, MT = WO / 16;
  constexpr int PS = (WO + 2) * kFwPst, RS = 2 * PS;
  constexpr int PF = kFwRows * W * 4 / 256;
  extern __shared__ __attribute__((aligned(16))) float c16_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int nt = wave & 1, r = wave >> 1;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, (int)p.a_bytes, 0x00020000);

  for (int i = tid; i < 2 * kFwRows * 2 * 2 * kFwPst; i += 256) {   // halo slots of every parity plane: zero, never written again
    const int e = i % kFwPst, side = (i / kFwPst) & 1, plane = i / (2 * kFwPst);
    c16_lds[plane * PS + (side ? (WO + 1) * kFwPst : 0) + e] = 0.f;
  }
  float4 wf[25];                                              // row operand: lane (li, kq) = wT[tap][16*nt + li][4*kq .. 4*kq+3]
#pragma unroll
  for (int t = 0; t < 25; ++t) wf[t] = *reinterpret_cast<const float4*>(p.Wt + ((size_t)t * 32 + nt * 16 + li) * 16 + 4 * kq);

  float4 pf[PF];
  auto prefetch = [&](int strip) {
    const int b = strip / p.strips_per_img, oy0 = (strip - b * p.strips_per_img) * 2;
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int item = i * 256 + tid;
      const int rr = item / (W * 4), rem = item - rr * (W * 4);
      const int y = 2 * oy0 - 1 + rr;
      const bool ok = strip < p.nstrips && (unsigned)y < (unsigned)p.Hs;
      pf[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsA, ok ? (unsigned)(((b * p.Hs + y) * W) * 64 + rem * 16) : kOob, 0, 0));
    }
  };
  auto stash = [&](int buf) {
    float* dst = c16_lds + buf * kFwRows * RS;
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int item = i * 256 + tid;
      const int rr = item / (W * 4), rem = item - rr * (W * 4);
      const int x = rem >> 2, q = rem & 3;
      *reinterpret_cast<float4*>(dst + rr * RS + (x & 1) * PS + ((x >> 1) + 1) * kFwPst + q * 4) = pf[i];
    }
  };

  float e_bias[4], e_mul[4];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    e_bias[rr] = p.bias ? p.bias[nt * 16 + 4 * kq + rr] : 0.f;
    e_mul[rr] = p.epi_mode == BG_EPI_AFFINE_LRELU ? p.ref[nt * 16 + 4 * kq + rr] : 1.f;
  }
  const int epi_kind = (p.epi_mode == BG_EPI_NONE && !p.bias) ? 0 : (p.epi_mode == BG_EPI_BIAS_LRELU ? 1 : 2);

  int strip = blockIdx.x;
  prefetch(strip);
  stash(0);
  __syncthreads();
  int buf = 0;
  for (; strip < p.nstrips; strip += gridDim.x, buf ^= 1) {
    prefetch(strip + gridDim.x);
    const int b = strip / p.strips_per_img, oy = (strip - b * p.strips_per_img) * 2 + r;
    const float* xb = c16_lds + buf * kFwRows * RS + (2 * r) * RS + (li + 1) * kFwPst + 4 * kq;
    floatx4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < 5; ++kh)
#pragma unroll
      for (int kw = 0; kw < 5; ++kw) {
        const int par = (kw + 1) & 1, fl = kw == 0 ? -1 : (kw - 1) / 2;
        const float4 wv = wf[kh * 5 + kw];
        // consecutive MFMAs go to DIFFERENT accumulators (k component outer, pixel tile inner): a chain of four on one accumulator
        // waits for each result in turn.  Every accumulator still sees its products in the same order: results unchanged.
        float4 xv[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xv[mt] = *reinterpret_cast<const float4*>(xb + kh * RS + par * PS + (mt * 16 + fl) * kFwPst);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.x, xv[mt].x, acc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.y, xv[mt].y, acc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.z, xv[mt].z, acc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.w, xv[mt].w, acc[mt], 0, 0, 0);
      }
    // reg rr of lane (li, kq) = out[pixel 16*mt + li][channel 16*nt + 4*kq + rr]
    const size_t rowbase = ((size_t)b * p.Hd + oy) * WO;
    uchar4 e_keep[MT];                                         // critic forward: all the mask bytes requested before the first is used
    if (epi_kind == 1 && p.keep) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const size_t idx = (rowbase + mt * 16 + li) * 32 + nt * 16 + 4 * kq;
        e_keep[mt] = *reinterpret_cast<const uchar4*>(p.keep + ((p.keep_elems == 0 || idx < p.keep_elems) ? idx : 0));
      }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const size_t idx = (rowbase + mt * 16 + li) * 32 + nt * 16 + 4 * kq;
      const floatx4 v = acc[mt];
      float4 o;
      if (epi_kind == 0) {
        o = make_float4(v[0], v[1], v[2], v[3]);
      } else if (epi_kind == 1) {                            // critic forward: bias, LeakyReLU, dropout mask
        float t0 = v[0] + e_bias[0], t1 = v[1] + e_bias[1], t2 = v[2] + e_bias[2], t3 = v[3] + e_bias[3];
        t0 = t0 > 0.f ? t0 : p.alpha * t0; t1 = t1 > 0.f ? t1 : p.alpha * t1; t2 = t2 > 0.f ? t2 : p.alpha * t2; t3 = t3 > 0.f ? t3 : p.alpha * t3;
        if (p.keep && (p.keep_elems == 0 || idx < p.keep_elems)) {
          const uchar4 k4 = e_keep[mt];
          t0 = k4.x ? t0 * p.scale : 0.f; t1 = k4.y ? t1 * p.scale : 0.f; t2 = k4.z ? t2 * p.scale : 0.f; t3 = k4.w ? t3 * p.scale : 0.f;
        }
        o = make_float4(t0, t1, t2, t3);
      } else {
        o.x = bg::apply_epilogue_pre(p, v[0], idx + 0, e_bias[0], e_mul[0]);
        o.y = bg::apply_epilogue_pre(p, v[1], idx + 1, e_bias[1], e_mul[1]);
        o.z = bg::apply_epilogue_pre(p, v[2], idx + 2, e_bias[2], e_mul[2]);
        o.w = bg::apply_epilogue_pre(p, v[3], idx + 3, e_bias[3], e_mul[3]);
      }
      *reinterpret_cast<float4*>(p.C + idx) = o;
    }
    stash(buf ^ 1);
    __syncthreads();
  }
}

void fill_epilogue(C16Params& p, const bg_epilogue* epi) {
  p.epi_mode = BG_EPI_NONE; p.alpha = 0.3f; p.scale = 1.f;
  if (epi) {
    p.epi_mode = epi->mode; p.bias = epi->bias; p.ref = epi->ref; p.keep = epi->keep; p.keep_elems = epi->keep_elems;
    p.alpha = epi->alpha; p.scale = epi->scale;
  }
}

// argument checks of the fused epilogues (only for shapes these kernels take)
int check_epilogue(const bg_epilogue* epi) {
  if (epi) {
    BG_REQUIRE(epi->mode >= BG_EPI_NONE && epi->mode <= BG_EPI_AFFINE_LRELU, BG_ERR_UNSUPPORTED, "conv c16: epilogue mode %d", epi->mode);
    BG_REQUIRE(epi->mode != BG_EPI_MUL_GRAD || epi->ref, BG_ERR_NULL, "conv c16: BG_EPI_MUL_GRAD needs ref");
    BG_REQUIRE(epi->mode != BG_EPI_AFFINE_LRELU || (epi->ref && epi->bias), BG_ERR_NULL, "conv c16: BG_EPI_AFFINE_LRELU needs ref and bias");
    // the epilogues work on 4 consecutive channels of a pixel: float4 loads of ref, uchar4 loads of the mask
    BG_REQUIRE(epi->mode != BG_EPI_MUL_GRAD || bg::aligned16(epi->ref), BG_ERR_BAD_ALIGNMENT, "conv c16: ref must be 16-byte aligned");
    BG_REQUIRE(!epi->keep || ((reinterpret_cast<uintptr_t>(epi->keep) & 3u) == 0 && epi->keep_elems % 4 == 0), BG_ERR_BAD_ALIGNMENT,
               "conv c16: keep mask must be 4-byte aligned and keep_elems a multiple of 4");
  }
  return BG_OK;
}

}  // namespace

namespace bg {

// 16-channel layers of the 128x128 stacks; *taken = 0 when the shape is not covered
int try_conv_c16(int bwd_data, const float* a, const float* w, float* c, int B, int H, int W, int Cin, int Cout, int k, int s,
                 const bg_epilogue* epi, void* stream, int* taken) {
  *taken = 0;
  static const int off = getenv("BG_NO_C16") ? 1 : 0;
  if (off || k != 5 || s != 2 || (H & 1) || (W & 1)) return BG_OK;
  C16Params p;
  memset(&p, 0, sizeof p);
  if (bwd_data) {      // dy [B,H/2,W/2,Cout] -> dx [B,H,W,Cin = 16]
    const int Hs = H / 2, Ws = W / 2;
    if (Cin != 16 || Cout != kDgCk || (Ws != 32 && Ws != 64) || (Hs & 1)) return BG_OK;
    if (int rc = check_epilogue(epi)) return rc;
    if ((size_t)B * H * W * 16 >= (1ull << 29) || (size_t)B * Hs * Ws * Cout >= (1ull << 29)) return BG_OK;
    p.A = a; p.Wt = w; p.C = c;
    p.B = B; p.Hs = Hs; p.Ws = Ws; p.Ck = Cout; p.Hd = H; p.Wd = W; p.N = 16;
    p.strips_per_img = Hs / 2;
    p.nstrips = B * p.strips_per_img;
    p.a_bytes = (unsigned)((size_t)B * Hs * Ws * Cout * sizeof(float));
    fill_epilogue(p, epi);
    const size_t lds = ((size_t)25 * 16 * kDgAst + (size_t)2 * kDgRows * (Ws + 2) * kDgAst) * sizeof(float);
    const dim3 grid((unsigned)std::min(p.nstrips, 256));
    const double flops = 2.0 * B * (double)Hs * Ws * Cin * Cout * 25;
    Launch L(stream, "conv_c16_dgrad", flops, 0);
    if (Ws == 64) {
      BG_LDS_ATTR_ONCE_V(conv_c16_dgrad_kernel<64>, 150 * 1024);
      bg::launch((conv_c16_dgrad_kernel<64>), grid, dim3(256), lds, L.s, p);
    } else {
      BG_LDS_ATTR_ONCE_V(conv_c16_dgrad_kernel<32>, 150 * 1024);
      bg::launch((conv_c16_dgrad_kernel<32>), grid, dim3(256), lds, L.s, p);
    }
    *taken = 1;
    return L.done("conv_c16_dgrad_kernel");
  }
  // forward: x [B,H,W,16] -> y [B,H/2,W/2,32]
  {
    const int Ho = H / 2, Wo = W / 2;
    if (Cin != 16 || Cout != 32 || (Wo != 32 && Wo != 64) || (Ho & 1)) return BG_OK;
    if (int rc = check_epilogue(epi)) return rc;
    if ((size_t)B * H * W * 16 >= (1ull << 29) || (size_t)B * Ho * Wo * 32 >= (1ull << 29)) return BG_OK;
    p.A = a; p.Wt = w; p.C = c;
    p.B = B; p.Hs = H; p.Ws = W; p.Ck = 16; p.Hd = Ho; p.Wd = Wo; p.N = 32;
    p.strips_per_img = Ho / 2;
    p.nstrips = B * p.strips_per_img;
    p.a_bytes = (unsigned)((size_t)B * H * W * 16 * sizeof(float));
    fill_epilogue(p, epi);
    const size_t lds = (size_t)2 * kFwRows * 2 * (Wo + 2) * kFwPst * sizeof(float);
    const dim3 grid((unsigned)std::min(p.nstrips, 256));
    const double flops = 2.0 * B * (double)Ho * Wo * Cin * Cout * 25;
    Launch L(stream, "conv_c16_fwd", flops, 0);
    BG_LDS_ATTR_ONCE_V(conv_c16_fwd_kernel<64>, 150 * 1024);
    BG_LDS_ATTR_ONCE_V(conv_c16_fwd_kernel<32>, 150 * 1024);
    if (Wo == 64) bg::launch((conv_c16_fwd_kernel<64>), grid, dim3(256), lds, L.s, p);
    else bg::launch((conv_c16_fwd_kernel<32>), grid, dim3(256), lds, L.s, p);
    *taken = 1;
    return L.done("conv_c16_fwd_kernel");
  }
}

}  // namespace bg
