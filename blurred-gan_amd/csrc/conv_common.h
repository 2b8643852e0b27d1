// Gather-GEMM description shared by the conv forward / data-gradient kernels.
//
// Both TF ops reduce to   C[m, n] = sum_{tap, c}  A[src(m, tap), c] * Wt[wi(tap)][n][c]
// with m running over an "anchor" grid (b, a, bx) of one phase:
//   forward  (Conv2D, demo_celeba.py:99-119):  src = (a*s + kh - pt, bx*s + kw - pl), dst = (a, bx), 1 phase
//   data-grad / Conv2DTranspose forward (demo_celeba.py:62-87): sub-pixel decomposition, s*s phases,
//            phase (py,px): dst = (a*s + py, bx*s + px); only taps with (py + pt - kh) % s == 0 contribute,
//            src = (a + (py+pt-kh)/s, bx + (px+pl-kw)/s)  -- no zero-insertion, no wasted MACs.
// TF 'SAME': out = ceil(in/s), pad_total = max((out-1)s + k - in, 0), pad_before = pad_total/2.
#pragma once
#include "common.h"

namespace bg {

constexpr int kMaxTaps = 25;
constexpr int kMaxPhases = 4;

struct GatherPhase {
  int Ha, Wa;          // anchor grid of this phase
  int py, px;          // destination offset
  int ntaps;
  int tap[kMaxTaps];   // (dy+64) | (dx+64)<<8 | wi<<16
};


struct GatherParams {
  const float* A;      // [B][Hs][Ws][Ck]
  const float* Wt;     // [taps][N][Ck]
  float* C;            // [B][Hd][Wd][N]
  int B, Hs, Ws, Ck;
  int Hd, Wd, N;
  int ss, ds;          // source / destination stride multipliers
  unsigned a_bytes, w_bytes;   // operand sizes for the buffer descriptors of the MFMA kernel
  int nphase;
  int mtiles, xcd_swizzle;   // MFMA kernel: M tiles per phase, XCD-aware tile order on/off
  int pos_major;       // M index order: 0 = (b, a, bx) image-major; 1 = (a, bx, b) position-major (small feature maps: the rows of
                       // a tile then share their spatial position, so zero-padding taps are skipped for the whole tile)
  int pmerge;          // MFMA kernel: sub-pixel phases handled by ONE workgroup (1, 2 or 4): the K loop runs through their tap lists
                       // back to back and each phase's tile is stored when its taps are done (short-K transposed convs)
  int ksplit;          // split-K factor of the MFMA kernel (1 = none); partial sums go to slab[split][B*Hd*Wd*N]
  float* slab;
  float* stats;        // MFMA kernel: per-workgroup column sums / sums of squares of the stored tile, [row][2][N]; null = off
  // MFMA kernel, position-major launches: cost-sorted tile order.  Tiles at different output positions run 9 to 16 of their 25
  // taps, and the hardware places workgroup w of a launch on XCD w % 8 and, within one round of resident workgroups, on the same
  // CU as w + 256, w + 512, w + 768 (tools/probes/placement.hip).  (phase group, position) pairs are sorted by live taps,
  // heaviest first, and dealt to the CUs in a snake (even rounds forward, odd rounds backward): every CU's co-resident workgroups
  // then add up to the same number of K steps, and multi-round grids start their longest workgroups first.
  unsigned char grp_order[4];      // plain order: phase groups by descending tap count (grid.z walks them heaviest first, so a multi-round
                                   // grid ends on its SHORT workgroups: the 4-tap phase of a stride-2 transposed conv, not the 9-tap one)
  int order_n;         // pairs in pp_order (0 = plain (n tile, m tile) x (phase, split) order)
  int m_fast;          // sorted order, inside a pair: 0 = N tile fastest (an XCD keeps ONE weight panel: weight-heavy layers),
                       // 1 = batch slice fastest (the N tiles of a slice follow each other on ONE XCD and hit its L2 for the
                       // activation rows: layers whose weights fit an L2 and whose activations are the bigger operand)
  int per_pair;        // workgroups per pair = ksplit * N tiles * (B / BM)
  unsigned char pp_order[256];     // (phase group << 6) | position, by descending cost
  // epilogue
  int epi_mode;
  const float* bias;
  const float* ref;
  const uint8_t* keep;
  size_t keep_elems;   // the mask covers output elements [0, keep_elems); 0 = all
  float alpha, scale;
  GatherPhase ph[kMaxPhases];
};

__host__ __device__ inline int tap_dy(int t) { return (t & 0xff) - 64; }
__host__ __device__ inline int tap_dx(int t) { return ((t >> 8) & 0xff) - 64; }
__host__ __device__ inline int tap_wi(int t) { return t >> 16; }
inline int pack_tap(int dy, int dx, int wi) { return (dy + 64) | ((dx + 64) << 8) | (wi << 16); }

inline void same_pads(int n, int k, int s, int* out, int* before) {
  const int o = (n + s - 1) / s;
  int tot = (o - 1) * s + k - n;
  if (tot < 0) tot = 0;
  *out = o;
  *before = tot / 2;
}

// H, W, Cin: conv input side; Cout: conv output side.
inline int make_fwd_params(GatherParams& p, int B, int H, int W, int Cin, int Cout, int k, int s) {
  int Ho, Wo, pt, pl;
  same_pads(H, k, s, &Ho, &pt);
  same_pads(W, k, s, &Wo, &pl);
  p.B = B; p.Hs = H; p.Ws = W; p.Ck = Cin;
  p.Hd = Ho; p.Wd = Wo; p.N = Cout;
  p.ss = s; p.ds = 1; p.nphase = 1;
  GatherPhase& g = p.ph[0];
  g.Ha = Ho; g.Wa = Wo; g.py = 0; g.px = 0; g.ntaps = 0;
  for (int kh = 0; kh < k; ++kh)
    for (int kw = 0; kw < k; ++kw) g.tap[g.ntaps++] = pack_tap(kh - pt, kw - pl, kh * k + kw);
  return 0;
}

inline int make_bwd_data_params(GatherParams& p, int B, int H, int W, int Cin, int Cout, int k, int s) {
  int Ho, Wo, pt, pl;
  same_pads(H, k, s, &Ho, &pt);
  same_pads(W, k, s, &Wo, &pl);
  p.B = B; p.Hs = Ho; p.Ws = Wo; p.Ck = Cout;
  p.Hd = H; p.Wd = W; p.N = Cin;
  p.ss = 1; p.ds = s; p.nphase = 0;
  for (int py = 0; py < s; ++py)
    for (int px = 0; px < s; ++px) {
      GatherPhase& g = p.ph[p.nphase++];
      g.py = py; g.px = px;
      g.Ha = (H - py + s - 1) / s;
      g.Wa = (W - px + s - 1) / s;
      g.ntaps = 0;
      for (int kh = 0; kh < k; ++kh) {
        if ((py + pt - kh) % s != 0) continue;
        for (int kw = 0; kw < k; ++kw) {
          if ((px + pl - kw) % s != 0) continue;
          g.tap[g.ntaps++] = pack_tap((py + pt - kh) / s, (px + pl - kw) / s, kh * k + kw);
        }
      }
    }
  return 0;
}

// device helpers -------------------------------------------------------------------------------
// XCD-aware work order: hardware deals workgroups round-robin over the 8 XCDs (each with a private 4 MiB L2), so
// blocks b and b+8 share an L2.  This bijection hands every XCD one contiguous run of the logical work list, so
// neighbours in that list (which share an operand panel) meet in the same L2.  Speed only, never correctness.
__device__ inline int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7, slot = bid >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

struct RowAnchor {
  int b, ay, ax;  // image index, anchor*ss (source-space origin); ay = INT_MIN/2 marks an invalid row
};

__device__ inline void decode_row(const GatherParams& p, const GatherPhase& g, int m, int Mph, RowAnchor& r, int& dst) {
  if (m < Mph) {
    const int hw = g.Ha * g.Wa;
    int b, rem;
    if (p.pos_major) { rem = m / p.B; b = m - rem * p.B; }
    else { b = m / hw; rem = m - b * hw; }
    const int a = rem / g.Wa;
    const int bx = rem - a * g.Wa;
    r.b = b; r.ay = a * p.ss; r.ax = bx * p.ss;
    dst = (b * p.Hd + a * p.ds + g.py) * p.Wd + bx * p.ds + g.px;
  } else {
    r.b = 0; r.ay = -(1 << 28); r.ax = 0;
    dst = -1;
  }
}

// P: any parameter block with the epilogue fields (epi_mode, bias, ref, keep, alpha, scale)
template <class P>
__device__ inline float apply_epilogue(const P& p, float v, size_t idx, int n) {
  if (p.epi_mode == BG_EPI_AFFINE_LRELU) {             // folded inference BatchNorm + LeakyReLU
    v = fmaf(v, p.ref[n], p.bias[n]);
    return v > 0.f ? v : p.alpha * v;
  }
  if (p.bias) v += p.bias[n];
  switch (p.epi_mode) {
    case BG_EPI_BIAS_LRELU:
      v = v > 0.f ? v : p.alpha * v;
      if (p.keep && (p.keep_elems == 0 || idx < p.keep_elems)) v = p.keep[idx] ? v * p.scale : 0.f;
      break;
    case BG_EPI_MUL_GRAD: {
      float f = p.ref[idx] > 0.f ? 1.f : p.alpha;
      if (p.keep && (p.keep_elems == 0 || idx < p.keep_elems)) f = p.keep[idx] ? f * p.scale : 0.f;
      v *= f;
      break;
    }
    case BG_EPI_TANH:
      v = tanhf(v);
      break;
    default:
      break;
  }
  return v;
}

// same, with the per-column operands (bias, folded BatchNorm scale) already in registers
template <class P>
__device__ inline float apply_epilogue_pre(const P& p, float v, size_t idx, float bias_n, float mul_n) {
  if (p.epi_mode == BG_EPI_AFFINE_LRELU) {
    v = fmaf(v, mul_n, bias_n);
    return v > 0.f ? v : p.alpha * v;
  }
  v += bias_n;
  switch (p.epi_mode) {
    case BG_EPI_BIAS_LRELU:
      v = v > 0.f ? v : p.alpha * v;
      if (p.keep && (p.keep_elems == 0 || idx < p.keep_elems)) v = p.keep[idx] ? v * p.scale : 0.f;
      break;
    case BG_EPI_MUL_GRAD: {
      float f = p.ref[idx] > 0.f ? 1.f : p.alpha;
      if (p.keep && (p.keep_elems == 0 || idx < p.keep_elems)) f = p.keep[idx] ? f * p.scale : 0.f;
      v *= f;
      break;
    }
    case BG_EPI_TANH:
      v = tanhf(v);
      break;
    default:
      break;
  }
  return v;
}

// Four consecutive output channels of one pixel at once (16-B aligned: N % 4 == 0, idx % 4 == 0): the same arithmetic as
// apply_epilogue_pre per component, with the per-element operands (LeakyReLU-gradient reference, dropout mask) fetched as one
// float4 and one 4-byte word instead of four loads each.
template <class P>
__device__ inline float4 apply_epilogue4(const P& p, float4 v, size_t idx, const float* bias4, const float* mul4) {
  const float4 b = *reinterpret_cast<const float4*>(bias4);
  if (p.epi_mode == BG_EPI_AFFINE_LRELU) {
    const float4 m = *reinterpret_cast<const float4*>(mul4);
    v.x = fmaf(v.x, m.x, b.x); v.y = fmaf(v.y, m.y, b.y); v.z = fmaf(v.z, m.z, b.z); v.w = fmaf(v.w, m.w, b.w);
    v.x = v.x > 0.f ? v.x : p.alpha * v.x; v.y = v.y > 0.f ? v.y : p.alpha * v.y;
    v.z = v.z > 0.f ? v.z : p.alpha * v.z; v.w = v.w > 0.f ? v.w : p.alpha * v.w;
    return v;
  }
  v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
  const bool has_mask = p.keep != nullptr;
  switch (p.epi_mode) {
    case BG_EPI_BIAS_LRELU: {
      v.x = v.x > 0.f ? v.x : p.alpha * v.x; v.y = v.y > 0.f ? v.y : p.alpha * v.y;
      v.z = v.z > 0.f ? v.z : p.alpha * v.z; v.w = v.w > 0.f ? v.w : p.alpha * v.w;
      if (has_mask) {
        const uchar4 k = *reinterpret_cast<const uchar4*>(p.keep + ((p.keep_elems == 0 || idx < p.keep_elems) ? idx : 0));
        if (p.keep_elems == 0 || idx + 0 < p.keep_elems) v.x = k.x ? v.x * p.scale : 0.f;
        if (p.keep_elems == 0 || idx + 1 < p.keep_elems) v.y = k.y ? v.y * p.scale : 0.f;
        if (p.keep_elems == 0 || idx + 2 < p.keep_elems) v.z = k.z ? v.z * p.scale : 0.f;
        if (p.keep_elems == 0 || idx + 3 < p.keep_elems) v.w = k.w ? v.w * p.scale : 0.f;
      }
      break;
    }
    case BG_EPI_MUL_GRAD: {
      const float4 r = *reinterpret_cast<const float4*>(p.ref + idx);
      float f0 = r.x > 0.f ? 1.f : p.alpha, f1 = r.y > 0.f ? 1.f : p.alpha, f2 = r.z > 0.f ? 1.f : p.alpha, f3 = r.w > 0.f ? 1.f : p.alpha;
      if (has_mask) {
        const uchar4 k = *reinterpret_cast<const uchar4*>(p.keep + ((p.keep_elems == 0 || idx < p.keep_elems) ? idx : 0));
        if (p.keep_elems == 0 || idx + 0 < p.keep_elems) f0 = k.x ? f0 * p.scale : 0.f;
        if (p.keep_elems == 0 || idx + 1 < p.keep_elems) f1 = k.y ? f1 * p.scale : 0.f;
        if (p.keep_elems == 0 || idx + 2 < p.keep_elems) f2 = k.z ? f2 * p.scale : 0.f;
        if (p.keep_elems == 0 || idx + 3 < p.keep_elems) f3 = k.w ? f3 * p.scale : 0.f;
      }
      v.x *= f0; v.y *= f1; v.z *= f2; v.w *= f3;
      break;
    }
    case BG_EPI_TANH:
      v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w);
      break;
    default:
      break;
  }
  return v;
}

// conv_rows.hip: row-MFMA kernel for thin-N forward (stride 1) / data gradient; *taken = 0 when the shape is not covered
int try_conv_rows(int bwd_data, const float* a, const float* w, float* c, int B, int H, int W, int Cin, int Cout, int k, int s,
                  const bg_epilogue* epi, void* stream, int* taken);

// conv_c16.hip: row-staged kernels for the 16-channel layers of the 128x128 stacks (k = 5, s = 2)
int try_conv_c16(int bwd_data, const float* a, const float* w, float* c, int B, int H, int W, int Cin, int Cout, int k, int s,
                 const bg_epilogue* epi, void* stream, int* taken);

int try_conv_rows_gather(int bwd_data, const float* a, const float* w, float* c, int B, int H, int W, int Cin, int Cout, int k, int s,
                         const bg_epilogue* epi, void* stream, int* taken);

// Rows of float4 from global memory into LDS with SU loads in flight per thread (256 threads).  `src(r, c4)` returns the address of
// float4 c4 of row r or nullptr (zero padding / halo).  The plain form -- one load, one LDS write per iteration -- waits for every
// load before the next is issued: a chain of global round trips per staged block (round 5: that chain, not the matrix work, was
// most of a workgroup's time in the thin kernels).
template <int SU, typename SrcFn>
__device__ __forceinline__ void stage_rows_f4(float* lds, int nrows, int q4, int row_stride, int tid, SrcFn src) {
  const int tot = nrows * q4;
  for (int i0 = tid; i0 < tot; i0 += 256 * SU) {
    float4 v[SU];
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int idx = i0 + u * 256;
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < tot) {
        const int r = idx / q4, c4 = idx - r * q4;
        const float* s = src(r, c4);
        if (s) v[u] = *reinterpret_cast<const float4*>(s);
      }
    }
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int idx = i0 + u * 256;
      if (idx < tot) {
        const int r = idx / q4, c4 = idx - r * q4;
        *reinterpret_cast<float4*>(lds + (size_t)r * row_stride + c4 * 4) = v[u];
      }
    }
  }
}


}  // namespace bg
