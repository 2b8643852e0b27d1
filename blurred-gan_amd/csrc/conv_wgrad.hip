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
  int tiles_n;
  float beta, scale; // applied only when ksplit == 1
};

template <int BM, int BN, int BKP, int WAVES_M, int WAVES_N, int WAVES_K>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradParams p) {
  static_assert(WAVES_M * WAVES_N * WAVES_K == 4, "4 waves");
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int MI = WTM / 32, NI = WTN / 32;
  static_assert(MI >= 1 && NI >= 1, "wave tile");
  constexpr int TPR_A = BM / 4, RPP_A = 256 / TPR_A, AP = (BKP + RPP_A - 1) / RPP_A;
  constexpr int TPR_B = BN / 4, RPP_B = 256 / TPR_B, BP = (BKP + RPP_B - 1) / RPP_B;
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
  float4 regA[AP], regB[BP];

  auto gload = [&](int step) {
    const int mb = m_begin + step * BKP;
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const int r = arow + i * RPP_A;
      const int m = mb + r;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < BKP && m < m_end && ci0 + aq * 4 < p.Ci) {
        const int b = m / HoWo;
        const int rem = m - b * HoWo;
        const int oh = rem / p.Wo;
        const int ow = rem - oh * p.Wo;
        const int iy = oh * p.s + kh - p.pt, ix = ow * p.s + kw - p.pl;
        if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
          v = *reinterpret_cast<const float4*>(p.X + ((size_t)(b * p.H + iy) * p.W + ix) * p.Ci + ci0 + aq * 4);
      }
      regA[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      const int r = brow + i * RPP_B;
      const int m = mb + r;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < BKP && m < m_end && co0 + bq * 4 < p.Co) v = *reinterpret_cast<const float4*>(p.DY + (size_t)m * p.Co + co0 + bq * 4);
      regB[i] = v;
    }
  };
  auto lstore = [&](int buf) {
    float* sa = smem + buf * STAGE;
    float* sb = sa + BKP * BM;
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const int r = arow + i * RPP_A;
      if (r < BKP) *reinterpret_cast<float4*>(sa + r * BM + aq * 4) = regA[i];
    }
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      const int r = brow + i * RPP_B;
      if (r < BKP) *reinterpret_cast<float4*>(sb + r * BN + bq * 4) = regB[i];
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

  // ---- store: acc reg r of lane l holds C[row(ci) = (r&3) + 8*(r>>2) + 4*(l>>5)][col(co) = l&31]
  float* out = p.out + (size_t)blockIdx.z * p.k * p.k * p.Ci * p.Co + (size_t)tap * p.Ci * p.Co;
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

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw, int n,
                                                           int ksplit, float beta, float scale) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  float s = 0.f;
  for (int z = 0; z < ksplit; ++z) s += slabs[(size_t)z * n + e];
  dw[e] = (beta != 0.f ? beta * dw[e] : 0.f) + scale * s;
}

struct WgradPlan {
  int mode;     // 0 direct, 1..: mfma config id
  int ksplit, chunk, tiles_m, tiles_n, bkp;
};

WgradPlan plan_wgrad(int B, int H, int W, int Ci, int Co, int k, int s) {
  WgradPlan pl{};
  int Ho, Wo, pt, pp;
  bg::same_pads(H, k, s, &Ho, &pt);
  bg::same_pads(W, k, s, &Wo, &pp);
  const long M = (long)B * Ho * Wo;
  const bool mfma = (Ci % 4 == 0) && (Co % 4 == 0) && Ci >= 16 && Co >= 16;
  int bm = 0, bn = 0;
  if (!mfma) {
    pl.mode = 0;
    const long nblk = bg::cdiv((size_t)k * k * Ci * Co, 256);
    long want = std::max(1L, 2048 / nblk);
    long chunk = std::max(256L, (M + want - 1) / want);
    pl.chunk = (int)chunk;
    pl.ksplit = (int)((M + chunk - 1) / chunk);
    return pl;
  }
  if (Ci > 64 && Co > 64) { pl.mode = 1; bm = 128; bn = 128; pl.bkp = 32; }
  else if (Ci > 32 && Co > 32) { pl.mode = 2; bm = 64; bn = 64; pl.bkp = 32; }
  else if (Ci > 32) { pl.mode = 3; bm = 64; bn = 32; pl.bkp = 64; }
  else if (Co > 32) { pl.mode = 4; bm = 32; bn = 64; pl.bkp = 64; }
  else { pl.mode = 5; bm = 32; bn = 32; pl.bkp = 128; }
  pl.tiles_m = bg::cdiv(Ci, bm);
  pl.tiles_n = bg::cdiv(Co, bn);
  const long base = (long)pl.tiles_m * pl.tiles_n * k * k;
  long want = std::max(1L, (1024 + base - 1) / base);           // aim for >= 1024 workgroups
  long steps = (M + pl.bkp - 1) / pl.bkp;
  want = std::min(want, std::max(1L, steps / 4));               // at least 4 K-steps per workgroup
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
  if (pl.ksplit <= 1) return 0;
  return (size_t)pl.ksplit * ksize * ksize * Cin * Cout * sizeof(float);
}

int bg_conv2d_bwd_filter(const float* x, const float* dy, float* dw, int B, int H, int W, int Cin, int Cout, int ksize,
                         int stride, float beta, float scale, void* ws_d, size_t ws_bytes, void* stream) {
  BG_REQUIRE(x && dy && dw, BG_ERR_NULL, "bg_conv2d_bwd_filter: null pointer");
  BG_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, BG_ERR_BAD_SHAPE, "bg_conv2d_bwd_filter: B=%d H=%d W=%d Cin=%d Cout=%d", B, H, W, Cin, Cout);
  BG_REQUIRE(ksize >= 1 && (ksize & 1) && ksize * ksize <= bg::kMaxTaps, BG_ERR_UNSUPPORTED, "bg_conv2d_bwd_filter: kernel size %d", ksize);
  BG_REQUIRE(stride == 1 || stride == 2, BG_ERR_UNSUPPORTED, "bg_conv2d_bwd_filter: stride %d", stride);
  BG_REQUIRE(bg::aligned16(x) && bg::aligned16(dy) && bg::aligned16(dw), BG_ERR_BAD_ALIGNMENT, "bg_conv2d_bwd_filter: pointers must be 16-byte aligned");
  WgradPlan pl = plan_wgrad(B, H, W, Cin, Cout, ksize, stride);
  const size_t nout = (size_t)ksize * ksize * Cin * Cout;
  const size_t need = pl.ksplit > 1 ? (size_t)pl.ksplit * nout * sizeof(float) : 0;
  BG_REQUIRE(need == 0 || (ws_d && ws_bytes >= need), BG_ERR_WORKSPACE, "bg_conv2d_bwd_filter: workspace %zu bytes < %zu needed", ws_bytes, need);
  WgradParams p;
  memset(&p, 0, sizeof p);
  p.X = x; p.DY = dy;
  p.B = B; p.H = H; p.W = W; p.Ci = Cin; p.Co = Cout; p.k = ksize; p.s = stride;
  bg::same_pads(H, ksize, stride, &p.Ho, &p.pt);
  bg::same_pads(W, ksize, stride, &p.Wo, &p.pl);
  p.M = B * p.Ho * p.Wo;
  BG_REQUIRE((size_t)B * H * W * (size_t)Cin < (1ull << 31) && (size_t)p.M * (size_t)Cout < (1ull << 31), BG_ERR_UNSUPPORTED,
             "bg_conv2d_bwd_filter: tensor exceeds 2^31 elements");
  p.chunk = pl.chunk; p.ksplit = pl.ksplit; p.tiles_n = pl.tiles_n;
  p.beta = beta; p.scale = scale;
  p.out = pl.ksplit > 1 ? static_cast<float*>(ws_d) : dw;
  const double flops = 2.0 * p.M * (double)nout;
  int rc;
  if (pl.mode == 0) {
    bg::Launch L(stream, "conv_wgrad_direct", flops, 0);
    hipLaunchKernelGGL(wgrad_direct_kernel, dim3(bg::cdiv(nout, 256), pl.ksplit), dim3(256), 0, L.s, p);
    rc = L.done("wgrad_direct_kernel");
  } else {
    dim3 grid(pl.tiles_m * pl.tiles_n, ksize * ksize, pl.ksplit);
    bg::Launch L(stream, "conv_wgrad_mfma", flops, 0);
    switch (pl.mode) {
      case 1: hipLaunchKernelGGL((conv_wgrad_kernel<128, 128, 32, 2, 2, 1>), grid, dim3(256), 0, L.s, p); break;
      case 2: hipLaunchKernelGGL((conv_wgrad_kernel<64, 64, 32, 2, 2, 1>), grid, dim3(256), 0, L.s, p); break;
      case 3: hipLaunchKernelGGL((conv_wgrad_kernel<64, 32, 64, 2, 1, 2>), grid, dim3(256), 0, L.s, p); break;
      case 4: hipLaunchKernelGGL((conv_wgrad_kernel<32, 64, 64, 1, 2, 2>), grid, dim3(256), 0, L.s, p); break;
      default: hipLaunchKernelGGL((conv_wgrad_kernel<32, 32, 128, 1, 1, 4>), grid, dim3(256), 0, L.s, p); break;
    }
    rc = L.done("conv_wgrad_kernel");
  }
  if (rc) return rc;
  if (pl.ksplit > 1) {
    bg::Launch L(stream, "conv_wgrad_reduce", 0, (double)(pl.ksplit + 1) * nout * 4);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(bg::cdiv(nout, 256)), dim3(256), 0, L.s, static_cast<const float*>(ws_d), dw,
                       (int)nout, pl.ksplit, beta, scale);
    rc = L.done("wgrad_reduce_kernel");
  }
  return rc;
}

}  // extern "C"
