// HBM-bound support ops of the training step: Dense, column reductions, BatchNorm(+LeakyReLU),
// pointwise GP helpers, losses, Adam, RNG.  Each replaces the TF op named beside it
// (SURVEY.md 8a rows T4-T12); all are bandwidth-bound, written as coalesced grid-stride kernels.
#include "common.h"
#include <algorithm>
#include <cmath>

namespace {

__device__ inline bool bg_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

constexpr int kT = 256;

inline unsigned grid_for(size_t n, int per_thread = 1) {
  return (unsigned)std::max<size_t>(1, std::min<size_t>(bg::cdiv(n, (size_t)kT * per_thread), 256 * 8));
}

// ---------------------------------------------------------------- Dense (layers.Dense, demo_celeba.py:55,124)
__global__ __launch_bounds__(kT) void gemm_naive_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                        float* __restrict__ C, int M, int N, int K, int transA, int transB,
                                                        const float* __restrict__ bias, float beta, float scale) {
  const size_t total = (size_t)M * N;
  for (size_t e = (size_t)blockIdx.x * kT + threadIdx.x; e < total; e += (size_t)gridDim.x * kT) {
    const int m = (int)(e / N), n = (int)(e - (size_t)m * N);
    float acc = 0.f;
    for (int k = 0; k < K; ++k) {
      const float a = transA ? A[(size_t)k * M + m] : A[(size_t)m * K + k];
      const float b = transB ? Bm[(size_t)n * K + k] : Bm[(size_t)k * N + n];
      acc = fmaf(a, b, acc);
    }
    float v = scale * acc;
    if (bias) v += bias[n];
    if (beta != 0.f) v += beta * C[e];
    C[e] = v;
  }
}

// LDS-tiled SGEMM on the matrix cores, 64x64 tile, 32-deep steps (generator's Dense(100 -> 8192) and its weight gradient).
// Small problem (0.4 GFLOP, K = 100 or the batch): what it waits for is the global -> LDS round trip of every step, so the
// steps are deep (4 round trips at K = 100 instead of 7) and the NEXT tile's loads are issued before the current tile's
// products (register staging).  Round 3: the products of a step run on v_mfma_f32_32x32x2_f32 (exact fp32; one 32x32 tile per
// wave, fragments read k-major from LDS: lanes along m / n are consecutive words, conflict-free) instead of 4x4 FMA blocks per
// thread: 22 -> see DESIGN.md section 7.
typedef float gemm_floatx16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(kT) void gemm_tiled_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                        float* __restrict__ C, int M, int N, int K, int transA, int transB,
                                                        const float* __restrict__ bias, float beta, float scale) {
  constexpr int KT = 32, NL = 64 * KT / kT;                   // 8 elements per thread per operand tile
  __shared__ float As[KT][64 + 4], Bs[KT][64 + 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, kk = lane >> 5, wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  gemm_floatx16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  float ra[NL], rb[NL];
  auto gload = [&](int k0) {                                  // A tile -> As[k][m]; B tile -> Bs[k][n]; contiguous index fastest
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int e = threadIdx.x + i * kT;
      int ka, mm;
      if (transA) { ka = e >> 6; mm = e & 63; } else { mm = e / KT; ka = e - mm * KT; }
      const int m = m0 + mm, k = k0 + ka;
      ra[i] = (m < M && k < K) ? (transA ? A[(size_t)k * M + m] : A[(size_t)m * K + k]) : 0.f;
      int kb, nn;
      if (transB) { nn = e / KT; kb = e - nn * KT; } else { kb = e >> 6; nn = e & 63; }
      const int n = n0 + nn, k2 = k0 + kb;
      rb[i] = (n < N && k2 < K) ? (transB ? Bm[(size_t)n * K + k2] : Bm[(size_t)k2 * N + n]) : 0.f;
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int e = threadIdx.x + i * kT;
      int ka, mm;
      if (transA) { ka = e >> 6; mm = e & 63; } else { mm = e / KT; ka = e - mm * KT; }
      As[ka][mm] = ra[i];
      int kb, nn;
      if (transB) { nn = e / KT; kb = e - nn * KT; } else { kb = e >> 6; nn = e & 63; }
      Bs[kb][nn] = rb[i];
    }
  };
  gload(0);
  for (int k0 = 0; k0 < K; k0 += KT) {
    lstore();
    __syncthreads();
    if (k0 + KT < K) gload(k0 + KT);                          // in flight under this tile's products
    const float* ap = &As[kk][wm * 32 + li];                  // + 2p rows: A[m = li][k = 2p + kk]
    const float* bp = &Bs[kk][wn * 32 + li];
#pragma unroll
    for (int p = 0; p < KT / 2; ++p) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * p * 68], bp[2 * p * 68], acc, 0, 0, 0);
    __syncthreads();
  }
  // acc[q] of lane (li, kk) = C[row (q & 3) + 8 (q >> 2) + 4 kk][col li] of the wave's 32x32 tile
  const int n = n0 + wn * 32 + li;
  if (n < N) {
    const float bn = bias ? bias[n] : 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int m = m0 + wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * kk;
      if (m >= M) continue;
      float v = scale * acc[q] + bn;
      const size_t e = (size_t)m * N + n;
      if (beta != 0.f) v += beta * C[e];
      C[e] = v;
    }
  }
}

// N == 1, transA: out[m] = sum_k A[k][m] * b[k]  (weight gradient of the critic's Dense(2048 -> 1): A = activations [B, 2048]).
// Block = 64 columns x 16 k-lanes, LDS tree over the k-lanes; coalesced along m.
__global__ __launch_bounds__(1024) void gemv_t_kernel(const float* __restrict__ A, const float* __restrict__ b, float* __restrict__ C,
                                                      int M, int K, const float* __restrict__ bias, float beta, float scale) {
  __shared__ float red[16][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int m = blockIdx.x * 64 + tx;
  float acc = 0.f;
  if (m < M)
    for (int k = ty; k < K; k += 16) acc = fmaf(A[(size_t)k * M + m], b[k], acc);
  red[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && m < M) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += red[i][tx];
    float v = scale * s;
    if (bias) v += bias[0];
    if (beta != 0.f) v += beta * C[m];
    C[m] = v;
  }
}

// N == 1, no transposes (critic's Dense(2048 -> 1); 6272 -> 1 for MNIST): ONE WORKGROUP per row, float4 loads, every load of a
// thread independent.  (One wave per row left 192 waves on 256 CUs for the MNIST critic and a 98-deep dependent load / FMA
// chain per lane: 31 us for 4.8 MB.)
__global__ __launch_bounds__(kT) void rowdot_kernel(const float* __restrict__ A, const float* __restrict__ w, float* __restrict__ C,
                                                    int M, int K, const float* __restrict__ bias, float beta, float scale) {
  __shared__ float red[kT / 64];
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* a = A + (size_t)row * K;
  float acc = 0.f;
  if ((K & 3) == 0 && bg_aligned16(a) && bg_aligned16(w)) {
    const float4* a4 = reinterpret_cast<const float4*>(a);
    const float4* w4 = reinterpret_cast<const float4*>(w);
    float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
    for (int k = tid; k < (K >> 2); k += kT) {
      const float4 x = a4[k], y = w4[k];
      p0 = fmaf(x.x, y.x, p0); p1 = fmaf(x.y, y.y, p1); p2 = fmaf(x.z, y.z, p2); p3 = fmaf(x.w, y.w, p3);
    }
    acc = (p0 + p1) + (p2 + p3);
  } else {
    for (int k = tid; k < K; k += kT) acc = fmaf(a[k], w[k], acc);
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) {
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < kT / 64; ++i) sum += red[i];
    float v = scale * sum;
    if (bias) v += bias[0];
    if (beta != 0.f) v += beta * C[row];
    C[row] = v;
  }
}

// ---------------------------------------------------------------- column reductions
// Block = 4 row-lanes x 64 columns; grid (row blocks, column groups); partial[blockIdx.x][q][n].
constexpr int kColBlocks = 256;

template <int NQ, class F>
__device__ inline void col_reduce(F f, int M, int N, float* partial) {
  __shared__ float red[NQ][4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int n = blockIdx.y * 64 + tx;
  float acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) acc[q] = 0.f;
  if (n < N) {
    const int rows_per = (M + gridDim.x - 1) / gridDim.x;
    const int r0 = blockIdx.x * rows_per, r1 = min(M, r0 + rows_per);
    for (int m = r0 + ty; m < r1; m += 4) f(m, n, acc);
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) red[q][ty][tx] = acc[q];
  __syncthreads();
  if (ty == 0 && n < N) {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      partial[((size_t)blockIdx.x * NQ + q) * N + n] = red[q][0][tx] + red[q][1][tx] + red[q][2][tx] + red[q][3][tx];
  }
}

// Fast path for C a power of two in [4, 1024]: the matrix is swept as a flat float4 stream, so every lane moves
// 16 B whatever C is; a thread's four channels are fixed ((4*tid) mod C) because 1024 mod C == 0.
// The sweep keeps U loads in flight per thread: `ld(q4)` fetches the operands of float4 number q4, `use(v, c4, acc)` adds them --
// loads of U iterations first, then their adds in the loop's own order (same sums, same bits).  With one load per iteration, each
// waited for before the next is issued, 512 workgroups of four waves hold 2 MB in flight and the pass ran at 3.6-4.3 TB/s.
template <int NQ, int U, class L, class F>
__device__ inline void flat_reduce(L ld, F use, int M, int C, float* partial) {
  __shared__ float4 red[NQ][kT];
  const int rows_per = (M + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * rows_per, r1 = min(M, r0 + rows_per);
  const int G = C >> 2;                           // distinct channel quads
  float4 acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (r1 > r0) {
    const size_t base4 = (size_t)r0 * G, total4 = (size_t)(r1 - r0) * G;
    const int c4 = (threadIdx.x % G) * 4;
    size_t q4 = threadIdx.x;
    for (; q4 + (size_t)(U - 1) * kT < total4; q4 += (size_t)U * kT) {
      decltype(ld((size_t)0)) v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = ld(base4 + q4 + (size_t)u * kT);
#pragma unroll
      for (int u = 0; u < U; ++u) use(v[u], c4, acc);
    }
    for (; q4 < total4; q4 += kT) use(ld(base4 + q4), c4, acc);
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) red[q][threadIdx.x] = acc[q];
  __syncthreads();
  if ((int)threadIdx.x < G) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      float4 s = red[q][threadIdx.x];
      for (int t = threadIdx.x + G; t < kT; t += G) {
        const float4 v = red[q][t];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      *reinterpret_cast<float4*>(partial + ((size_t)blockIdx.x * NQ + q) * C + threadIdx.x * 4) = s;
    }
  }
}

inline bool flat_ok(int M, int C) { return C >= 4 && C <= 1024 && (C & (C - 1)) == 0 && M >= 64; }
inline int flat_blocks(int M, int C) {
  const size_t total4 = (size_t)M * C / 4;
  static const size_t cap = getenv("BG_FLAT_BLOCKS") ? (size_t)atoi(getenv("BG_FLAT_BLOCKS")) : 512;
  return (int)std::max<size_t>(1, std::min<size_t>({cap, total4 / (kT * 8), (size_t)M}));
}

__global__ __launch_bounds__(kT) void colsum_flat_kernel(const float* __restrict__ x, int M, int C, int square, float* partial) {
  flat_reduce<1, 8>([&](size_t q4) { return reinterpret_cast<const float4*>(x)[q4]; }, [&](const float4& v, int, float4* a) {
    if (square) { a[0].x = fmaf(v.x, v.x, a[0].x); a[0].y = fmaf(v.y, v.y, a[0].y); a[0].z = fmaf(v.z, v.z, a[0].z); a[0].w = fmaf(v.w, v.w, a[0].w); }
    else { a[0].x += v.x; a[0].y += v.y; a[0].z += v.z; a[0].w += v.w; }
  }, M, C, partial);
}

__global__ __launch_bounds__(kT) void bn_stats_flat_kernel(const float* __restrict__ x, int M, int C, float* partial) {
  flat_reduce<2, 8>([&](size_t q4) { return reinterpret_cast<const float4*>(x)[q4]; }, [&](const float4& v, int, float4* a) {
    a[0].x += v.x; a[0].y += v.y; a[0].z += v.z; a[0].w += v.w;
    a[1].x = fmaf(v.x, v.x, a[1].x); a[1].y = fmaf(v.y, v.y, a[1].y); a[1].z = fmaf(v.z, v.z, a[1].z); a[1].w = fmaf(v.w, v.w, a[1].w);
  }, M, C, partial);
}

// y == nullptr (all three backward kernels): the sign of the activation is re-derived from x with the forward's own expression
// (bn_apply_kernel: v = gamma * ((x - mean) * inv) + beta, y = v > 0 ? v : alpha * v, so y > 0 <=> v > 0 for alpha >= 0 -- the same
// operations on the same inputs, hence the same bits) and the saved activation is not read: one tensor less per pass.
__device__ __forceinline__ float bn_pre_act(float x, float g, float m, float iv, float bt) { return g * ((x - m) * iv) + bt; }

__global__ __launch_bounds__(kT) void bn_bwd_flat_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                         const float* __restrict__ x, int M, int C, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ mean,
                                                         const float* __restrict__ inv, float alpha, float* partial) {
  struct Ops { float4 d, xx, yy; };
  flat_reduce<2, 4>([&](size_t q4) {
    Ops o;
    o.d = reinterpret_cast<const float4*>(dy)[q4];
    o.xx = reinterpret_cast<const float4*>(x)[q4];
    o.yy = y ? reinterpret_cast<const float4*>(y)[q4] : make_float4(0.f, 0.f, 0.f, 0.f);
    return o;
  }, [&](const Ops& o, int c4, float4* a) {
    const float4 d = o.d, xx = o.xx;
    const float4 mu = *reinterpret_cast<const float4*>(mean + c4), iv = *reinterpret_cast<const float4*>(inv + c4);
    float4 yy = o.yy;
    if (!y) {
      const float4 g = make_float4(gamma[c4], gamma[c4 + 1], gamma[c4 + 2], gamma[c4 + 3]);
      const float4 bt = make_float4(beta[c4], beta[c4 + 1], beta[c4 + 2], beta[c4 + 3]);
      yy.x = bn_pre_act(xx.x, g.x, mu.x, iv.x, bt.x); yy.y = bn_pre_act(xx.y, g.y, mu.y, iv.y, bt.y);
      yy.z = bn_pre_act(xx.z, g.z, mu.z, iv.z, bt.z); yy.w = bn_pre_act(xx.w, g.w, mu.w, iv.w, bt.w);
    }
    const float dz0 = d.x * (yy.x > 0.f ? 1.f : alpha), dz1 = d.y * (yy.y > 0.f ? 1.f : alpha);
    const float dz2 = d.z * (yy.z > 0.f ? 1.f : alpha), dz3 = d.w * (yy.w > 0.f ? 1.f : alpha);
    a[0].x += dz0; a[0].y += dz1; a[0].z += dz2; a[0].w += dz3;
    a[1].x = fmaf(dz0, (xx.x - mu.x) * iv.x, a[1].x); a[1].y = fmaf(dz1, (xx.y - mu.y) * iv.y, a[1].y);
    a[1].z = fmaf(dz2, (xx.z - mu.z) * iv.z, a[1].z); a[1].w = fmaf(dz3, (xx.w - mu.w) * iv.w, a[1].w);
  }, M, C, partial);
}

// one wave per channel sums the per-block partials: partial[(b*NQ + q)*C + c]
// (the loads of a lane are issued eight at a time and added in the loop's own order: the sum keeps its bits, and a lane waits for
// memory once per eight partial rows instead of once per row -- these finals are pure latency, 4-7 us for microseconds of work)
__device__ inline float wave_sum_partials(const float* __restrict__ partial, int nblk, int NQ, int q, int C, int c) {
  float s = 0.f;
  int b = threadIdx.x & 63;
  const size_t step = (size_t)64 * NQ * C;
  const float* p = partial + ((size_t)b * NQ + q) * C + c;
  for (; b + 7 * 64 < nblk; b += 8 * 64, p += 8 * step) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = p[i * step];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
  }
  if (b < nblk) {                               // the last, partial batch (all of it when nblk < 512): predicated, still one wait
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = b + i * 64 < nblk ? p[i * step] : 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (b + i * 64 < nblk) s += v[i];
  }
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  return __shfl(s, 0, 64);
}

__global__ __launch_bounds__(kT) void colsum_partial_kernel(const float* __restrict__ x, int M, int N, int square, float* partial) {
  col_reduce<1>([&](int m, int n, float* a) {
    const float v = x[(size_t)m * N + n];
    a[0] += square ? v * v : v;
  }, M, N, partial);
}

__global__ __launch_bounds__(kT) void colsum_final_kernel(const float* __restrict__ partial, int nblk, int N, float* out, float beta,
                                                          float scale) {
  const int n = blockIdx.x * (kT / 64) + (threadIdx.x >> 6);
  if (n >= N) return;
  const float s = wave_sum_partials(partial, nblk, 1, 0, N, n);
  if ((threadIdx.x & 63) == 0) out[n] = (beta != 0.f ? beta * out[n] : 0.f) + scale * s;
}

int col_blocks(int M) { return std::max(1, std::min(kColBlocks, M / 16)); }
int red_blocks(int M, int C) { return flat_ok(M, C) ? flat_blocks(M, C) : col_blocks(M); }

// ---------------------------------------------------------------- BatchNormalization + LeakyReLU
__global__ __launch_bounds__(kT) void bn_stats_partial_kernel(const float* __restrict__ x, int M, int C, float* partial) {
  col_reduce<2>([&](int m, int n, float* a) {
    const float v = x[(size_t)m * C + n];
    a[0] += v;
    a[1] = fmaf(v, v, a[1]);
  }, M, C, partial);
}

__global__ __launch_bounds__(kT) void bn_stats_final_kernel(const float* __restrict__ partial, int nblk, int M, int C, float* save_mean,
                                                            float* save_inv, float* moving_mean, float* moving_var, float eps,
                                                            float momentum, int unbiased) {
  const int c = blockIdx.x * (kT / 64) + (threadIdx.x >> 6);
  if (c >= C) return;
  const float s = wave_sum_partials(partial, nblk, 2, 0, C, c), s2 = wave_sum_partials(partial, nblk, 2, 1, C, c);
  if ((threadIdx.x & 63) != 0) return;
  const float mean = s / (float)M;
  const float var = fmaxf(s2 / (float)M - mean * mean, 0.f);
  save_mean[c] = mean;
  save_inv[c] = 1.0f / sqrtf(var + eps);
  if (moving_mean) {
    const float vu = unbiased ? var * ((float)M / (float)max(M - 1, 1)) : var;
    moving_mean[c] = moving_mean[c] * momentum + mean * (1.f - momentum);
    moving_var[c] = moving_var[c] * momentum + vu * (1.f - momentum);
  }
}

__global__ __launch_bounds__(kT) void bn_fold_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ mm, const float* __restrict__ mv, float eps, int C,
                                                     float* scale, float* shift) {
  const int c = blockIdx.x * kT + threadIdx.x;
  if (c >= C) return;
  const float s = gamma[c] * (1.0f / sqrtf(mv[c] + eps));
  scale[c] = s;
  shift[c] = beta[c] - mm[c] * s;
}

// bn_fold_kernel for up to kFoldMax layers in ONE launch (blockIdx.y = layer): the generator's inference forward inside the D-step
// folds every BatchNorm before its first conv instead of one tiny launch between each pair of convs
constexpr int kFoldMax = 8;
struct FoldMany {
  const float* gamma[kFoldMax]; const float* beta[kFoldMax]; const float* mm[kFoldMax]; const float* mv[kFoldMax];
  float* scale[kFoldMax]; float* shift[kFoldMax];
  float eps[kFoldMax]; int C[kFoldMax];
};
__global__ __launch_bounds__(kT) void bn_fold_many_kernel(const FoldMany f) {
  const int l = blockIdx.y, c = blockIdx.x * kT + threadIdx.x;
  if (c >= f.C[l]) return;
  const float s = f.gamma[l][c] * (1.0f / sqrtf(f.mv[l][c] + f.eps[l]));
  f.scale[l][c] = s;
  f.shift[l][c] = f.beta[l][c] - f.mm[l][c] * s;
}

// sums[q*C + c] = sum over blocks of partial[(b*2 + q)*C + c]   (one wave per channel)
__global__ __launch_bounds__(kT) void partials_to_sums_kernel(const float* __restrict__ partial, int nblk, int C, float* sums) {
  const int c = blockIdx.x * (kT / 64) + (threadIdx.x >> 6);
  if (c >= C) return;
  const float s0 = wave_sum_partials(partial, nblk, 2, 0, C, c), s1 = wave_sum_partials(partial, nblk, 2, 1, C, c);
  if ((threadIdx.x & 63) == 0) {
    sums[c] = s0;
    sums[C + c] = s1;
  }
}

__global__ __launch_bounds__(kT) void bn_finalize_sums_kernel(const float* __restrict__ sums, int M, int C, float* save_mean,
                                                              float* save_inv, float* moving_mean, float* moving_var, float eps,
                                                              float momentum, int unbiased) {
  const int c = blockIdx.x * kT + threadIdx.x;
  if (c >= C) return;
  const float mean = sums[c] / (float)M;
  const float var = fmaxf(sums[C + c] / (float)M - mean * mean, 0.f);
  save_mean[c] = mean;
  save_inv[c] = 1.0f / sqrtf(var + eps);
  if (moving_mean) {
    const float vu = unbiased ? var * ((float)M / (float)max(M - 1, 1)) : var;
    moving_mean[c] = moving_mean[c] * momentum + mean * (1.f - momentum);
    moving_var[c] = moving_var[c] * momentum + vu * (1.f - momentum);
  }
}

// four consecutive per-channel parameters: one float4 when the array is 16-byte aligned (a slice of a flat parameter buffer need
// not be), four floats otherwise
__device__ inline float4 load4_param(const float* __restrict__ p, int c, bool al16) {
  if (al16) return *reinterpret_cast<const float4*>(p + c);
  return make_float4(p[c], p[c + 1], p[c + 2], p[c + 3]);
}
__device__ inline bool ptr_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// y = lrelu(gamma*(x-mean)*inv + beta); mean/inv either saved batch stats or derived from moving stats
// bn_finalize_sums_kernel + bn_apply_kernel in one launch (SyncBN forward: conv -> sums -> all-reduce -> THIS): every thread derives
// mean / inv of its channels from the (all-reduced) sums with the expressions of bn_finalize_sums_kernel, so y is bit-identical to
// the two launches; the threads that hold the first row (e < C) also leave save_mean / save_inv and update the moving statistics.
__global__ __launch_bounds__(kT) void bn_finalize_apply_kernel(const float* __restrict__ sums, int M, const float* __restrict__ x,
                                                               float* __restrict__ y, size_t total, int C, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* save_mean, float* save_inv,
                                                               float* moving_mean, float* moving_var, float eps, float momentum,
                                                               int unbiased, float alpha) {
  for (size_t e = (size_t)blockIdx.x * kT + threadIdx.x; e < total; e += (size_t)gridDim.x * kT) {
    const int c = (int)(e % C);
    const float mean = sums[c] / (float)M;
    const float var = fmaxf(sums[C + c] / (float)M - mean * mean, 0.f);
    const float inv = 1.0f / sqrtf(var + eps);
    if (e < (size_t)C) {
      save_mean[c] = mean;
      save_inv[c] = inv;
      if (moving_mean) {
        const float vu = unbiased ? var * ((float)M / (float)max(M - 1, 1)) : var;
        moving_mean[c] = moving_mean[c] * momentum + mean * (1.f - momentum);
        moving_var[c] = moving_var[c] * momentum + vu * (1.f - momentum);
      }
    }
    const float v = gamma[c] * ((x[e] - mean) * inv) + beta[c];
    y[e] = v > 0.f ? v : alpha * v;
  }
}

__global__ __launch_bounds__(kT) void bn_apply_kernel(const float* __restrict__ x, float* __restrict__ y, size_t total, int C,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ mean, const float* __restrict__ inv_or_var,
                                                      int is_var, float eps, float alpha) {
  // four channels per thread where the layout allows (C % 4 == 0, 16-byte aligned activations, < 2^32 elements): float4 moves and one
  // 32-bit modulo per four elements instead of a 64-bit one per element; the arithmetic per element is the scalar path's
  if ((C & 3) == 0 && ptr_al16(x) && ptr_al16(y) && total <= 0xffffffffull) {
    const bool pa = ptr_al16(gamma) && ptr_al16(beta) && ptr_al16(mean) && ptr_al16(inv_or_var);
    const unsigned total4 = (unsigned)(total >> 2);
    const unsigned stride = gridDim.x * kT;
    auto one = [&](const float4& xv, const float4& g, const float4& bt, const float4& m, const float4& iv) {
      float4 v;
      v.x = g.x * ((xv.x - m.x) * iv.x) + bt.x; v.y = g.y * ((xv.y - m.y) * iv.y) + bt.y;
      v.z = g.z * ((xv.z - m.z) * iv.z) + bt.z; v.w = g.w * ((xv.w - m.w) * iv.w) + bt.w;
      v.x = v.x > 0.f ? v.x : alpha * v.x; v.y = v.y > 0.f ? v.y : alpha * v.y;
      v.z = v.z > 0.f ? v.z : alpha * v.z; v.w = v.w > 0.f ? v.w : alpha * v.w;
      return v;
    };
    auto params = [&](int c, float4& g, float4& bt, float4& m, float4& iv) {
      g = load4_param(gamma, c, pa); bt = load4_param(beta, c, pa); m = load4_param(mean, c, pa);
      iv = load4_param(inv_or_var, c, pa);
      if (is_var) { iv.x = 1.0f / sqrtf(iv.x + eps); iv.y = 1.0f / sqrtf(iv.y + eps); iv.z = 1.0f / sqrtf(iv.z + eps); iv.w = 1.0f / sqrtf(iv.w + eps); }
    };
    unsigned q = blockIdx.x * kT + threadIdx.x;
    if ((stride * 4u) % (unsigned)C == 0u) {
      // a thread meets the same four channels in every iteration: their parameters are read once, and two float4 of the stream
      // are in flight per thread (the per-iteration form re-read four parameter quads and waited for each before its one store)
      float4 g, bt, m, iv;
      params((int)((q * 4u) % (unsigned)C), g, bt, m, iv);
      for (; q + stride < total4; q += 2 * stride) {
        const float4 x0 = *reinterpret_cast<const float4*>(x + q * 4u), x1 = *reinterpret_cast<const float4*>(x + (q + stride) * 4u);
        *reinterpret_cast<float4*>(y + q * 4u) = one(x0, g, bt, m, iv);
        *reinterpret_cast<float4*>(y + (q + stride) * 4u) = one(x1, g, bt, m, iv);
      }
      if (q < total4) *reinterpret_cast<float4*>(y + q * 4u) = one(*reinterpret_cast<const float4*>(x + q * 4u), g, bt, m, iv);
      return;
    }
    for (; q < total4; q += stride) {
      const unsigned e = q * 4u;
      float4 g, bt, m, iv;
      params((int)(e % (unsigned)C), g, bt, m, iv);
      *reinterpret_cast<float4*>(y + e) = one(*reinterpret_cast<const float4*>(x + e), g, bt, m, iv);
    }
    return;
  }
  for (size_t e = (size_t)blockIdx.x * kT + threadIdx.x; e < total; e += (size_t)gridDim.x * kT) {
    const int c = (int)(e % C);
    const float inv = is_var ? 1.0f / sqrtf(inv_or_var[c] + eps) : inv_or_var[c];
    float v = gamma[c] * ((x[e] - mean[c]) * inv) + beta[c];
    y[e] = v > 0.f ? v : alpha * v;
  }
}

__global__ __launch_bounds__(kT) void bn_bwd_partial_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                            const float* __restrict__ x, int M, int C, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            const float* __restrict__ mean, const float* __restrict__ inv, float alpha,
                                                            float* partial) {
  col_reduce<2>([&](int m, int n, float* a) {
    const size_t e = (size_t)m * C + n;
    const float yv = y ? y[e] : bn_pre_act(x[e], gamma[n], mean[n], inv[n], beta[n]);
    const float dz = dy[e] * (yv > 0.f ? 1.f : alpha);
    a[0] += dz;
    a[1] = fmaf(dz, (x[e] - mean[n]) * inv[n], a[1]);
  }, M, C, partial);
}

__global__ __launch_bounds__(kT) void bn_bwd_final_kernel(const float* __restrict__ partial, int nblk, int C, float* dgamma, float* dbeta) {
  const int c = blockIdx.x * (kT / 64) + (threadIdx.x >> 6);
  if (c >= C) return;
  const float s = wave_sum_partials(partial, nblk, 2, 0, C, c), s2 = wave_sum_partials(partial, nblk, 2, 1, C, c);
  if ((threadIdx.x & 63) != 0) return;
  dbeta[c] = s;
  dgamma[c] = s2;
}

__global__ __launch_bounds__(kT) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                          const float* __restrict__ x, float* __restrict__ dx, size_t total, int M /* rows behind the statistics */, int C,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ inv, const float* __restrict__ dgamma,
                                                          const float* __restrict__ dbeta, float alpha) {
  const float invM = 1.0f / (float)M;
  if ((C & 3) == 0 && ptr_al16(dy) && ptr_al16(y) && ptr_al16(x) && ptr_al16(dx) && total <= 0xffffffffull) {     // see bn_apply_kernel
    const bool pa = ptr_al16(gamma) && ptr_al16(mean) && ptr_al16(inv) && ptr_al16(dgamma) && ptr_al16(dbeta) && ptr_al16(beta);
    const unsigned total4 = (unsigned)(total >> 2);
    const float fM = (float)M;
    const unsigned stride = gridDim.x * kT;
    struct P { float4 g, m, iv, dg, db, bt; };
    auto params = [&](int c) {
      P p;
      p.g = load4_param(gamma, c, pa); p.m = load4_param(mean, c, pa); p.iv = load4_param(inv, c, pa);
      p.dg = load4_param(dgamma, c, pa); p.db = load4_param(dbeta, c, pa);
      p.bt = y ? make_float4(0.f, 0.f, 0.f, 0.f) : load4_param(beta, c, pa);
      return p;
    };
    auto one = [&](const float4& dyv, const float4& xv, float4 yv, const P& p) {
      if (!y) {
        yv.x = bn_pre_act(xv.x, p.g.x, p.m.x, p.iv.x, p.bt.x); yv.y = bn_pre_act(xv.y, p.g.y, p.m.y, p.iv.y, p.bt.y);
        yv.z = bn_pre_act(xv.z, p.g.z, p.m.z, p.iv.z, p.bt.z); yv.w = bn_pre_act(xv.w, p.g.w, p.m.w, p.iv.w, p.bt.w);
      }
      float4 o;
#define BG_BN_BWD1(k)                                                   \
  {                                                                     \
    const float dz = dyv.k * (yv.k > 0.f ? 1.f : alpha);                \
    const float xh = (xv.k - p.m.k) * p.iv.k;                           \
    o.k = p.g.k * p.iv.k * invM * (fM * dz - p.db.k - xh * p.dg.k);     \
  }
      BG_BN_BWD1(x) BG_BN_BWD1(y) BG_BN_BWD1(z) BG_BN_BWD1(w)
#undef BG_BN_BWD1
      return o;
    };
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned q = blockIdx.x * kT + threadIdx.x;
    if ((stride * 4u) % (unsigned)C == 0u) {       // fixed channels per thread: parameters once, two stream quads in flight (see bn_apply_kernel)
      const P p = params((int)((q * 4u) % (unsigned)C));
      for (; q + stride < total4; q += 2 * stride) {
        const unsigned e0 = q * 4u, e1 = (q + stride) * 4u;
        const float4 d0 = *reinterpret_cast<const float4*>(dy + e0), d1 = *reinterpret_cast<const float4*>(dy + e1);
        const float4 x0 = *reinterpret_cast<const float4*>(x + e0), x1 = *reinterpret_cast<const float4*>(x + e1);
        const float4 y0 = y ? *reinterpret_cast<const float4*>(y + e0) : z4, y1 = y ? *reinterpret_cast<const float4*>(y + e1) : z4;
        *reinterpret_cast<float4*>(dx + e0) = one(d0, x0, y0, p);
        *reinterpret_cast<float4*>(dx + e1) = one(d1, x1, y1, p);
      }
      if (q < total4) {
        const unsigned e = q * 4u;
        *reinterpret_cast<float4*>(dx + e) = one(*reinterpret_cast<const float4*>(dy + e), *reinterpret_cast<const float4*>(x + e),
                                                 y ? *reinterpret_cast<const float4*>(y + e) : z4, p);
      }
      return;
    }
    for (; q < total4; q += stride) {
      const unsigned e = q * 4u;
      const P p = params((int)(e % (unsigned)C));
      *reinterpret_cast<float4*>(dx + e) = one(*reinterpret_cast<const float4*>(dy + e), *reinterpret_cast<const float4*>(x + e),
                                               y ? *reinterpret_cast<const float4*>(y + e) : z4, p);
    }
    return;
  }
  for (size_t e = (size_t)blockIdx.x * kT + threadIdx.x; e < total; e += (size_t)gridDim.x * kT) {
    const int c = (int)(e % C);
    const float yv = y ? y[e] : bn_pre_act(x[e], gamma[c], mean[c], inv[c], beta[c]);
    const float dz = dy[e] * (yv > 0.f ? 1.f : alpha);
    const float xh = (x[e] - mean[c]) * inv[c];
    dx[e] = gamma[c] * inv[c] * invM * ((float)M * dz - dbeta[c] - xh * dgamma[c]);
  }
}

// dgamma = scale * sums[C + c], dbeta = scale * sums[c]: the parameter gradients of a SyncBN layer out of the (already global)
// backward sums, pre-divided by the replica count so that the SUM all-reduce of the flat gradient buffer restores them
__global__ __launch_bounds__(kT) void bn_param_grads_kernel(const float* __restrict__ sums, int C, float scale, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta) {
  const int c = blockIdx.x * kT + threadIdx.x;
  if (c >= C) return;
  dbeta[c] = scale * sums[c];
  dgamma[c] = scale * sums[C + c];
}

// ---------------------------------------------------------------- pointwise
__global__ __launch_bounds__(kT) void lerp_kernel(const float* __restrict__ r, const float* __restrict__ f, const float* __restrict__ alpha,
                                                  float* __restrict__ out, size_t total, int n_per) {
  for (size_t e = (size_t)blockIdx.x * kT + threadIdx.x; e < total; e += (size_t)gridDim.x * kT) {
    const float a = alpha[e / n_per];
    out[e] = r[e] + a * (f[e] - r[e]);
  }
}

// one workgroup of 1024 threads per sample (B is 128-256: few workgroups, so each must keep many loads in flight); fixed
// summation order: per-thread strided partials, wave shuffles, then the 16 wave sums in order
__global__ __launch_bounds__(1024) void row_norm_kernel(const float* __restrict__ g, float* __restrict__ norm, int n_per) {
  __shared__ float red[16];
  const float* row = g + (size_t)blockIdx.x * n_per;
  float acc = 0.f;
  if ((n_per & 3) == 0 && ((reinterpret_cast<uintptr_t>(row) & 15u) == 0)) {
    const float4* r4 = reinterpret_cast<const float4*>(row);
    for (int i = threadIdx.x; i < n_per / 4; i += 1024) {
      const float4 v = r4[i];
      acc = fmaf(v.x, v.x, acc); acc = fmaf(v.y, v.y, acc); acc = fmaf(v.z, v.z, acc); acc = fmaf(v.w, v.w, acc);
    }
  } else {
    for (int i = threadIdx.x; i < n_per; i += 1024) acc = fmaf(row[i], row[i], acc);
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < 16; ++w) s += red[w];
    norm[blockIdx.x] = sqrtf(s);
  }
}

// GUARD = false reproduces the reference: a sample whose input gradient is exactly zero gives (0 - 1) / 0 * 0 = NaN, as
// tf.norm's gradient does (wgan.py:245).  GUARD = true (build-side switch) takes the subgradient 0 there.
template <bool GUARD>
__global__ __launch_bounds__(kT) void gp_seed_kernel(const float* __restrict__ g, const float* __restrict__ norm, float coef,
                                                     float* __restrict__ out, size_t total, int n_per) {
  for (size_t e = (size_t)blockIdx.x * kT + threadIdx.x; e < total; e += (size_t)gridDim.x * kT) {
    const float n = norm[e / n_per];
    out[e] = (GUARD && n == 0.f) ? 0.f : coef * ((n - 1.f) / n) * g[e];
  }
}

__global__ __launch_bounds__(kT) void mul_grad_kernel(const float* __restrict__ d, const float* __restrict__ ref,
                                                      const uint8_t* __restrict__ keep, float alpha, float scale,
                                                      float* __restrict__ out, size_t n) {
  for (size_t e = (size_t)blockIdx.x * kT + threadIdx.x; e < n; e += (size_t)gridDim.x * kT) {
    float f = ref[e] > 0.f ? 1.f : alpha;
    if (keep) f = keep[e] ? f * scale : 0.f;
    out[e] = d[e] * f;
  }
}

__global__ __launch_bounds__(kT) void tanh_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ out,
                                                      size_t n) {
  for (size_t e = (size_t)blockIdx.x * kT + threadIdx.x; e < n; e += (size_t)gridDim.x * kT) out[e] = dy[e] * (1.f - y[e] * y[e]);
}

__global__ __launch_bounds__(kT) void outer_kernel(const float* __restrict__ s, const float* __restrict__ w, float* __restrict__ out,
                                                   size_t total, int K) {
  for (size_t e = (size_t)blockIdx.x * kT + threadIdx.x; e < total; e += (size_t)gridDim.x * kT) out[e] = s[e / K] * w[e % K];
}

__global__ __launch_bounds__(kT) void fill_kernel(float* x, float v, size_t n) {
  for (size_t e = (size_t)blockIdx.x * kT + threadIdx.x; e < n; e += (size_t)gridDim.x * kT) x[e] = v;
}

// dst = src; float4 body when both are 16-byte aligned, scalar tail / fallback otherwise
__global__ __launch_bounds__(kT) void copy_kernel(float* __restrict__ dst, const float* __restrict__ src, size_t n, int vec) {
  const size_t tid = (size_t)blockIdx.x * kT + threadIdx.x, nth = (size_t)gridDim.x * kT;
  size_t done = 0;
  if (vec) {
    const size_t n4 = n / 4;
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (size_t e = tid; e < n4; e += nth) d4[e] = s4[e];
    done = n4 * 4;
  }
  for (size_t e = done + tid; e < n; e += nth) dst[e] = src[e];
}

__global__ __launch_bounds__(kT) void scale_kernel(float* x, float v, size_t n) {
  for (size_t e = (size_t)blockIdx.x * kT + threadIdx.x; e < n; e += (size_t)gridDim.x * kT) x[e] *= v;
}

// ---------------------------------------------------------------- losses (single workgroup; B is small)
__device__ inline float block_sum(float v, float* red) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__device__ inline float sgnf(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

__global__ __launch_bounds__(kT) void d_loss_kernel(const float* __restrict__ fs, const float* __restrict__ rs,
                                                    const float* __restrict__ norm, int B, float inv_gbs, float gp_coef, float e_drift,
                                                    float vec_scale, float* dfs, float* drs, float* metrics) {
  __shared__ float red[4];
  float sf = 0.f, sr = 0.f, sn = 0.f, sg = 0.f;
  for (int b = threadIdx.x; b < B; b += kT) {
    const float f = fs[b], r = rs[b];
    sf += f;
    sr += r;
    sn += e_drift * (fabsf(f) + fabsf(r));
    if (norm) {
      const float d = norm[b] - 1.f;
      sg = fmaf(d, d, sg);
    }
    dfs[b] = vec_scale * inv_gbs + e_drift * sgnf(f);
    drs[b] = -vec_scale * inv_gbs + e_drift * sgnf(r);
  }
  sf = block_sum(sf, red);
  sr = block_sum(sr, red);
  sn = block_sum(sn, red);
  sg = block_sum(sg, red);
  if (threadIdx.x == 0) {
    const float gp = sg / (float)B;
    const float lw = (sf - sr) * inv_gbs;
    metrics[0] = sf / (float)B;
    metrics[1] = sr / (float)B;
    metrics[2] = lw + gp_coef * gp + sn / (float)B;   // mean of the [B] loss vector (wgan.py:145)
    metrics[3] = gp_coef * gp;
    metrics[4] = sn / (float)B;
    metrics[5] = gp;
  }
}

__global__ __launch_bounds__(kT) void g_loss_kernel(const float* __restrict__ s, int B, float inv_gbs, float* ds, float* metrics) {
  __shared__ float red[4];
  float ss = 0.f;
  for (int b = threadIdx.x; b < B; b += kT) {
    ss += s[b];
    ds[b] = -inv_gbs;
  }
  ss = block_sum(ss, red);
  if (threadIdx.x == 0) {
    metrics[0] = ss / (float)B;
    metrics[1] = -ss * inv_gbs;
  }
}

// ---------------------------------------------------------------- input pipeline: u8 -> normalised f32 (+ bilinear resize)
__global__ __launch_bounds__(kT) void u8_normalize_resize_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, size_t total,
                                                                 int Hs, int Ws, int C, int Hd, int Wd, float sy, float sx) {
  for (size_t e = (size_t)blockIdx.x * kT + threadIdx.x; e < total; e += (size_t)gridDim.x * kT) {
    const int c = (int)(e % C);
    size_t t = e / C;
    const int x = (int)(t % Wd);
    t /= Wd;
    const int y = (int)(t % Hd);
    const int b = (int)(t / Hd);
    // [TF] resize_bilinear, half_pixel_centers: in = (out + 0.5) * scale - 0.5; lower = max(floor(in), 0),
    // upper = min(ceil(in), size - 1), lerp = in - floor(in)
    const float fy = ((float)y + 0.5f) * sy - 0.5f, fx = ((float)x + 0.5f) * sx - 0.5f;
    const float fy0 = floorf(fy), fx0 = floorf(fx);
    const int y0 = max((int)fy0, 0), y1 = min((int)ceilf(fy), Hs - 1);
    const int x0 = max((int)fx0, 0), x1 = min((int)ceilf(fx), Ws - 1);
    const float ly = fy - fy0, lx = fx - fx0;
    const uint8_t* img = src + (size_t)b * Hs * Ws * C + c;
    auto px = [&](int yy, int xx) { return ((float)img[((size_t)yy * Ws + xx) * C] - 127.5f) / 127.5f; };
    const float top = px(y0, x0) + (px(y0, x1) - px(y0, x0)) * lx;
    const float bot = px(y1, x0) + (px(y1, x1) - px(y1, x0)) * lx;
    dst[e] = top + (bot - top) * ly;
  }
}

// ---------------------------------------------------------------- Adam
__global__ __launch_bounds__(kT) void adam_kernel(float* __restrict__ theta, float* __restrict__ m, float* __restrict__ v,
                                                  const float* __restrict__ g, size_t n, float lr_t, float b1, float b2, float eps) {
  for (size_t e = (size_t)blockIdx.x * kT + threadIdx.x; e < n; e += (size_t)gridDim.x * kT) {
    const float gi = g[e];
    const float mi = b1 * m[e] + (1.f - b1) * gi;
    const float vi = b2 * v[e] + (1.f - b2) * gi * gi;
    m[e] = mi;
    v[e] = vi;
    theta[e] = theta[e] - lr_t * mi / (sqrtf(vi) + eps);
  }
}

// ---------------------------------------------------------------- RNG: Philox4x32-10, counter = element index / 4
__device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ inline float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

__global__ __launch_bounds__(kT) void uniform_kernel(float* out, size_t n, uint64_t seed, uint64_t offset) {
  const size_t nq = (n + 3) / 4;
  for (size_t q = (size_t)blockIdx.x * kT + threadIdx.x; q < nq; q += (size_t)gridDim.x * kT) {
    const uint64_t ctr = offset + q;
    uint32_t r[4];
    philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (q * 4 + i < n) out[q * 4 + i] = u01(r[i]);
  }
}

// element e takes word e % 4 of the Philox block with counter offset + e / 4 (the layout the oracle restates); a thread forms 16
// consecutive elements (four blocks) and leaves them with ONE 16-byte store -- byte stores held this kernel at 1.3 TB/s
__global__ __launch_bounds__(kT) void keep_mask_kernel(uint8_t* out, size_t n, float keep_prob, uint64_t seed, uint64_t offset) {
  const size_t n16 = (n + 15) / 16;
  const bool al16 = (reinterpret_cast<uintptr_t>(out) & 15) == 0;
  for (size_t t = (size_t)blockIdx.x * kT + threadIdx.x; t < n16; t += (size_t)gridDim.x * kT) {
    uint32_t w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint64_t ctr = offset + t * 4 + j;
      uint32_t r[4];
      philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 1u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
      w[j] = (u01(r[0]) < keep_prob ? 1u : 0u) | (u01(r[1]) < keep_prob ? 0x100u : 0u) | (u01(r[2]) < keep_prob ? 0x10000u : 0u) |
             (u01(r[3]) < keep_prob ? 0x1000000u : 0u);
    }
    if (al16 && t * 16 + 15 < n) {
      *reinterpret_cast<uint4*>(out + t * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
      for (int i = 0; i < 16; ++i)
        if (t * 16 + i < n) out[t * 16 + i] = (uint8_t)((w[i >> 2] >> (8 * (i & 3))) & 0xffu);
    }
  }
}

}  // namespace

#define BG_POINTWISE_PROLOGUE(fn, cond, n_)                                     \
  BG_REQUIRE(cond, BG_ERR_NULL, fn ": null pointer");                           \
  BG_REQUIRE((n_) > 0, BG_ERR_BAD_SHAPE, fn ": empty tensor")

extern "C" {

int bg_gemm_f32(const float* A, const float* Bm, float* C, int M, int N, int K, int transA, int transB, const float* bias,
                float beta, float scale, void* stream) {
  BG_REQUIRE(A && Bm && C, BG_ERR_NULL, "bg_gemm_f32: null pointer");
  BG_REQUIRE(M > 0 && N > 0 && K > 0, BG_ERR_BAD_SHAPE, "bg_gemm_f32: M=%d N=%d K=%d", M, N, K);
  const double flops = 2.0 * M * (double)N * K;
  if (N == 1 && !transA && K >= 64) {
    bg::Launch L(stream, "dense_rowdot", flops, 4.0 * M * K);
    bg::launch(rowdot_kernel, dim3(M), dim3(kT), 0, L.s, A, Bm, C, M, K, bias, beta, scale);
    return L.done("rowdot_kernel");
  }
  if (N == 1 && transA && K >= 64) {
    bg::Launch L(stream, "dense_gemv_t", flops, 4.0 * M * K);
    bg::launch(gemv_t_kernel, dim3(bg::cdiv(M, 64)), dim3(1024), 0, L.s, A, Bm, C, M, K, bias, beta, scale);
    return L.done("gemv_t_kernel");
  }
  if ((size_t)M * N >= 4096 && K >= 8) {
    bg::Launch L(stream, "dense_gemm_tiled", flops, 0);
    bg::launch(gemm_tiled_kernel, dim3(bg::cdiv(N, 64), bg::cdiv(M, 64)), dim3(kT), 0, L.s, A, Bm, C, M, N, K, transA, transB, bias, beta, scale);
    return L.done("gemm_tiled_kernel");
  }
  bg::Launch L(stream, "dense_gemm", flops, 0);
  bg::launch(gemm_naive_kernel, dim3(grid_for((size_t)M * N)), dim3(kT), 0, L.s, A, Bm, C, M, N, K, transA, transB, bias, beta, scale);
  return L.done("gemm_naive_kernel");
}

size_t bg_colsum_workspace_bytes(int M, int N) {
  if (M <= 0 || N <= 0) return 0;
  return (size_t)std::max(col_blocks(M), 512) * 2 * N * sizeof(float);
}

int bg_colsum_f32(const float* x, float* out, int M, int N, int square, float beta, float scale, void* ws_d, size_t ws_bytes,
                  void* stream) {
  BG_REQUIRE(x && out, BG_ERR_NULL, "bg_colsum_f32: null pointer");
  BG_REQUIRE(M > 0 && N > 0, BG_ERR_BAD_SHAPE, "bg_colsum_f32: M=%d N=%d", M, N);
  BG_REQUIRE(ws_d && ws_bytes >= bg_colsum_workspace_bytes(M, N), BG_ERR_WORKSPACE, "bg_colsum_f32: workspace too small");
  const int nblk = red_blocks(M, N);
  float* partial = static_cast<float*>(ws_d);
  {
    bg::Launch L(stream, "colsum_partial", 0, 4.0 * M * N);
    if (flat_ok(M, N) && bg::aligned16(x)) bg::launch(colsum_flat_kernel, dim3(nblk), dim3(kT), 0, L.s, x, M, N, square, partial);
    else bg::launch(colsum_partial_kernel, dim3(nblk, bg::cdiv(N, 64)), dim3(kT), 0, L.s, x, M, N, square, partial);
    int rc = L.done("colsum_partial_kernel");
    if (rc) return rc;
  }
  bg::Launch L(stream, "colsum_final", 0, 0);
  bg::launch(colsum_final_kernel, dim3(bg::cdiv(N, kT / 64)), dim3(kT), 0, L.s, partial, nblk, N, out, beta, scale);
  return L.done("colsum_final_kernel");
}

size_t bg_bn_workspace_bytes(int M, int C) { return bg_colsum_workspace_bytes(M, C); }

int bg_bn_train_fwd(const float* x, float* y, int M, int C, const float* gamma, const float* beta, float* moving_mean,
                    float* moving_var, float* save_mean, float* save_inv, float eps, float momentum, int unbiased,
                    float lrelu_alpha, void* ws_d, size_t ws_bytes, void* stream) {
  BG_REQUIRE(x && y && gamma && beta && save_mean && save_inv, BG_ERR_NULL, "bg_bn_train_fwd: null pointer");
  BG_REQUIRE((moving_mean == nullptr) == (moving_var == nullptr), BG_ERR_NULL, "bg_bn_train_fwd: moving_mean/var must both be given or both NULL");
  BG_REQUIRE(M > 0 && C > 0, BG_ERR_BAD_SHAPE, "bg_bn_train_fwd: M=%d C=%d", M, C);
  BG_REQUIRE(ws_d && ws_bytes >= bg_bn_workspace_bytes(M, C), BG_ERR_WORKSPACE, "bg_bn_train_fwd: workspace too small");
  const int nblk = red_blocks(M, C);
  float* partial = static_cast<float*>(ws_d);
  const size_t total = (size_t)M * C;
  {
    bg::Launch L(stream, "bn_stats_partial", 0, 4.0 * total);
    if (flat_ok(M, C) && bg::aligned16(x)) bg::launch(bn_stats_flat_kernel, dim3(nblk), dim3(kT), 0, L.s, x, M, C, partial);
    else bg::launch(bn_stats_partial_kernel, dim3(nblk, bg::cdiv(C, 64)), dim3(kT), 0, L.s, x, M, C, partial);
    int rc = L.done("bn_stats_partial_kernel");
    if (rc) return rc;
  }
  {
    bg::Launch L(stream, "bn_stats_final", 0, 0);
    bg::launch(bn_stats_final_kernel, dim3(bg::cdiv(C, kT / 64)), dim3(kT), 0, L.s, partial, nblk, M, C, save_mean, save_inv,
                       moving_mean, moving_var, eps, momentum, unbiased);
    int rc = L.done("bn_stats_final_kernel");
    if (rc) return rc;
  }
  bg::Launch L(stream, "bn_apply_lrelu", 0, 8.0 * total);
  bg::launch(bn_apply_kernel, dim3(grid_for(total)), dim3(kT), 0, L.s, x, y, total, C, gamma, beta, save_mean, save_inv, 0, eps,
                     lrelu_alpha);
  return L.done("bn_apply_kernel");
}

int bg_bn_train_fwd_partials(const float* partial_d, int nrows, const float* x, float* y, int M, int C, const float* gamma,
                             const float* beta, float* moving_mean, float* moving_var, float* save_mean, float* save_inv, float eps,
                             float momentum, int unbiased, float lrelu_alpha, void* stream) {
  BG_REQUIRE(partial_d && x && y && gamma && beta && save_mean && save_inv, BG_ERR_NULL, "bg_bn_train_fwd_partials: null pointer");
  BG_REQUIRE((moving_mean == nullptr) == (moving_var == nullptr), BG_ERR_NULL, "bg_bn_train_fwd_partials: moving_mean/var must both be given or both NULL");
  BG_REQUIRE(M > 0 && C > 0 && nrows > 0, BG_ERR_BAD_SHAPE, "bg_bn_train_fwd_partials: M=%d C=%d nrows=%d", M, C, nrows);
  const size_t total = (size_t)M * C;
  {
    bg::Launch L(stream, "bn_stats_final", 0, 0);
    bg::launch(bn_stats_final_kernel, dim3(bg::cdiv(C, kT / 64)), dim3(kT), 0, L.s, partial_d, nrows, M, C, save_mean, save_inv,
                       moving_mean, moving_var, eps, momentum, unbiased);
    int rc = L.done("bn_stats_final_kernel");
    if (rc) return rc;
  }
  bg::Launch L(stream, "bn_apply_lrelu", 0, 8.0 * total);
  bg::launch(bn_apply_kernel, dim3(grid_for(total)), dim3(kT), 0, L.s, x, y, total, C, gamma, beta, save_mean, save_inv, 0, eps,
                     lrelu_alpha);
  return L.done("bn_apply_kernel");
}

int bg_bn_sums_from_partials(const float* partial_d, int nrows, int C, float* sums_d, void* stream) {
  BG_REQUIRE(partial_d && sums_d, BG_ERR_NULL, "bg_bn_sums_from_partials: null pointer");
  BG_REQUIRE(C > 0 && nrows > 0, BG_ERR_BAD_SHAPE, "bg_bn_sums_from_partials: C=%d nrows=%d", C, nrows);
  bg::Launch L(stream, "bn_stats_final", 0, 0);
  bg::launch(partials_to_sums_kernel, dim3(bg::cdiv(C, kT / 64)), dim3(kT), 0, L.s, partial_d, nrows, C, sums_d);
  return L.done("partials_to_sums_kernel");
}

int bg_bn_infer_fwd(const float* x, float* y, int M, int C, const float* gamma, const float* beta, const float* moving_mean,
                    const float* moving_var, float eps, float lrelu_alpha, void* stream) {
  BG_REQUIRE(x && y && gamma && beta && moving_mean && moving_var, BG_ERR_NULL, "bg_bn_infer_fwd: null pointer");
  BG_REQUIRE(M > 0 && C > 0, BG_ERR_BAD_SHAPE, "bg_bn_infer_fwd: M=%d C=%d", M, C);
  const size_t total = (size_t)M * C;
  bg::Launch L(stream, "bn_apply_lrelu", 0, 8.0 * total);
  bg::launch(bn_apply_kernel, dim3(grid_for(total)), dim3(kT), 0, L.s, x, y, total, C, gamma, beta, moving_mean, moving_var, 1,
                     eps, lrelu_alpha);
  return L.done("bn_apply_kernel");
}

int bg_bn_train_bwd(const float* dy, const float* y, const float* x, float* dx, int M, int C, const float* gamma, const float* beta,
                    const float* save_mean, const float* save_inv, float* dgamma, float* dbeta, float lrelu_alpha, void* ws_d,
                    size_t ws_bytes, void* stream) {
  BG_REQUIRE(dy && (y || beta) && x && dx && gamma && save_mean && save_inv && dgamma && dbeta, BG_ERR_NULL, "bg_bn_train_bwd: null pointer");
  BG_REQUIRE(y || lrelu_alpha >= 0.f, BG_ERR_UNSUPPORTED, "bg_bn_train_bwd: the sign of y follows from x only for lrelu_alpha >= 0");
  BG_REQUIRE(M > 0 && C > 0, BG_ERR_BAD_SHAPE, "bg_bn_train_bwd: M=%d C=%d", M, C);
  BG_REQUIRE(ws_d && ws_bytes >= bg_bn_workspace_bytes(M, C), BG_ERR_WORKSPACE, "bg_bn_train_bwd: workspace too small");
  const int nblk = red_blocks(M, C);
  float* partial = static_cast<float*>(ws_d);
  const size_t total = (size_t)M * C;
  {
    bg::Launch L(stream, "bn_bwd_partial", 0, (y ? 12.0 : 8.0) * total);
    if (flat_ok(M, C) && bg::aligned16(dy) && bg::aligned16(y) && bg::aligned16(x) && bg::aligned16(save_mean) && bg::aligned16(save_inv))
      bg::launch(bn_bwd_flat_kernel, dim3(nblk), dim3(kT), 0, L.s, dy, y, x, M, C, gamma, beta, save_mean, save_inv, lrelu_alpha, partial);
    else
      bg::launch(bn_bwd_partial_kernel, dim3(nblk, bg::cdiv(C, 64)), dim3(kT), 0, L.s, dy, y, x, M, C, gamma, beta, save_mean, save_inv,
                         lrelu_alpha, partial);
    int rc = L.done("bn_bwd_partial_kernel");
    if (rc) return rc;
  }
  {
    bg::Launch L(stream, "bn_bwd_final", 0, 0);
    bg::launch(bn_bwd_final_kernel, dim3(bg::cdiv(C, kT / 64)), dim3(kT), 0, L.s, partial, nblk, C, dgamma, dbeta);
    int rc = L.done("bn_bwd_final_kernel");
    if (rc) return rc;
  }
  bg::Launch L(stream, "bn_bwd_apply", 0, (y ? 16.0 : 12.0) * total);
  bg::launch(bn_bwd_apply_kernel, dim3(grid_for(total)), dim3(kT), 0, L.s, dy, y, x, dx, total, M, C, gamma, beta, save_mean, save_inv,
                     dgamma, dbeta, lrelu_alpha);
  return L.done("bn_bwd_apply_kernel");
}

int bg_bn_fold_f32(const float* gamma, const float* beta, const float* moving_mean, const float* moving_var, float eps, int C,
                   float* scale_out, float* shift_out, void* stream) {
  BG_REQUIRE(gamma && beta && moving_mean && moving_var && scale_out && shift_out, BG_ERR_NULL, "bg_bn_fold_f32: null pointer");
  BG_REQUIRE(C > 0, BG_ERR_BAD_SHAPE, "bg_bn_fold_f32: C=%d", C);
  bg::Launch L(stream, "bn_fold", 0, 0);
  bg::launch(bn_fold_kernel, dim3(bg::cdiv(C, kT)), dim3(kT), 0, L.s, gamma, beta, moving_mean, moving_var, eps, C, scale_out, shift_out);
  return L.done("bn_fold_kernel");
}

int bg_bn_fold_many_f32(int n, const float* const* gamma, const float* const* beta, const float* const* moving_mean,
                        const float* const* moving_var, const float* eps, const int* C, float* const* scale_out, float* const* shift_out,
                        void* stream) {
  BG_REQUIRE(gamma && beta && moving_mean && moving_var && eps && C && scale_out && shift_out, BG_ERR_NULL, "bg_bn_fold_many_f32: null pointer");
  BG_REQUIRE(n > 0 && n <= kFoldMax, BG_ERR_BAD_SHAPE, "bg_bn_fold_many_f32: n=%d (1..%d)", n, kFoldMax);
  FoldMany f{};
  int cmax = 0;
  for (int l = 0; l < n; ++l) {
    BG_REQUIRE(gamma[l] && beta[l] && moving_mean[l] && moving_var[l] && scale_out[l] && shift_out[l], BG_ERR_NULL,
               "bg_bn_fold_many_f32: null pointer in layer %d", l);
    BG_REQUIRE(C[l] > 0, BG_ERR_BAD_SHAPE, "bg_bn_fold_many_f32: C[%d]=%d", l, C[l]);
    f.gamma[l] = gamma[l]; f.beta[l] = beta[l]; f.mm[l] = moving_mean[l]; f.mv[l] = moving_var[l];
    f.scale[l] = scale_out[l]; f.shift[l] = shift_out[l]; f.eps[l] = eps[l]; f.C[l] = C[l];
    cmax = std::max(cmax, C[l]);
  }
  bg::Launch L(stream, "bn_fold", 0, 0);
  bg::launch(bn_fold_many_kernel, dim3(bg::cdiv(cmax, kT), n), dim3(kT), 0, L.s, f);
  return L.done("bn_fold_many_kernel");
}

static int bn_partials(const char* fn, const float* x, int M, int C, float* partial, void* stream) {
  const int nblk = red_blocks(M, C);
  bg::Launch L(stream, "bn_stats_partial", 0, 4.0 * M * C);
  if (flat_ok(M, C) && bg::aligned16(x)) bg::launch(bn_stats_flat_kernel, dim3(nblk), dim3(kT), 0, L.s, x, M, C, partial);
  else bg::launch(bn_stats_partial_kernel, dim3(nblk, bg::cdiv(C, 64)), dim3(kT), 0, L.s, x, M, C, partial);
  return L.done(fn);
}

int bg_bn_stats_f32(const float* x, int M, int C, float* sums_d, void* ws_d, size_t ws_bytes, void* stream) {
  BG_REQUIRE(x && sums_d, BG_ERR_NULL, "bg_bn_stats_f32: null pointer");
  BG_REQUIRE(M > 0 && C > 0, BG_ERR_BAD_SHAPE, "bg_bn_stats_f32: M=%d C=%d", M, C);
  BG_REQUIRE(ws_d && ws_bytes >= bg_bn_workspace_bytes(M, C), BG_ERR_WORKSPACE, "bg_bn_stats_f32: workspace too small");
  float* partial = static_cast<float*>(ws_d);
  int rc = bn_partials("bn_stats_partial_kernel", x, M, C, partial, stream);
  if (rc) return rc;
  bg::Launch L(stream, "bn_partials_to_sums", 0, 0);
  bg::launch(partials_to_sums_kernel, dim3(bg::cdiv(C, kT / 64)), dim3(kT), 0, L.s, partial, red_blocks(M, C), C, sums_d);
  return L.done("partials_to_sums_kernel");
}

int bg_bn_finalize_f32(const float* sums_d, int M_total, int C, float* save_mean, float* save_inv, float* moving_mean,
                       float* moving_var, float eps, float momentum, int unbiased, void* stream) {
  BG_REQUIRE(sums_d && save_mean && save_inv, BG_ERR_NULL, "bg_bn_finalize_f32: null pointer");
  BG_REQUIRE((moving_mean == nullptr) == (moving_var == nullptr), BG_ERR_NULL, "bg_bn_finalize_f32: moving_mean/var must both be given or both NULL");
  BG_REQUIRE(M_total > 0 && C > 0, BG_ERR_BAD_SHAPE, "bg_bn_finalize_f32: M_total=%d C=%d", M_total, C);
  bg::Launch L(stream, "bn_finalize", 0, 0);
  bg::launch(bn_finalize_sums_kernel, dim3(bg::cdiv(C, kT)), dim3(kT), 0, L.s, sums_d, M_total, C, save_mean, save_inv, moving_mean,
                     moving_var, eps, momentum, unbiased);
  return L.done("bn_finalize_sums_kernel");
}

int bg_bn_apply_f32(const float* x, float* y, int M, int C, const float* gamma, const float* beta, const float* mean,
                    const float* inv, float lrelu_alpha, void* stream) {
  BG_REQUIRE(x && y && gamma && beta && mean && inv, BG_ERR_NULL, "bg_bn_apply_f32: null pointer");
  BG_REQUIRE(M > 0 && C > 0, BG_ERR_BAD_SHAPE, "bg_bn_apply_f32: M=%d C=%d", M, C);
  const size_t total = (size_t)M * C;
  bg::Launch L(stream, "bn_apply_lrelu", 0, 8.0 * total);
  bg::launch(bn_apply_kernel, dim3(grid_for(total)), dim3(kT), 0, L.s, x, y, total, C, gamma, beta, mean, inv, 0, 0.f, lrelu_alpha);
  return L.done("bn_apply_kernel");
}

int bg_bn_finalize_apply_f32(const float* sums_d, int M_total, const float* x, float* y, int M, int C, const float* gamma, const float* beta,
                             float* save_mean, float* save_inv, float* moving_mean, float* moving_var, float eps, float momentum,
                             int unbiased, float lrelu_alpha, void* stream) {
  BG_REQUIRE(sums_d && x && y && gamma && beta && save_mean && save_inv, BG_ERR_NULL, "bg_bn_finalize_apply_f32: null pointer");
  BG_REQUIRE(M > 0 && C > 0 && M_total >= M, BG_ERR_BAD_SHAPE, "bg_bn_finalize_apply_f32: M=%d M_total=%d C=%d", M, M_total, C);
  const size_t total = (size_t)M * C;
  bg::Launch L(stream, "bn_finalize_apply", 0, 8.0 * total);
  bg::launch(bn_finalize_apply_kernel, dim3(grid_for(total)), dim3(kT), 0, L.s, sums_d, M_total, x, y, total, C, gamma, beta, save_mean, save_inv,
             moving_mean, moving_var, eps, momentum, unbiased, lrelu_alpha);
  return L.done("bn_finalize_apply_kernel");
}

int bg_bn_bwd_stats_f32(const float* dy, const float* y, const float* x, int M, int C, const float* gamma, const float* beta,
                        const float* save_mean, const float* save_inv, float lrelu_alpha, float* sums_d, void* ws_d, size_t ws_bytes,
                        void* stream) {
  BG_REQUIRE(dy && (y || (gamma && beta)) && x && save_mean && save_inv && sums_d, BG_ERR_NULL, "bg_bn_bwd_stats_f32: null pointer");
  BG_REQUIRE(y || lrelu_alpha >= 0.f, BG_ERR_UNSUPPORTED, "bg_bn_bwd_stats_f32: the sign of y follows from x only for lrelu_alpha >= 0");
  BG_REQUIRE(M > 0 && C > 0, BG_ERR_BAD_SHAPE, "bg_bn_bwd_stats_f32: M=%d C=%d", M, C);
  BG_REQUIRE(ws_d && ws_bytes >= bg_bn_workspace_bytes(M, C), BG_ERR_WORKSPACE, "bg_bn_bwd_stats_f32: workspace too small");
  const int nblk = red_blocks(M, C);
  float* partial = static_cast<float*>(ws_d);
  {
    bg::Launch L(stream, "bn_bwd_partial", 0, (y ? 12.0 : 8.0) * M * C);
    if (flat_ok(M, C) && bg::aligned16(dy) && bg::aligned16(y) && bg::aligned16(x) && bg::aligned16(save_mean) && bg::aligned16(save_inv))
      bg::launch(bn_bwd_flat_kernel, dim3(nblk), dim3(kT), 0, L.s, dy, y, x, M, C, gamma, beta, save_mean, save_inv, lrelu_alpha, partial);
    else
      bg::launch(bn_bwd_partial_kernel, dim3(nblk, bg::cdiv(C, 64)), dim3(kT), 0, L.s, dy, y, x, M, C, gamma, beta, save_mean, save_inv,
                         lrelu_alpha, partial);
    int rc = L.done("bn_bwd_partial_kernel");
    if (rc) return rc;
  }
  bg::Launch L(stream, "bn_partials_to_sums", 0, 0);
  bg::launch(partials_to_sums_kernel, dim3(bg::cdiv(C, kT / 64)), dim3(kT), 0, L.s, partial, nblk, C, sums_d);
  return L.done("partials_to_sums_kernel");
}

int bg_bn_bwd_apply_f32(const float* dy, const float* y, const float* x, float* dx, int M, int M_total, int C, const float* gamma,
                        const float* beta, const float* save_mean, const float* save_inv, const float* sums_d, float lrelu_alpha,
                        void* stream) {
  BG_REQUIRE(dy && (y || beta) && x && dx && gamma && save_mean && save_inv && sums_d, BG_ERR_NULL, "bg_bn_bwd_apply_f32: null pointer");
  BG_REQUIRE(y || lrelu_alpha >= 0.f, BG_ERR_UNSUPPORTED, "bg_bn_bwd_apply_f32: the sign of y follows from x only for lrelu_alpha >= 0");
  BG_REQUIRE(M > 0 && M_total >= M && C > 0, BG_ERR_BAD_SHAPE, "bg_bn_bwd_apply_f32: M=%d M_total=%d C=%d", M, M_total, C);
  const size_t total = (size_t)M * C;
  bg::Launch L(stream, "bn_bwd_apply", 0, (y ? 16.0 : 12.0) * total);
  bg::launch(bn_bwd_apply_kernel, dim3(grid_for(total)), dim3(kT), 0, L.s, dy, y, x, dx, total, M_total, C, gamma, beta, save_mean, save_inv,
                     sums_d + C /* dgamma = sum dz*xhat */, sums_d /* dbeta = sum dz */, lrelu_alpha);
  return L.done("bn_bwd_apply_kernel");
}

int bg_bn_param_grads_f32(const float* sums_d, int C, float scale, float* dgamma, float* dbeta, void* stream) {
  BG_REQUIRE(sums_d && dgamma && dbeta, BG_ERR_NULL, "bg_bn_param_grads_f32: null pointer");
  BG_REQUIRE(C > 0, BG_ERR_BAD_SHAPE, "bg_bn_param_grads_f32: C=%d", C);
  bg::Launch L(stream, "bn_param_grads", 0, 0);
  bg::launch(bn_param_grads_kernel, dim3(bg::cdiv(C, kT)), dim3(kT), 0, L.s, sums_d, C, scale, dgamma, dbeta);
  return L.done("bn_param_grads_kernel");
}

int bg_lerp_f32(const float* r, const float* f, const float* alpha_b, float* xhat, int B, int n_per, void* stream) {
  BG_POINTWISE_PROLOGUE("bg_lerp_f32", r && f && alpha_b && xhat, (long)B * n_per);
  const size_t total = (size_t)B * n_per;
  bg::Launch L(stream, "lerp", 0, 12.0 * total);
  bg::launch(lerp_kernel, dim3(grid_for(total)), dim3(kT), 0, L.s, r, f, alpha_b, xhat, total, n_per);
  return L.done("lerp_kernel");
}

int bg_row_norm_f32(const float* g, float* norm_b, int B, int n_per, void* stream) {
  BG_POINTWISE_PROLOGUE("bg_row_norm_f32", g && norm_b, (long)B * n_per);
  bg::Launch L(stream, "row_norm", 0, 4.0 * B * n_per);
  bg::launch(row_norm_kernel, dim3(B), dim3(1024), 0, L.s, g, norm_b, n_per);
  return L.done("row_norm_kernel");
}

int bg_gp_seed_f32(const float* g, const float* norm_b, float coef, float* out, int B, int n_per, void* stream) {
  BG_POINTWISE_PROLOGUE("bg_gp_seed_f32", g && norm_b && out, (long)B * n_per);
  const size_t total = (size_t)B * n_per;
  bg::Launch L(stream, "gp_seed", 0, 8.0 * total);
  bg::launch(gp_seed_kernel<false>, dim3(grid_for(total)), dim3(kT), 0, L.s, g, norm_b, coef, out, total, n_per);
  return L.done("gp_seed_kernel");
}

int bg_gp_seed_guarded_f32(const float* g, const float* norm_b, float coef, float* out, int B, int n_per, void* stream) {
  BG_POINTWISE_PROLOGUE("bg_gp_seed_guarded_f32", g && norm_b && out, (long)B * n_per);
  const size_t total = (size_t)B * n_per;
  bg::Launch L(stream, "gp_seed", 0, 8.0 * total);
  bg::launch(gp_seed_kernel<true>, dim3(grid_for(total)), dim3(kT), 0, L.s, g, norm_b, coef, out, total, n_per);
  return L.done("gp_seed_kernel");
}

int bg_mul_grad_f32(const float* d, const float* ref, const uint8_t* keep, float alpha, float scale, float* out, size_t n,
                    void* stream) {
  BG_POINTWISE_PROLOGUE("bg_mul_grad_f32", d && ref && out, n);
  bg::Launch L(stream, "mul_grad", 0, 12.0 * n);
  bg::launch(mul_grad_kernel, dim3(grid_for(n)), dim3(kT), 0, L.s, d, ref, keep, alpha, scale, out, n);
  return L.done("mul_grad_kernel");
}

int bg_tanh_bwd_f32(const float* dy, const float* y, float* out, size_t n, void* stream) {
  BG_POINTWISE_PROLOGUE("bg_tanh_bwd_f32", dy && y && out, n);
  bg::Launch L(stream, "tanh_bwd", 0, 12.0 * n);
  bg::launch(tanh_bwd_kernel, dim3(grid_for(n)), dim3(kT), 0, L.s, dy, y, out, n);
  return L.done("tanh_bwd_kernel");
}

int bg_outer_f32(const float* s_b, const float* w_k, float* out, int B, int K, void* stream) {
  BG_POINTWISE_PROLOGUE("bg_outer_f32", s_b && w_k && out, (long)B * K);
  const size_t total = (size_t)B * K;
  bg::Launch L(stream, "outer", 0, 4.0 * total);
  bg::launch(outer_kernel, dim3(grid_for(total)), dim3(kT), 0, L.s, s_b, w_k, out, total, K);
  return L.done("outer_kernel");
}

int bg_fill_f32(float* x, float v, size_t n, void* stream) {
  BG_POINTWISE_PROLOGUE("bg_fill_f32", x, n);
  bg::Launch L(stream, "fill", 0, 4.0 * n);
  bg::launch(fill_kernel, dim3(grid_for(n)), dim3(kT), 0, L.s, x, v, n);
  return L.done("fill_kernel");
}

int bg_copy_f32(float* dst, const float* src, size_t n, void* stream) {
  BG_POINTWISE_PROLOGUE("bg_copy_f32", dst && src, n);
  bg::Launch L(stream, "copy", 0, 8.0 * n);
  const int vec = bg::aligned16(dst) && bg::aligned16(src);
  bg::launch(copy_kernel, dim3(grid_for(n, vec ? 4 : 1)), dim3(kT), 0, L.s, dst, src, n, vec);
  return L.done("copy_kernel");
}

int bg_scale_f32(float* x, float v, size_t n, void* stream) {
  BG_POINTWISE_PROLOGUE("bg_scale_f32", x, n);
  bg::Launch L(stream, "scale", 0, 8.0 * n);
  bg::launch(scale_kernel, dim3(grid_for(n)), dim3(kT), 0, L.s, x, v, n);
  return L.done("scale_kernel");
}

int bg_wgangp_d_loss(const float* fs, const float* rs, const float* norm_b, int B, float inv_gbs, float gp_coef, float e_drift,
                     float vec_scale, float* dfs, float* drs, float* metrics_d, void* stream) {
  BG_POINTWISE_PROLOGUE("bg_wgangp_d_loss", fs && rs && dfs && drs && metrics_d, B);
  bg::Launch L(stream, "d_loss", 0, 0);
  bg::launch(d_loss_kernel, dim3(1), dim3(kT), 0, L.s, fs, rs, norm_b, B, inv_gbs, gp_coef, e_drift, vec_scale, dfs, drs, metrics_d);
  return L.done("d_loss_kernel");
}

int bg_wgan_g_loss(const float* s, int B, float inv_gbs, float* ds, float* metrics_d, void* stream) {
  BG_POINTWISE_PROLOGUE("bg_wgan_g_loss", s && ds && metrics_d, B);
  bg::Launch L(stream, "g_loss", 0, 0);
  bg::launch(g_loss_kernel, dim3(1), dim3(kT), 0, L.s, s, B, inv_gbs, ds, metrics_d);
  return L.done("g_loss_kernel");
}

int bg_u8_normalize_resize_f32(const uint8_t* src, float* dst, int B, int Hs, int Ws, int C, int Hd, int Wd, void* stream) {
  BG_REQUIRE(src && dst, BG_ERR_NULL, "bg_u8_normalize_resize_f32: null pointer");
  BG_REQUIRE(B > 0 && Hs > 0 && Ws > 0 && C > 0 && Hd > 0 && Wd > 0, BG_ERR_BAD_SHAPE, "bg_u8_normalize_resize_f32: B=%d %dx%dx%d -> %dx%d", B, Hs, Ws, C, Hd, Wd);
  const size_t total = (size_t)B * Hd * Wd * C;
  bg::Launch L(stream, "u8_normalize_resize", 0, (double)B * Hs * Ws * C + 4.0 * total);
  bg::launch(u8_normalize_resize_kernel, dim3(grid_for(total)), dim3(kT), 0, L.s, src, dst, total, Hs, Ws, C, Hd, Wd,
                     (float)Hs / (float)Hd, (float)Ws / (float)Wd);
  return L.done("u8_normalize_resize_kernel");
}

int bg_adam_f32(float* theta, float* m, float* v, const float* g, size_t n, float lr_t, float b1, float b2, float eps,
                void* stream) {
  BG_POINTWISE_PROLOGUE("bg_adam_f32", theta && m && v && g, n);
  bg::Launch L(stream, "adam", 0, 28.0 * n);
  const int slot = bg::take_bind(BG_BIND_ADAM_LR);        // step program: lr_t re-read from a slot before every replay
  bg::launch(adam_kernel, dim3(grid_for(n)), dim3(kT), 0, L.s, theta, m, v, g, n, lr_t, b1, b2, eps);
  bg::bind_last(5, bg::BIND_F32_FROM_F64, slot);
  return L.done("adam_kernel");
}

int bg_uniform_f32(float* out, size_t n, uint64_t seed, uint64_t offset, void* stream) {
  BG_POINTWISE_PROLOGUE("bg_uniform_f32", out, n);
  bg::Launch L(stream, "rng_uniform", 0, 4.0 * n);
  const int slot = bg::take_bind(BG_BIND_RNG_OFFSET);     // step program: the counter offset advances per replay
  bg::launch(uniform_kernel, dim3(grid_for(n, 4)), dim3(kT), 0, L.s, out, n, seed, offset);
  bg::bind_last(3, bg::BIND_U64, slot);
  return L.done("uniform_kernel");
}

int bg_keep_mask_u8(uint8_t* out, size_t n, float keep_prob, uint64_t seed, uint64_t offset, void* stream) {
  BG_POINTWISE_PROLOGUE("bg_keep_mask_u8", out, n);
  BG_REQUIRE(keep_prob > 0.f && keep_prob <= 1.f, BG_ERR_BAD_SHAPE, "bg_keep_mask_u8: keep_prob=%g", keep_prob);
  bg::Launch L(stream, "rng_keep_mask", 0, 1.0 * n);
  const int slot = bg::take_bind(BG_BIND_RNG_OFFSET);
  bg::launch(keep_mask_kernel, dim3(grid_for(n, 16)), dim3(kT), 0, L.s, out, n, keep_prob, seed, offset);
  bg::bind_last(4, bg::BIND_U64, slot);
  return L.done("keep_mask_kernel");
}

}  // extern "C"
