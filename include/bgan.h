/*
 * bgan.h -- C ABI of libbgan_hip.so: the MI355X (gfx950) kernels behind blurred-GAN's
 * WGAN-GP training step.
 *
 * The reference (lebrice/blurred-GAN) has no FFI seam of its own: its hot path is Python that
 * calls TensorFlow ops.  This header is the seam the build introduces: one
 * extern "C" entry point per TF op family the reference's step invokes, each citing the
 * reference call site it replaces.  Conventions:
 *
 *   - every pointer named *_d / typed `const float*` etc. is a raw DEVICE address (HBM); the caller
 *     (Python via torch.Tensor.data_ptr()) owns every buffer including workspaces; the library
 *     keeps no device memory between calls;
 *   - tensors are dense NHWC float32; Conv2D kernels are [kh,kw,Cin,Cout], Conv2DTranspose kernels
 *     [kh,kw,Cout,Cin], Dense kernels [in,out] -- exactly as TF/Keras stores them;
 *   - `stream` is a hipStream_t (torch.cuda.current_stream().cuda_stream); launches are
 *     asynchronous, nothing synchronises the device;
 *   - every function returns 0 on success or a negative bg_status; bg_last_error() gives the
 *     thread-local message; nothing aborts or throws across the boundary.
 */
#ifndef BGAN_H
#define BGAN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BG_ABI_VERSION 5

typedef enum {
  BG_OK = 0,
  BG_ERR_BAD_SHAPE = -1,      /* non-positive / inconsistent dimensions                       */
  BG_ERR_BAD_ALIGNMENT = -2,  /* pointer not aligned as the kernel requires (16 B)             */
  BG_ERR_UNSUPPORTED = -3,    /* valid request this build has no kernel for                    */
  BG_ERR_HIP = -4,            /* a HIP runtime call failed (message has hipGetErrorString)     */
  BG_ERR_WORKSPACE = -5,      /* workspace missing or too small (see *_workspace_bytes)        */
  BG_ERR_NULL = -6,           /* required pointer is NULL                                      */
  BG_ERR_RCCL = -7            /* RCCL missing or a collective call failed (message has the RCCL string) */
} bg_status;

/* ---- meta ------------------------------------------------------------------------------- */
int bg_version(void);
const char* bg_last_error(void);
const char* bg_status_string(int status);

/* ---- profiling hooks (bench.py roofline: per-kernel HIP-event timing on the launch stream) */
int bg_prof_enable(int on);                       /* record an event pair around every launch */
int bg_prof_reset(void);
int bg_prof_count(void);                          /* synchronises the recorded events          */
int bg_prof_get(int i, char* name, int name_cap, float* ms, double* flops, double* bytes);
/* flops record i ISSUED on the matrix pipe: whole MFMA tiles (row / column padding counted), minus the padding taps the
   position-major tiles skip.  Equals the algorithmic figure of bg_prof_get for kernels that do not report their own. */
int bg_prof_get_exec(int i, double* exec_flops);
/* flops of record i that multiply REAL data: SURVEY 8d's F_l (bg_prof_get) charges all 25 taps at every output position of a
   SAME convolution (demo_celeba.py:62,99); the taps that land on the zero padding -- 51 % of them on a 4x4 map, 28 % on 8x8 --
   are work no implementation has to do.  "useful" counts exactly the (output pixel, tap) pairs inside the image, with no tile
   padding either, so  useful <= executed-or-algorithmic  and a rate priced by it cannot read above the roof.  Equals the
   algorithmic figure for launches that are not convolutions.  bg_conv2d_useful_flops is the closed form (conv input side
   H x W x Cin; one count serves the forward, the data gradient / transposed convolution and the filter gradient). */
int bg_prof_get_useful(int i, double* useful_flops);
double bg_conv2d_useful_flops(int B, int H, int W, int Cin, int Cout, int ksize, int stride);

/* Named ranges on the profiler timeline (rocprofv3 --marker-trace): D-step / G-step / blur / all-reduce of one train_on_batch
   (wgan.py:86-114).  roctx (librocprofiler-sdk-roctx.so) is bound with dlopen on first use; without it, or with
   bg_range_enable(0) (the default), both calls return immediately.  bg_range_enable returns 1 when ranges will be emitted. */
int bg_range_enable(int on);
int bg_range_push(const char* name);
int bg_range_pop(void);

/* ---- Gaussian blur: gaussian_blur.py:15-132 ---------------------------------------------- */
/* gaussian_blur.py:21-31,58-72 (appropriate_kernel_size, appropriate_std, clip, max): host maths, float32. */
int bg_blur_policy(float sigma, int H, int W, float* kernel_size, float* sigma_eff, int* n_taps);
/* gaussian_blur.py:83-88 (gaussian_kernel_1d): writes n_taps float32 weights to HOST memory. */
int bg_gauss_kernel_1d(float sigma_eff, float kernel_size, float* taps_host, int cap, int* n_taps);
/* gaussian_blur.py:91-132 (two tf.nn.depthwise_conv2d, SAME, zero padding): y = blur(x), NHWC.
 * taps_d: n_taps device floats.  tmp_d: scratch of the same size as x (may be NULL when the whole
 * image fits the fused kernel, see bg_blur_workspace_bytes).  The op is self-adjoint, so the same
 * call is its own backward. */
size_t bg_blur_workspace_bytes(int B, int H, int W, int C, int n_taps);
int bg_blur_nhwc_f32(const float* x, float* y, int B, int H, int W, int C,
                     const float* taps_d, int n_taps, float* tmp_d, void* stream);

/* The critic's batch of one discriminator_step in ONE launch (wgan.py:138-139 fakes and reals, wgan.py:239-240 x-hat):
 *   y3[0:B] = blur(f), y3[B:2B] = blur(r), y3[2B:3B] = blur(r + alpha[b] * (f - r))
 * x-hat is formed while its rows are staged (the expression of bg_lerp_f32: bit-identical to bg_lerp_f32 followed by bg_blur_nhwc_f32)
 * and never written to memory.  Only geometries of the row-block kernel (images up to 64 x 64, <= 4 channels, >= 13 taps;
 * bg_blur3_lerp_supported tells) -- anything else returns BG_ERR_UNSUPPORTED and the caller runs the three separate calls. */
int bg_blur3_lerp_supported(int B, int H, int W, int C, int n_taps);
int bg_blur3_lerp_nhwc_f32(const float* f, const float* r, const float* alpha_b, float* y3, int B, int H, int W, int C,
                           const float* taps_d, int n_taps, void* stream);

/* ---- convolution family: layers.Conv2D / Conv2DTranspose (demo_celeba.py:62-119,
 *      demo_mnist.py:60-81) and their tape gradients (wgan.py:140,166,244) -------------------
 * Geometry is TF 'SAME' for a kxk kernel (k odd, k*k <= 25), stride 1 or 2:
 *   Ho = ceil(H/s), pad_total = max((Ho-1)s + k - H, 0), pad_before = pad_total/2.
 * H, W, Cin always describe the conv's INPUT side, Ho/Wo/Cout its OUTPUT side, also for the
 * backward calls.  A Conv2DTranspose with kernel [k,k,Cout_t,Cin_t] is the data-gradient of the
 * conv whose kernel is that same array read as [k,k,Cin=Cout_t,Cout=Cin_t]; call bg_conv2d_bwd_data
 * for its forward, bg_conv2d_fwd for its data-gradient, bg_conv2d_bwd_filter(x := dy_t, dy := x_t)
 * for its filter-gradient.
 *
 * Weight operand layout expected by the MFMA kernels is "NK": [tap][n][k] with the contraction
 * index k contiguous.  For bg_conv2d_bwd_data that IS the TF layout [kh,kw,Cin,Cout]; for
 * bg_conv2d_fwd it is the last-two-dims transpose [kh,kw,Cout,Cin] (bg_transpose_last2 makes it).
 */
typedef enum {
  BG_EPI_NONE = 0,        /* y = acc (+ bias)                                                   */
  BG_EPI_BIAS_LRELU = 1,  /* y = lrelu(acc + bias); then y *= keep ? scale : 0 when keep != NULL */
  BG_EPI_MUL_GRAD = 2,    /* y = acc * (ref > 0 ? 1 : alpha) [* keep ? scale : 0].  `ref` MAY ALIAS the output (the penalty's
                           * linearised forward overwrites the activations whose signs it consumes): every kernel reads
                           * ref[i] in the thread that stores y[i], before that store                      */
  BG_EPI_TANH = 3,        /* y = tanh(acc + bias)                                               */
  BG_EPI_AFFINE_LRELU = 4 /* y = lrelu(acc * ref[c] + bias[c]): inference BatchNormalization folded (bg_bn_fold_f32), ref = per-channel scale */
} bg_epi_mode;

typedef struct {
  int mode;               /* bg_epi_mode                                                        */
  const float* bias;      /* [Cout] or NULL                                                     */
  const float* ref;       /* BG_EPI_MUL_GRAD: activation whose sign selects 1 / alpha, same shape as y */
  const uint8_t* keep;    /* dropout keep mask (1 = kept), same shape as y, or NULL             */
  float alpha;            /* LeakyReLU slope (Keras default 0.3)                                */
  float scale;            /* 1 / keep_prob                                                      */
  void* splitk_ws;        /* optional scratch for split-K partial sums (small-M layers); NULL = never split */
  size_t splitk_ws_bytes; /* >= bg_conv2d_splitk_workspace_bytes(...) for the call to use split-K      */
  size_t keep_elems;      /* the keep mask covers output elements [0, keep_elems) only (a batch whose leading samples ran with
                           * Dropout and whose trailing samples without: fake+real and x-hat of wgan.py:138,240 in one pass);
                           * 0 = every element */
  float* stats;           /* optional (mode BG_EPI_NONE, no bias): the MFMA gather kernel also leaves per-workgroup column sums and
                           * sums of squares of what it stores -- partial[row][2][Cout] -- so that the BatchNormalization that
                           * follows (demo_celeba.py:62-90) needs no statistics pass over the tensor; `*stats_rows` tells how many
                           * rows THIS call wrote (0: this geometry took another kernel, run the normal pass) */
  size_t stats_capacity;  /* floats available behind `stats` */
  int* stats_rows;        /* HOST pointer, out (required when `stats` is set): rows of `stats` this call wrote.  Returned through
                           * the call itself -- ABI 2 kept it in per-thread last-call state (bg_conv2d_stats_rows, removed in 3) */
} bg_epilogue;

/* bytes of split-K scratch the forward (bwd_data = 0) / data-gradient (bwd_data = 1) call can use; 0 = never splits */
size_t bg_conv2d_splitk_workspace_bytes(int bwd_data, int B, int H, int W, int Cin, int Cout, int ksize, int stride);
/* Alignment: x / w / y (dy / w / dx) must be 16-byte aligned (BG_ERR_BAD_ALIGNMENT otherwise).  The epilogue's operands may sit
 * anywhere, but the matrix-core kernels read them four at a time: they are taken when bias / ref / the split-K scratch are
 * 16-byte aligned, the mask 4-byte aligned with keep_elems % 4 == 0, and the output channel count a multiple of 4 -- what a
 * framework allocator hands out; anything else runs on the direct kernel (same results, slower). */
/* y[B,Ho,Wo,Cout] = conv(x[B,H,W,Cin], w) ; wT_d = [k*k][Cout][Cin] */
int bg_conv2d_fwd(const float* x, const float* wT_d, float* y, int B, int H, int W, int Cin, int Cout,
                  int ksize, int stride, const bg_epilogue* epi, void* stream);
/* dx[B,H,W,Cin] = conv^T(dy[B,Ho,Wo,Cout], w) ; w_d = [k*k][Cin][Cout] (TF Conv2D layout) */
int bg_conv2d_bwd_data(const float* dy, const float* w_d, float* dx, int B, int H, int W, int Cin, int Cout,
                       int ksize, int stride, const bg_epilogue* epi, void* stream);
/* dw[k,k,Cin,Cout] = beta*dw + scale * sum_pixels x (x) dy.  ws_d: bg_conv2d_bwd_filter_workspace_bytes */
size_t bg_conv2d_bwd_filter_workspace_bytes(int B, int H, int W, int Cin, int Cout, int ksize, int stride);
int bg_conv2d_bwd_filter(const float* x, const float* dy, float* dw, int B, int H, int W, int Cin, int Cout,
                         int ksize, int stride, float beta, float scale, void* ws_d, size_t ws_bytes, void* stream);
/* dst[t][c][r] = src[t][r][c] */
int bg_transpose_last2(const float* src, float* dst, int T, int R, int C, void* stream);
/* The same for every conv kernel of a network in one launch (after each optimiser step: layers.ParamStore.refresh_transposed).
 * desc_d: n x 6 int32 on the device {src_off, dst_off, T, R, C, first_tile}: float offsets from the two bases (multiples of 4),
 * 64x64 tiles numbered from first_tile in [t][ceil(R/64)][ceil(C/64)] order, first_tile ascending; total_tiles = their sum. */
int bg_transpose_last2_batched(const float* src_base, float* dst_base, const int* desc_d, int n, int total_tiles, void* stream);

/* ---- Dense (demo_celeba.py:55,124): row-major C[M,N] = op(A)[M,K] * op(B)[K,N] (+ bias[N]) --- */
int bg_gemm_f32(const float* A, const float* Bm, float* C, int M, int N, int K, int transA, int transB,
                const float* bias, float beta, float scale, void* stream);

/* ---- reductions: column sums of a row-major [M,N] matrix (bias grads, BN statistics, JVP tail) */
size_t bg_colsum_workspace_bytes(int M, int N);
/* out[n] = beta*out[n] + scale * sum_m f(x[m,n]) ; square != 0 sums x^2 */
int bg_colsum_f32(const float* x, float* out, int M, int N, int square, float beta, float scale,
                  void* ws_d, size_t ws_bytes, void* stream);

/* ---- BatchNormalization + LeakyReLU (demo_celeba.py:56-90), x viewed as [M, C] --------------- */
/* training forward: batch mean / biased var -> y = lrelu(gamma*(x-mean)*inv + beta); saves mean, inv;
 * moving stats <- moving*momentum + stat*(1-momentum), var scaled by M/(M-1) when unbiased != 0. */
size_t bg_bn_workspace_bytes(int M, int C);
int bg_bn_train_fwd(const float* x, float* y, int M, int C, const float* gamma, const float* beta,
                    float* moving_mean, float* moving_var, float* save_mean, float* save_inv,
                    float eps, float momentum, int unbiased, float lrelu_alpha,
                    void* ws_d, size_t ws_bytes, void* stream);
/* inference forward with moving stats (generator inside the D-step, wgan.py:135). */
int bg_bn_infer_fwd(const float* x, float* y, int M, int C, const float* gamma, const float* beta,
                    const float* moving_mean, const float* moving_var, float eps, float lrelu_alpha, void* stream);
/* backward through lrelu + training BN: dy is the gradient w.r.t. y (post-lrelu), y the saved output.
 * y may be NULL (here and in bg_bn_bwd_stats_f32 / bg_bn_bwd_apply_f32) when beta is given and lrelu_alpha >= 0: the sign of y is
 * then re-derived from x with the forward's own expression gamma * ((x - mean) * inv) + beta -- the same operations on the same
 * inputs, so the result is bit-identical -- and each pass reads one tensor less. */
int bg_bn_train_bwd(const float* dy, const float* y, const float* x, float* dx, int M, int C,
                    const float* gamma, const float* beta, const float* save_mean, const float* save_inv,
                    float* dgamma, float* dbeta, float lrelu_alpha, void* ws_d, size_t ws_bytes, void* stream);

/* inference BatchNormalization as a per-channel affine: scale = gamma / sqrt(moving_var + eps), shift = beta - moving_mean * scale
 * (feeds BG_EPI_AFFINE_LRELU so the generator's inference forward inside the D-step, wgan.py:135, needs no separate BN pass) */
int bg_bn_fold_f32(const float* gamma, const float* beta, const float* moving_mean, const float* moving_var, float eps, int C,
                   float* scale_out, float* shift_out, void* stream);
/* the same for n <= 8 layers in ONE launch (HOST arrays of n device pointers / values; same arithmetic per channel): every
 * BatchNorm of the generator's inference forward is folded before its first conv */
int bg_bn_fold_many_f32(int n, const float* const* gamma, const float* const* beta, const float* const* moving_mean,
                        const float* const* moving_var, const float* eps, const int* C, float* const* scale_out,
                        float* const* shift_out, void* stream);
/* The same BatchNormalization in separable pieces, for data-parallel SyncBN: the per-channel sums are
 * all-reduced by the caller between the pieces.  sums_d = [2*C]: forward {sum x, sum x^2}; backward {sum dz, sum dz*xhat}
 * with dz = dy * lrelu'(y).  M_total = rows summed over all replicas. */
int bg_bn_stats_f32(const float* x, int M, int C, float* sums_d, void* ws_d, size_t ws_bytes, void* stream);
int bg_bn_finalize_f32(const float* sums_d, int M_total, int C, float* save_mean, float* save_inv, float* moving_mean,
                       float* moving_var, float eps, float momentum, int unbiased, void* stream);
int bg_bn_apply_f32(const float* x, float* y, int M, int C, const float* gamma, const float* beta, const float* mean,
                    const float* inv, float lrelu_alpha, void* stream);
/* bg_bn_finalize_f32 + bg_bn_apply_f32 in ONE launch (same arithmetic, bit-identical y / saved / moving statistics): behind a conv
 * that left its statistics in the epilogue (bg_epilogue.stats) the SyncBN forward chain is bg_bn_sums_from_partials -> all-reduce
 * -> this, two launches per layer beside the exchange. */
int bg_bn_finalize_apply_f32(const float* sums_d, int M_total, const float* x, float* y, int M, int C, const float* gamma,
                             const float* beta, float* save_mean, float* save_inv, float* moving_mean, float* moving_var, float eps,
                             float momentum, int unbiased, float lrelu_alpha, void* stream);
/* training-mode BatchNormalization + LeakyReLU from statistics PARTIALS partial[nrows][2][C] (column sums, sums of squares) left
   by the producing conv (bg_epilogue.stats): finalize (+ moving statistics) and apply, no statistics pass over x */
int bg_bn_train_fwd_partials(const float* partial_d, int nrows, const float* x, float* y, int M, int C, const float* gamma,
                             const float* beta, float* moving_mean, float* moving_var, float* save_mean, float* save_inv, float eps,
                             float momentum, int unbiased, float lrelu_alpha, void* stream);
/* sums[0:C] = column sums, sums[C:2C] = sums of squares out of the same partials (SyncBN: what gets all-reduced) */
int bg_bn_sums_from_partials(const float* partial_d, int nrows, int C, float* sums_d, void* stream);
int bg_bn_bwd_stats_f32(const float* dy, const float* y, const float* x, int M, int C, const float* gamma, const float* beta,
                        const float* save_mean, const float* save_inv, float lrelu_alpha, float* sums_d, void* ws_d, size_t ws_bytes,
                        void* stream);
int bg_bn_bwd_apply_f32(const float* dy, const float* y, const float* x, float* dx, int M, int M_total, int C, const float* gamma,
                        const float* beta, const float* save_mean, const float* save_inv, const float* sums_d, float lrelu_alpha,
                        void* stream);
/* dbeta[c] = scale * sums[c], dgamma[c] = scale * sums[C + c]: gamma / beta gradients out of the all-reduced backward sums
   (scale = 1 / replicas, so that the SUM all-reduce of the flat gradient buffer restores the global value) */
int bg_bn_param_grads_f32(const float* sums_d, int C, float scale, float* dgamma, float* dbeta, void* stream);

/* ---- pointwise ------------------------------------------------------------------------------ */
/* wgan.py:239: xhat[b,:] = r[b,:] + alpha[b] * (f[b,:] - r[b,:]) */
int bg_lerp_f32(const float* r, const float* f, const float* alpha_b, float* xhat, int B, int n_per, void* stream);
/* wgan.py:245: norm[b] = ||g[b,:]||_2 */
int bg_row_norm_f32(const float* g, float* norm_b, int B, int n_per, void* stream);
/* second-order seed: out[b,:] = coef * (norm[b]-1)/norm[b] * g[b,:]   (d mean((n-1)^2) / dg) */
int bg_gp_seed_f32(const float* g, const float* norm_b, float coef, float* out, int B, int n_per, void* stream);
/* the same with the subgradient 0 for a sample whose norm is exactly 0 (the reference, like tf.norm's gradient at 0, yields NaN
   there and Adam then spreads it into every weight): build-side switch WGANGP(gp_zero_norm_guard=True), off by default */
int bg_gp_seed_guarded_f32(const float* g, const float* norm_b, float coef, float* out, int B, int n_per, void* stream);
/* out = d * (ref > 0 ? 1 : alpha) [* keep ? scale : 0] */
int bg_mul_grad_f32(const float* d, const float* ref, const uint8_t* keep, float alpha, float scale,
                    float* out, size_t n, void* stream);
/* out = dy * (1 - y^2) */
int bg_tanh_bwd_f32(const float* dy, const float* y, float* out, size_t n, void* stream);
/* out[b, k] = s[b] * w[k] */
int bg_outer_f32(const float* s_b, const float* w_k, float* out, int B, int K, void* stream);
int bg_fill_f32(float* x, float v, size_t n, void* stream);
int bg_scale_f32(float* x, float v, size_t n, void* stream);
/* dst[0:n] = src[0:n] as a KERNEL (a blur-less critic's [fakes; reals; x-hat] batch is assembled with it: a launch that a step
   program records like any other, which a runtime memcpy would not be) */
int bg_copy_f32(float* dst, const float* src, size_t n, void* stream);

/* ---- losses (wgan.py:128-130,155-157,272-285) ----------------------------------------------- */
/* From scores fs[B], rs[B] and the per-sample norms n[B] of the GP gradient:
 *   metrics_d[0..5] = mean(fs), mean(rs), disc_loss (mean of the [B] vector), gp_term, mean(norm_term), GP
 *   dfs[b] = vec_scale*inv_gbs + e_drift*sign(fs[b]);  drs[b] = -vec_scale*inv_gbs + e_drift*sign(rs[b])
 * vec_scale is B when the reference's [B]-vector loss quirk (Q1) is reproduced, 1 otherwise. */
int bg_wgangp_d_loss(const float* fs, const float* rs, const float* norm_b, int B, float inv_gbs,
                     float gp_coef, float e_drift, float vec_scale, float* dfs, float* drs,
                     float* metrics_d, void* stream);
/* metrics_d[0] = mean(s), metrics_d[1] = -sum(s)*inv_gbs ; ds[b] = -inv_gbs */
int bg_wgan_g_loss(const float* s, int B, float inv_gbs, float* ds, float* metrics_d, void* stream);

/* ---- input pipeline (demo_celeba.py:22-35, demo_mnist.py:24-31): uint8 NHWC -> float32, (x - 127.5) / 127.5, then
 *      tf.image.resize(bilinear, half-pixel centres) to [Hd, Wd] (normalise BEFORE resize, aspect not preserved).
 *      Hd == Hs && Wd == Ws is the MNIST pipeline (normalise only). */
int bg_u8_normalize_resize_f32(const uint8_t* src, float* dst, int B, int Hs, int Ws, int C, int Hd, int Wd, void* stream);

/* ---- optimiser: tf.keras.optimizers.Adam (wgan.py:56-61,141,167) ----------------------------- */
/* lr_t = lr*sqrt(1-b2^t)/(1-b1^t) is computed by the caller (host) per step. */
int bg_adam_f32(float* theta, float* m, float* v, const float* g, size_t n, float lr_t, float b1, float b2,
                float eps, void* stream);

/* ---- RNG: tf.random.uniform (wgan.py:118,237) and Dropout masks; counter-based, own stream ---- */
int bg_uniform_f32(float* out, size_t n, uint64_t seed, uint64_t offset, void* stream);
int bg_keep_mask_u8(uint8_t* out, size_t n, float keep_prob, uint64_t seed, uint64_t offset, void* stream);

/* ---- step programs: the step-level entry points of SURVEY.md 8b (bg_dstep / bg_gstep) -----------------------------------
 * The reference runs one Python pass over its TF ops per batch (wgan.py:86-114: discriminator_step :132-151, generator_step
 * :159-172).  With static shapes and caller-owned persistent buffers the launch list of such a step is the same every batch, so
 * it is recorded ONCE and replayed with ONE call:
 *   bg_program_record_begin(p); ...the entry points above, exactly as the eager step calls them...; bg_program_record_end(p);
 * While recording, every kernel launch of the calling thread is executed AND appended to `p` as (kernel, grid, block, LDS bytes,
 * a by-value copy of every kernel argument).  bg_program_replay(p, first, last, stream) issues nodes [first, last) (last < 0 =
 * all) on `stream` -- no geometry checks, tap tables, grid planning or host language in between.  What changes per step:
 *   - Adam's lr_t and the RNG counter offsets: announce bg_program_bind_next(what, slot) right before the call that owns the
 *     argument; the recorded launch then re-reads slots_f64[slot] (BG_BIND_ADAM_LR) / slots_u64[slot] (BG_BIND_RNG_OFFSET) before
 *     every replay.  The slot arrays belong to the program and are written directly by the host.
 *   - the blur taps' VALUES live in a caller-owned device buffer the host refreshes in stream order; a changed tap COUNT selects
 *     other kernels and needs a newly recorded program (the host keeps one per count).
 *   - collectives (data parallel) are the host's: it splits the replay at the node indices bg_program_size() returned when the
 *     collective was issued during recording.
 * Every device pointer seen while recording must stay valid for the life of the program (the host keeps the tensors alive).
 * bg_dstep / bg_gstep replay a whole program: one call per discriminator_step / generator_step.
 * bg_program_graph_launch: the same node range as one hipGraph (kernel nodes in a linear chain, bound arguments refreshed with
 * hipGraphExecKernelNodeSetParams); with bg_prof_enable(1) it falls back to the node-by-node replay.
 * Profiling brackets travel with the program: a replay under bg_prof_enable(1) yields the same records as the eager step.
 * Synchronisation: no call of this section waits for the device, with ONE exception -- destroying, evicting or re-recording a
 * program that has been launched in graph form (bg_program_destroy, bg_program_record_begin) waits for the STREAM of its last
 * bg_program_graph_launch (hipStreamSynchronize, never hipDeviceSynchronize) before the executable graphs are freed.  The
 * node-by-node replay (the default) never synchronises.
 * A binding that does not fit the launch it was announced for (wrong argument size, no launch at all) fails
 * bg_program_record_end: a silently dropped one would replay the recording step's value for ever. */
typedef struct bg_program bg_program;
#define BG_BIND_ADAM_LR 1     /* bg_adam_f32: lr_t <- (float) slots_f64[slot]                         */
#define BG_BIND_RNG_OFFSET 2  /* bg_uniform_f32, bg_keep_mask_u8: offset <- slots_u64[slot]            */
int bg_program_create(bg_program** out, int n_slots);
int bg_program_destroy(bg_program* p);
int bg_program_record_begin(bg_program* p);
int bg_program_record_end(bg_program* p);
int bg_program_size(const bg_program* p);       /* nodes so far (launches + profiling brackets): replay range boundaries */
int bg_program_launches(const bg_program* p);   /* kernel launches recorded                                               */
int bg_program_binds(const bg_program* p);      /* bound arguments recorded                                               */
int bg_program_bind_next(int what, int slot);
double* bg_program_slots_f64(bg_program* p);
uint64_t* bg_program_slots_u64(bg_program* p);
int bg_program_replay(bg_program* p, int first, int last, void* stream);
int bg_program_graph_launch(bg_program* p, int first, int last, void* stream);
int bg_dstep(bg_program* p, void* stream);      /* wgan.py:132-151 as recorded */
int bg_gstep(bg_program* p, void* stream);      /* wgan.py:159-172 as recorded */

/* ---- data-parallel exchange step (the reference never ran multi-GPU, demo_mnist.py:116) --------
 * One process per GPU.  Rank 0 makes an id and hands its BG_COMM_ID_BYTES bytes to the other ranks over the host's
 * own channel (env, file, torch.distributed store ...); every rank then calls bg_comm_init with the HIP device it
 * trains on current.  bg_allreduce_sum_f32 is the step's only collective: in-place SUM of a flat fp32 gradient range
 * over RCCL (xGMI inside a node), asynchronous on `stream`.  RCCL is loaded on first use (BGAN_RCCL_LIB overrides
 * the library name); BG_ERR_RCCL when it is absent or a call fails. */
#define BG_COMM_ID_BYTES 128
typedef struct bg_comm bg_comm;
int bg_comm_unique_id(unsigned char* id_out);
int bg_comm_init(bg_comm** out, int rank, int nranks, const unsigned char* id_bytes);
/* One communicator may be driven from TWO streams (the step issues its bucketed gradient reductions on a private stream and the
 * SyncBN statistics exchanges in line on the compute stream): the calls are then ordered by RCCL in ISSUE order on the host, so
 * every rank must issue them in the same order (the step program guarantees it); a host that cannot should keep to one stream. */
int bg_allreduce_sum_f32(bg_comm* comm, float* buf_d, size_t n, void* stream);
/* ranks and own rank as the communicator itself reports them (ncclCommCount / ncclCommUserRank): evidence for a bench line
   that the collective really ran over N peers (bench.py --gpus N "rccl_nranks"). */
int bg_comm_query(bg_comm* comm, int* nranks, int* rank);
int bg_comm_destroy(bg_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* BGAN_H */
