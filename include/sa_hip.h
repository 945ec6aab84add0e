/* sa_hip.h -- C ABI of libsa_hip.so, the MI355X (gfx950) compute library behind the
 * speech-anonymization ConvAE + gender-adversarial train step.
 *
 * The reference (viswavi/speech-anonymization) has no native / FFI boundary: the hot path is
 * PyTorch ops reached from three Python seams.  This header is the boundary a
 * maintainer binds instead (ctypes stub: INTEGRATION.md); each entry point cites the
 * reference code whose arithmetic it replaces.  Conventions:
 *   - plain C: pointers + sizes, no torch / HIP types; `stream` is a hipStream_t passed as void*
 *     (torch.cuda.current_stream().cuda_stream); every call only ENQUEUES work on it;
 *   - every buffer (inputs, outputs, saved tensors, workspaces) is caller-allocated device
 *     memory; the library allocates nothing and keeps no state besides kernel attributes --
 *     except, per process, ONE RCCL communicator + one side stream + two events between
 *     sa_comm_init and sa_comm_destroy (data-parallel exchange, at the end of this header);
 *   - return 0 on success, -EINVAL (-22) for bad arguments, -ENOSYS (-38) for a shape that is
 *     not instantiated, or -(hipError_t) from the launch; nothing throws across the ABI;
 *   - dtype: SA_F32 (0) or SA_BF16 (1) = storage type of activations and packed weights;
 *     accumulation and all statistics are fp32 (fp64 in the tiny finalisers);
 *   - activations are CHANNELS-LAST [B][L][C]; the reference's [B][C][L] tensors are never
 *     materialised (C = 1 at both ends of the auto-encoder, so the module boundary
 *     feats[B][T][80] -> recon[B][T][80] needs no conversion).
 */
#ifndef SA_HIP_H
#define SA_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

#define SA_F32 0
#define SA_BF16 1
#define SA_BF16X3 2   /* fp32 storage, split-bf16 operands, 3 bf16 MFMAs per k-step */
#define SA_BF16X1F 3  /* fp32 storage, operands rounded to bf16 once, 1 bf16 MFMA (sa_wgrad only) */
#define SA_FP8 4      /* FORWARD-OPERAND EXPERIMENT, not a training mode and not BASELINE config 5 (which is
                       * listed as not built: DESIGN.md 3): bf16 storage; MFMA operands OCP e4m3 -- the
                       * activation rows are quantised unscaled while they are staged, the weight image is
                       * e4m3 scaled by a per-tensor power of two (SaConvArgs.wscale, undone in the
                       * epilogue); fp32 accumulation and statistics.  Forward-type launches of
                       * sa_conv_gemm only; gradients stay on SA_BF16.  Measured: same speed as SA_BF16
                       * (the non-scaled K = 16 fp8 MFMA runs at the bf16 rate), reconstruction 2.4e-2
                       * off the fp32 oracle, the adversarial (classifier-branch) gradient direction lost
                       * (cosine 0.13).  Kept because the kernel is exact against emulated e4m3 operands
                       * (tests) and pins the e4m3 fragment layout for a block-scaled follow-up. */
#define SA_F64 5      /* sa_comm_allreduce only: the SyncBatchNorm element counts */
#define SA_MAX_TAPS 5
#define SA_COMM_ID_BYTES 128

/* ---- implicit-GEMM convolution (sa_conv_gemm.hip) -------------------------------------
 * Row-gather GEMM covering nn.Conv1d, nn.ConvTranspose1d(stride 2) and both data gradients
 * (models/ConvAutoEncoder.py:141-172 encoder/decoder, :33-43 TDNN; backward via
 * speechbrain_convae_train.py:241).  For base row m and phase ph < U the output row is
 * o = m*U + ph and
 *   y[b,o,co] = bias[co] + sum_{t<ntaps[ph]} sum_ci P(x)[b, m*SA + off[ph][t], ci] * W[widx[ph][t]][ci][co]
 * rows outside [0,Lin) are zero.  P = prologue: v*s1[b][ci]+t1[b][ci] -> x*sigmoid(x) if swish
 * -> v*s2[ci]+t2[ci] (any pointer may be NULL).  Epilogue: +bias, ReLU if relu, store, and if
 * stats != NULL per-tile partial (sum, sumsq) of the stored values -> stats[B][ntiles][COUT][2]
 * with ntiles = sa_conv_gemm_ntiles(cin, cout, u, Lout). */
typedef struct SaTaps {
  int ntaps[2];
  int off[2][SA_MAX_TAPS];
  int widx[2][SA_MAX_TAPS];
} SaTaps;

typedef struct SaConvArgs {
  const void* x;
  const void* wp;            /* sa_pack_weights image */
  const float* bias;
  void* y;
  const float* s1; const float* t1;
  const float* s2; const float* t2;
  int swish;
  int relu;
  float* stats;
  int B, Lin, Lout, ntiles;  /* ntiles, rowmin, nrows, wlo_off are filled in by the library */
  int rowmin, nrows, wlo_off;
  SaTaps taps;
  /* fused backward epilogue (dgrad launches): ep_mode 0 = off; 1 = g' = (acc + ep_g2) * swish'(z),
   * z = ep_x*ep_s1[b][c] + ep_t1[b][c]; 2 = g' = acc + ep_g2.  g' is what is stored in y, and stats
   * becomes (sum g', sum g'*xhat), xhat = (xv - ep_mean)*ep_rstd with xv = ep_x, or swish(z) when
   * ep_xp_is_act; ep_mean / ep_rstd are indexed [b*ep_bstride + c] (ep_bstride = COUT or 0). */
  int ep_mode, ep_xp_is_act, ep_bstride;
  const void* ep_x; const void* ep_g2;
  const float* ep_s1; const float* ep_t1; const float* ep_mean; const float* ep_rstd;
  /* optional second output: the transformed input rows P(x) = s2*act(s1*x+t1)+t2, rounded to bf16,
   * [B][Lin][CIN] -- what sa_wgrad multiplies with when SaWgradArgs.x_pre is set */
  void* a_out;
  /* normalisation-backward prologue (optional, SA_BF16X3 and SA_BF16 launches): x holds d z of the layer
   * above and nb_x its stored forward tensor y (same shape); the staged rows become
   * d y = nb_c1*dz + nb_c2*y + nb_c3 [zeroed where y <= 0 if nb_relu_mask], coefficients indexed
   * [b*nb_bstride + c] (nb_bstride = CIN or 0); s1..swish are ignored.  This is sa_ew_apply fused
   * into its consumer; with a_out the bf16 d y also feeds sa_wgrad (dy_pre).  nb_colsum
   * [B][ntiles][CIN]: per-tile column sums of d y (bias gradient of the layer below). */
  const void* nb_x; const float* nb_c1; const float* nb_c2; const float* nb_c3;
  int nb_bstride, nb_relu_mask; float* nb_colsum;
  /* optional (ep_mode with ep_g2): ep_g2 is d(BN output) of a BatchNorm over the activation
   * swish(z); the epilogue uses k1[c]*ep_g2 + k2[c]*swish(z) + k3[c] in its place */
  const float* ep_g2k1; const float* ep_g2k2; const float* ep_g2k3;
  /* optional, launches with s1/t1 + swish and no s2: per-tile (sum, sum of squares) of the
   * transformed input rows P(x), [B][ntiles][CIN][2] */
  float* pro_stats;
  /* SA_FP8: device scalar the e4m3 weight image was multiplied with (accumulators are divided by it) */
  const float* wscale;
  /* output rows per workgroup of THIS launch on the one-tile kernel: 0 = the per-shape policy, 64 or
   * 128 (size `stats` with sa_conv_gemm_ntiles_tm) */
  int tile_rows; int pad2_;
} SaConvArgs;

int sa_conv_gemm(int dtype, int cin, int cout, int sa, int u, const SaConvArgs* a, void* stream);
/* sizeof(SaConvArgs | SaWgradArgs | SaEwArgs | SaPackDesc | SaTaps | SaFinArgs | SaBiasMulti | SaWredMulti) for which = 0..7: lets a binding
 * verify its mirror of these records (the library reads every field) */
int sa_abi_sizeof(int which);
int sa_conv_gemm_ntiles(int cin, int cout, int u, int Lout);
int sa_conv_gemm_ntiles_tm(int tile_rows, int u, int Lout);       /* tiles per utterance at an explicit tile height */
int sa_conv_gemm_set_tile_rows(int rows);   /* tuning knob: 0 (default policy), 64 or 128 */
/* Geometry of a launch with this dtype code and shape: *ntiles = slabs per utterance of nb_colsum /
 * pro_stats ([B][*ntiles][cin] / [B][*ntiles][cin][2]), *nslabs = statistics slabs per utterance
 * (`stats` is [B][*nslabs][cout][2]); the reducers sum slabs in index order.  It depends on the
 * kernel choice below (the ping-pong kernel writes one slab per wave that shares a column block). */
int sa_conv_gemm_geometry(int dtype, int cin, int cout, int u, int Lout, int* ntiles, int* nslabs);
/* Kernel choice of sa_conv_gemm, process-wide.  2 (default): the weight-stationary kernel
 * (sa_conv_ws.hip) serves the bf16x3 128->128, 64->64 stride-1, 64->128 stride-2 and 128->64 / 64->32 transposed launches with >= 1536 tiles that it
 * covers (5 taps at unit spacing, 128 channels also 3 taps over 4 / 6 rows; no fused backward
 * epilogue, no normalisation-backward prologue), the one-tile-per-workgroup kernel everything else -- same geometry and, given the same
 * inputs, the same output bits.  0: one-tile kernel only.  1: the ping-pong kernel (sa_conv_pp.hip)
 * for the f32 / bf16x3 policies (opt-in, A/B timing).  sa_conv_pp_set_tile_rows: 0 (policy), 64, 128. */
int sa_conv_gemm_set_impl(int impl);
/* which kernel serves this launch under the current choice: 0 one-tile, 1 ping-pong, 2 weight-stationary
 * (profiling tools name the kernel they time with it) */
int sa_conv_gemm_route(int dtype, int cin, int cout, int sa, int u, const SaConvArgs* a);
int sa_conv_pp_set_tile_rows(int rows);
/* persistent kernels: cost of an un-overlapped iteration in QUARTER tiles (an overlapped one costs 4; default 9),
 * for the equal-cost tile ranges (tuning; 4..64; bit 16: timing A/B of the first-tile overlap) */
int sa_conv_ws_set_bcost(int tiles);
/* The eight XCDs of a chip do not run the persistent kernels at one speed (workgroup lifetimes differ by up to
 * 12 % by XCD, stable within a process, different from chip to chip).  sa_conv_ws_set_xcd_weights: relative speed per
 * XCD, one byte each (64 = nominal, >= 16); workgroup i (XCD i % 8) gets that share of the tile cost.  Results do
 * not depend on it.  sa_conv_ws_calibrate_read: entry / exit times (100 MHz ticks) of the up to 512 workgroups
 * of the most recent sa_conv_wsd launch, [512][2] -- synchronises; the host derives the weights from it
 * (ops.calibrate_xcd). */
int sa_conv_ws_set_xcd_weights(const unsigned char* w8);
int sa_conv_ws_calibrate_read(unsigned long long* life512x2);
int sa_conv_wsd_set_bcost(int tiles);

/* fp32 master weights -> fragment-major MFMA operand image (K = GEMM reduction channels,
 * N = produced channels; element W(t,k,n) = src[k*sk + n*sn + t*st]).
 * nn.Conv1d.weight [Cout][Cin][Kw]: forward sk=Kw, sn=Cin*Kw, st=1; dgrad sk=Cin*Kw, sn=Kw.
 * nn.ConvTranspose1d.weight [Cin][Cout][Kw]: forward sk=Cout*Kw, sn=Kw; dgrad sk=Kw, sn=Cout*Kw. */
int sa_pack_weights(int dtype, const float* src, void* dst, int ntaps, int K, int N, int sk, int sn,
                    int st, void* stream);

/* n images in one launch: descs is an array of n records in DEVICE memory (the record of image i
 * holds the arguments sa_pack_weights would take for it). */
typedef struct SaPackDesc {
  const float* src; void* dst;
  int dtype, ntaps, K, N, sk, sn, st, pad_;
  float* scale;          /* SA_FP8 images: device scalar, written by sa_pack_scales_multi, read by the packer */
} SaPackDesc;
int sa_pack_weights_multi(const SaPackDesc* descs, int n, int blocks_per_image, void* stream);
/* per-tensor power-of-two scale of every SA_FP8 image of the table: 2^floor(log2(448 / max|w|))
 * (one workgroup per image; run it before sa_pack_weights_multi) */
int sa_pack_scales_multi(const SaPackDesc* descs, int n, void* stream);

/* ---- weight gradients (sa_wgrad.hip) --------------------------------------------------
 * dW[t][ci][co] = sum_b sum_{m<Mrows} P(x)[b, m*SA+off[t], ci] * dy[b, m*U+ph[t], co];
 * grid nchunk*B: each workgroup covers `chunk` (multiple of 64) base rows, every tap and the
 * whole CIN x COUT block; slabs[b][chunk][kw][t][CIN][COUT] with
 * kw < sa_wgrad_kw(cin, cout); sa_wgrad_reduce sums the B*nchunk*kw slabs in a fixed order into
 * dst[ci*sk + co*sn + t*st]. */
typedef struct SaWgradArgs {
  const void* x;
  const void* dy;
  float* slabs;
  const float* s1; const float* t1; const float* s2; const float* t2; int swish;
  int B, Lin, Ldy, Mrows, chunk, nchunk;
  int ntaps; int off[SA_MAX_TAPS]; int ph[SA_MAX_TAPS];
  int x_pre;   /* x is SaConvArgs.a_out of the forward launch (bf16, already transformed; s1..swish
                * ignored).  SA_BF16X1F and SA_BF16. */
  int dy_pre;  /* dy is SaConvArgs.a_out of the data-gradient launch that consumed it (bf16 d y);
                * requires x_pre. */
} SaWgradArgs;

int sa_wgrad(int dtype, int cin, int cout, int sa, int u, const SaWgradArgs* a, void* stream);
int sa_wgrad_kw(int cin, int cout);
int sa_wgrad_reduce(const float* slabs, float* dst, int nslab, int ntaps, int cin, int cout, int sk,
                    int sn, int st, int accumulate, void* stream);
/* the reducers of up to SA_WRED_MAX weight gradients in one launch (the end of a backward stage): each record is
 * the argument list of sa_wgrad_reduce (vec is filled in by the library); same order, same bits */
#define SA_WRED_MAX 8
typedef struct SaWredDesc {
  const float* slabs; float* dst;
  int nslab, ntaps, cin, cout, sk, sn, st, accumulate, vec, pad_;
} SaWredDesc;
typedef struct SaWredMulti { int n, pad_; SaWredDesc d[SA_WRED_MAX]; } SaWredMulti;
int sa_wgrad_reduce_multi(const SaWredMulti* m, void* stream);

/* ---- single-channel ends (sa_small.hip): encoder.0 Conv1d(1,32,15,p7) / decoder.8
 * Conv1d(32,1,15,p7), models/ConvAutoEncoder.py:142,171 ------------------------------- */
/* stats [B][ntiles][32][2].  ep_x (optional, y's layout): backward epilogue -- y = conv * swish'(z),
 * z = ep_x*ep_s1[b][c]+ep_t1[b][c]; stats = (sum y, sum y*(ep_x-ep_mean[b][c])*ep_rstd[b][c]) */
int sa_conv1toC(int dtype, const float* x, const float* w, const float* bias, void* y, int B, int L,
                int flip, float* stats, const void* ep_x, const float* ep_s1, const float* ep_t1,
                const float* ep_mean, const float* ep_rstd, void* stream);
int sa_conv1toC_ntiles(int L);
int sa_convCto1(int dtype, const void* x, const float* w, const float* bias, float* y, int B, int L,
                const float* s1, const float* t1, int swish, int flip, void* stream);
int sa_wgrad1C(int dtype, const float* u, const void* v, float* slabs, int B, int L, int chunk,
               int flip, const float* s1, const float* t1, int swish, void* stream);
int sa_wgrad1C_nchunk(int L, int chunk);
int sa_sum_slabs(const float* slabs, float* dst, int nslab, int n, int accumulate, void* stream);

/* ---- normalisation / activation backward + statistics finalisers (sa_elementwise.hip):
 * nn.InstanceNorm1d(affine), nn.BatchNorm1d (train), x*sigmoid(x), GradReverse
 * (models/ConvAutoEncoder.py:12-28,33-44,119-120,146-169) ---------------------------- */
typedef struct SaEwArgs {
  const void* g; const void* g2; const void* x; void* out;
  const float* s1; const float* t1;
  const float* mean; const float* rstd;
  const float* c1; const float* c2; const float* c3;
  int actbwd, xp_is_act, relu_mask, bstride;
  float* stats;
  int B, L, ntiles;
} SaEwArgs;

int sa_ew_stats(int dtype, int C, const SaEwArgs* a, void* stream);
int sa_ew_apply(int dtype, int C, const SaEwArgs* a, void* stream);
int sa_ew_ntiles(int L);
int sa_act_stats(int dtype, int C, const void* x, const float* s1, const float* t1, int swish,
                 float* stats, int B, int L, void* stream);
int sa_sum_partials(const float* slabs, double* dst, int nbatch, int nslab, int n, void* stream);
int sa_sum_rows_d(const double* src, double* dst, int R, int n, void* stream);
int sa_fin_in_fwd(const double* sums, int B, int C, int n, const float* gamma, const float* beta,
                  float eps, float* mean, float* rstd, float* scale, float* shift, void* stream);
/* sums of the BatchNorm finalisers may be R partial rows ([R][groups][2], added in row order).
 * count_dev / n_dev (optional, device fp64 scalar): the element count is read ON THE DEVICE and the
 * host value is ignored -- SyncBatchNorm over ranks with ragged batches all-reduces the per-rank
 * counts beside the sums (torch.nn.SyncBatchNorm exchanges counts the same way) without a host
 * round trip. */
int sa_fin_bn_fwd(const double* sums, int R, int C, double count, const float* gamma, const float* beta,
                  float eps, float momentum, float* run_mean, float* run_var, float* mean,
                  float* rstd, float* scale, float* shift, const double* count_dev, void* stream);
int sa_fin_bn_eval(int C, const float* gamma, const float* beta, float eps, const float* run_mean,
                   const float* run_var, float* mean, float* rstd, float* scale, float* shift,
                   void* stream);
int sa_fin_norm_bwd(const double* sums, const double* lsums, int R, int groups, int C, double n,
                    const float* gamma, const float* mean, const float* rstd, float sign, float* c1,
                    float* c2, float* c3, float* dgamma, float* dbeta, const double* n_dev,
                    void* stream);
int sa_fin_bias(const double* sums, int B, int C, int ncomp, float* db, void* stream);  /* sums [B][C][ncomp] */

/* Slab reduction + the finaliser that consumes it in one launch (replaces sa_sum_partials ->
 * [sa_sum_rows_d ->] sa_fin_* chains; same fp64 sums in the same order, bit-identical results).
 * part [nbatch][nslab][n] fp32 slabs, n = C * ncomp; rows [nbatch][n] fp64 (written; required except
 * for SA_FIN_IN_FWD); tickets: >= ceil(n/32) zero-initialised uint32 (self-resetting; one buffer may
 * serve consecutive launches on one stream).  Modes:
 *   SA_FIN_IN_FWD  per (b,c): o0..o3 = mean, rstd, scale, shift [nbatch][C]          (count = L)
 *   SA_FIN_IN_BWD  per (b,c): o0..o2 = c1, c2, c3 [nbatch][C]; dgamma/dbeta [C] = sums over b
 *   SA_FIN_BN_FWD  sums over the nbatch rows: o0..o3 [C], running statistics        (count = B*L)
 *   SA_FIN_BN_BWD  sums over the rows: o0..o2 = c1, c2, c3 [C], dgamma/dbeta [C]
 *   SA_FIN_BIAS    db[c] = sum over rows of component 0 (ncomp 1 or 2)
 * Single-process statistics only: when sums are all-reduced across ranks (SyncBatchNorm) use the
 * separate launches. */
#define SA_FIN_IN_FWD 1
#define SA_FIN_IN_BWD 2
#define SA_FIN_BN_FWD 3
#define SA_FIN_BN_BWD 4
#define SA_FIN_BIAS 5
typedef struct SaFinArgs {
  const float* part; double* rows; unsigned int* tickets;
  int nbatch, nslab, n, C, ncomp, mode;
  double count;
  float eps, momentum, sign, pad_;
  const float* gamma; const float* beta; const float* mean; const float* rstd;
  float* o0; float* o1; float* o2; float* o3;
  float* dgamma; float* dbeta; float* db; float* run_mean; float* run_var;
} SaFinArgs;
int sa_reduce_finalize(const SaFinArgs* a, void* stream);
/* Bias gradients of several layers at once (the end of a backward stage): for each of the n <= SA_BIAS_MAX
 * records, db[c] = sum over the nbatch x nslab slabs of part[b][s][c][0] (part [nbatch][nslab][C][ncomp] fp32,
 * the per-tile statistics or column-sum slabs of a data-gradient launch) -- what sa_sum_partials + sa_fin_bias do
 * for one layer, same summation order and bits, as TWO launches for all records (rows: fp64 scratch
 * [nbatch][C] per record). */
/* clip_grad_norm_(params, max_norm) of speechbrain's check_gradients (speechbrain_convae_train.py:249) on up to
 * SA_FLATS_MAX flat fp32 gradient buffers (the stage buckets of a backward): total = sqrt(sum g^2),
 * coef = min(1, max_norm / (total + eps)), g *= coef; partials: fp64 scratch [SA_FLATS_MAX * 64]; total_norm
 * (optional) receives the norm.  Two launches. */
#define SA_FLATS_MAX 4
typedef struct SaFlat { float* p; long long n; } SaFlat;
typedef struct SaFlats { int n, pad_; SaFlat f[SA_FLATS_MAX]; } SaFlats;
int sa_clip_grads(const SaFlats* f, float max_norm, float eps, double* partials, float* total_norm, void* stream);
#define SA_BIAS_MAX 8
typedef struct SaBiasDesc {
  const float* part; double* rows; float* db;
  int nbatch, nslab, C, ncomp;
} SaBiasDesc;
typedef struct SaBiasMulti { int n, pad_; SaBiasDesc d[SA_BIAS_MAX]; } SaBiasMulti;
int sa_bias_multi(const SaBiasMulti* m, void* stream);

/* ---- classifier head + losses (sa_head.hip): TDNNSexClassifier.forward reshape + pooling
 * (models/ConvAutoEncoder.py:61-66), classify (:47-55), log_softmax (:68); losses at
 * speechbrain_convae_train.py:105-108; utils/cosine_similarity_loss.py:53-56 ---------- */
/* two stages: part [B][nseg][128 channels][128 row residues][2] (nseg = sa_pool_nseg(B)), then
 * sa_pool_gather rotates / sums it into sums [B][128 pooled columns][2] (fp64) */
int sa_pool_fwd(int dtype, const void* r, const float* scale, const float* shift, float* part, int B,
                int L, int nseg, void* stream);
int sa_pool_nseg(int B);
int sa_pool_gather(const float* part, int B, int nseg, int L, double* sums, void* stream);
int sa_pool_fin(const double* sums, int B, int n, const float* noise, float eps, float* pooled,
                float* mean, float* stdraw, void* stream);
/* stats (optional, [B][ceil(L/256)][128][2]): partial (sum g, sum g*(r-bn_mean[c])*bn_rstd[c]) of
 * the written gradient, i.e. what sa_ew_stats would compute from g and r in a second pass */
int sa_pool_bwd(int dtype, const void* r, const float* scale, const float* shift,
                const float* dpooled, const float* mean, const float* stdraw, void* g, int B, int L,
                const float* bn_mean, const float* bn_rstd, float* stats, void* stream);
int sa_dense(const float* X, int lda, const float* ps, const float* pt, const float* W, int sbk,
             int sbn, const float* bias, float* Y, int ldy, int M, int N, int K, int relu,
             void* stream);
/* out0 / out1 (either may be NULL): sums[n][0] / sums[n][1] rounded to fp32, written straight into a
 * parameter-gradient tensor (the bias / BatchNorm affine gradients of the FC head) */
int sa_colsums(const float* X, const float* H, const float* hmean, const float* hrstd, int M, int N,
               double* sums, float* out0, float* out1, void* stream);
int sa_bn2d_bwd(const float* G, const float* H, const double* sums, double count, const float* gamma,
                const float* mean, const float* rstd, int relu_mask, int M, int N, float* dH,
                const double* count_dev, void* stream);
int sa_dense_wgrad(const float* dY, const float* X, const float* ps, const float* pt, int M, int N,
                   int K, float* dW, void* stream);
int sa_log_softmax(const float* X, float* Y, int M, int N, void* stream);
int sa_log_softmax_bwd(const float* dY, const float* Y, float* dX, int M, int N, void* stream);
int sa_loss_workspace_bytes(void);
int sa_recon_loss(const float* a, const float* b, long long n, int kind, float* grad, float* loss,
                  void* workspace, void* stream);                /* kind 0 = L1, 1 = MSE */
int sa_cls_losses(const float* logp, const long long* label, int B, int NC, float* out, float* dnll,
                  float* dconf, void* stream);                   /* out = (nll, confusion) */
int sa_cosine_loss(const float* x1, const float* x2, int B, int S, int D, float* rowloss,
                   float* loss, float* dx1, void* stream);

/* The FC head (classify: Linear(256,128) ReLU BatchNorm Linear(128,64) ReLU BatchNorm Linear(64,2),
 * models/ConvAutoEncoder.py:47-55, + log_softmax :68) as ONE forward and ONE backward launch
 * (sa_head_fused.hip): train mode, local BatchNorm statistics, M <= sa_head_max_rows() rows.
 * sa_head_fwd: pooled [M][256]; w / b: nn.Linear weight / bias; g / be: BatchNorm weight / bias;
 *   rm / rv: running statistics (updated; may be NULL); outputs h1 [M][128], h2 [M][64] (post-ReLU),
 *   f1 [4][128], f2 [4][64] (mean, rstd, scale, shift), logp [M][2].
 * sa_head_bwd: dlogp [M][2] -> every parameter gradient (any may be NULL) and dpooled [M][256].
 * Larger batches, eval mode and SyncBatchNorm use the separate launches above. */
int sa_head_fwd(const float* pooled, const float* w1, const float* b1, const float* g1, const float* be1,
                float* rm1, float* rv1, const float* w2, const float* b2, const float* g2, const float* be2,
                float* rm2, float* rv2, const float* w3, const float* b3, float* h1, float* f1, float* h2,
                float* f2, float* logp, int M, float eps, float momentum, void* stream);
int sa_head_bwd(const float* dlogp, const float* logp, const float* pooled, const float* h1, const float* f1,
                const float* h2, const float* f2, const float* w1, const float* g1, const float* w2,
                const float* g2, const float* w3, float* dw1, float* db1, float* dg1, float* dbe1, float* dw2,
                float* db2, float* dg2, float* dbe2, float* dw3, float* db3, float* dpooled, int M, void* stream);
int sa_head_max_rows(void);

/* x-vector gender classifier forward (models/external_gender_classifiers.py:71-115,144-183;
 * evaluator_inference.yaml:34-48): TDNN block = speechbrain Conv1d (reflect "same" padding) ->
 * LeakyReLU -> BatchNorm1d(eval); StatisticsPooling over time with relative lengths. */
/* wp: sa_pack_weights(SA_BF16X3, ...) image of the Conv1d weight zero-padded to Npad output channels
 * (Npad % 128 == 0; ntaps = K, K = Cin (% 16 == 0), N = Npad, sk = K, sn = Cin*K, st = 1) */
/* mask (optional, [B][T][Cout] bytes): 1 where the LeakyReLU input was positive (for sa_tdnn_bwd_input) */
int sa_tdnn_fwd(const float* x, const void* wp, const float* bias, const float* bn_s, const float* bn_t,
                float* y, int B, int T, int Cin, int Cout, int Npad, int K, int dil, float slope,
                unsigned char* mask, void* stream);
int sa_time_pool(const float* x, const float* lens, const float* noise, int B, int T, int C, float eps,
                 float* out, void* stream);
int sa_leaky_affine(const float* x, const float* s, const float* t, float slope, int M, int C, float* y,
                    void* stream);
/* the same classifier INSIDE the training graph (models/EndToEnd.py:57-61,81: pretrained, frozen):
 * gradient with respect to the input features only.
 * sa_tdnn_bwd_input: dy [B][T][Cy], mask = the forward's LeakyReLU branch mask -> dxe [B][T + dil*(K-1)][Cin],
 *   the data gradient on the range extended by the "same" padding; wp = split-bf16 image of the
 *   Conv1d weight packed as a data-gradient operand (reduction Cred >= Cy, % 16; produced Npad >= Cin,
 *   % 128).  sa_tdnn_fold applies the adjoint of the reflect padding: dxe -> dx [B][T][C].
 * sa_time_pool_bwd: pooled = forward output without the noise offset.  */
int sa_tdnn_bwd_input(const float* dy, const unsigned char* mask, const float* bn_s, const void* wp,
                      float* dxe, int B, int T, int Cy, int Cred, int Cin, int Npad, int K, int dil,
                      float slope, void* stream);
int sa_tdnn_fold(const float* dxe, float* dx, int B, int T, int C, int pad, void* stream);
int sa_time_pool_bwd(const float* x, const float* lens, const float* g, const float* pooled, int B,
                     int T, int C, float eps, float* dx, void* stream);
int sa_leaky_affine_bwd(const float* dy, const float* x, const float* s, float slope, int M, int C,
                        float* dx, void* stream);

/* ---- k-NN mutual information (sa_mi.hip): utils/ClusterMI.py:88-121,
 * utils/GroupSamplingMI.py:49-61, utils/mi_loss.py:14-17 ------------------------------ */
int sa_cluster_mi(const float* X, const long long* y, const long long* idx, int iters, int n, int D,
                  int ncls, int k, float* mi, void* stream);

/* ---- feature front-end (sa_fbank.hip): speechbrain Fbank + InputNormalization at
 * speechbrain_convae_train.py:58-63,82-87 (convae.yaml:93-95,269-271,289-292) --------- */
/* dft / mel: bf16 fragment-major operand images built once by the host (layout in sa_fbank.hip;
 * speech-anonymization_amd/features.py builds them), sa_fbank_table_elems(0 | 1) elements each */
int sa_fbank(const float* wav, int B, int N, const float* window, const void* dft,
             const void* mel, float* feats, float* tilemax, void* stream);
int sa_fbank_table_elems(int which);
int sa_fbank_ntiles(int T);
int sa_fbank_scratch_bytes(int B);
int sa_fbank_normalize(const float* feats, const float* tilemax, int B, int T, int Tp,
                       const float* lens, float top_db, int batch_max, int update, int epoch,
                       int update_until_epoch, float* state, float* scratch, float* out,
                       void* stream);

/* ---- element-wise passes of the frozen recogniser (sa_asr.hip; SURVEY 8f-2, models/SpeechBrain_ASR.py:16-30;
 * bf16 storage, fp32 arithmetic; the GEMMs around them are library calls).
 *   sa_add_layernorm_fwd: s = bf16(x + r) (r may be NULL), y = LayerNorm_d(s) * gamma + beta over rows of d
 *     elements (d in {256, 512, 768, 1024}); s_out (the tensor the backward re-reads) and stat [rows][2] =
 *     (mean, rstd) are optional: NULL under no_grad.  speechbrain's post-norm layers, x = norm(x + f(x)).
 *   sa_layernorm_bwd: d s from d y, s, stat, gamma -- the gradient of both addends of the forward.
 *   sa_reflect_pad_fwd / _bwd: F.pad(., (1, 1, 1, 1), "reflect") on the (T, F) axes of [B][T][F][C] rows
 *     (the "same" padding of ConvolutionFrontEnd's 3 x 3 convolutions) and its adjoint; T, F >= 3. */
int sa_add_layernorm_fwd(const void* x, const void* r, const void* gamma, const void* beta, void* y, void* s_out,
                         float* stat, int rows, int d, float eps, void* stream);
int sa_layernorm_bwd(const void* dy, const void* s, const float* stat, const void* gamma, void* ds, int rows,
                     int d, void* stream);
/* sa_ln_leaky_fwd / _bwd: the front end's LayerNorm over (frequency, channel) + LeakyReLU in one pass each way;
 *   rows of d in {5120, 10240} elements (-ENOSYS otherwise); the backward recomputes the activation's branch
 *   from x and stat, so only x (the convolution's output) and [rows][2] statistics are kept. */
int sa_ln_leaky_fwd(const void* x, const void* gamma, const void* beta, void* y, float* stat, int rows, int d,
                    float eps, float slope, void* stream);
int sa_ln_leaky_bwd(const void* dy, const void* x, const float* stat, const void* gamma, const void* beta, void* dx,
                    int rows, int d, float slope, void* stream);
/* sa_asr_block0_fwd / _bwd: block 0 of the front end -- Conv2d(1 -> C = 128, 3 x 3, stride 2, reflect "same"
 *   padding) + LayerNorm over (F' = 40, C) + LeakyReLU on x [B][T][F = 80] -> y [B][ceil(T/2)][40][128] -- as one
 *   pass each way (-ENOSYS for other F / C).  w [C][3][3], bias [C], gamma / beta [40][128], all bf16.
 *   stat [B * ceil(T/2)][2] (mean, rstd; NULL under no_grad) is all the backward keeps besides x: it recomputes
 *   the convolution.  part: fp32 scratch [B * ceil(T/2)][3][F + 2]; dx [B][T][F] bf16. */
int sa_asr_block0_fwd(const void* x, const void* w, const void* bias, const void* gamma, const void* beta, void* y,
                      float* stat, int B, int T, int F, int C, float eps, float slope, void* stream);
int sa_asr_block0_bwd(const void* dy, const void* x, const void* w, const void* bias, const void* gamma,
                      const void* beta, const float* stat, float* part, void* dx, int B, int T, int F, int C,
                      float slope, void* stream);
int sa_reflect_pad_fwd(const void* x, void* y, int B, int T, int F, int C, void* stream);
int sa_reflect_pad_bwd(const void* dy, void* dx, int B, int T, int F, int C, void* stream);

/* ---- data-parallel exchange (sa_comm.hip): what DistributedDataParallel / SyncBatchNorm do for
 * the reference once speechbrain_convae_train.py:524 (ddp_init_group) has run -- the gradient
 * average and the BatchNorm statistic sums -- as in-place RCCL all-reduces on a side stream the
 * library owns.  One process per GPU; rank 0 calls sa_comm_unique_id and hands the 128 bytes to
 * the other ranks by any host channel (the Python side uses the torch.distributed store), then
 * every rank calls sa_comm_init (collective: returns when all `world` ranks have joined).
 *   sa_comm_allreduce: the side stream waits for everything enqueued so far on `producer_stream`,
 *     then reduces buf[n] in place (dtype SA_F32 | SA_F64; avg != 0: ncclAvg, else sum).  Returns
 *     at once.  sa_comm_allreduce_inline: the same collective enqueued in `stream` itself (no side
 *     stream, no events: producer and consumer are that stream's neighbours).  sa_comm_join: `consumer_stream` waits for every all-reduce enqueued so far.
 *   Codes: -ENOSYS no RCCL library in the process or on the loader path (it is bound by dlopen
 *     at the first sa_comm_* call, never at load time), -ENOTCONN before sa_comm_init, -EEXIST
 *     second sa_comm_init, -(1000 + ncclResult_t) from RCCL, -(hipError_t) from HIP.
 *   sa_comm_world: 0 before init.  sa_comm_ncalls: all-reduces enqueued since init (tests). */
int sa_comm_unique_id(void* id128);
int sa_comm_init(int rank, int world, const void* id128, int device);
int sa_comm_world(void);
int sa_comm_allreduce(void* buf, long long n, int dtype, int avg, void* producer_stream);
int sa_comm_allreduce_inline(void* buf, long long n, int dtype, int avg, void* stream);
int sa_comm_join(void* consumer_stream);
int sa_comm_ncalls(void);
int sa_comm_destroy(void);

#ifdef __cplusplus
}
#endif
#endif /* SA_HIP_H */
