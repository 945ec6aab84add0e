// Normalisation / activation backward passes and the statistics finalisers (HBM-bound).
//
// Forward normalisation is never a pass of its own: InstanceNorm1d / BatchNorm1d statistics
// come out of the producing conv's epilogue (sa_conv_gemm) as per-tile partial sums and the
// affine + x*sigmoid(x) is applied in the consuming kernel's prologue.  Backward needs two
// phases per normalised tensor (the gradient of a mean couples every row):
//   sa_ew_stats : g' = (g [+ g2]) * swish'(z)           -> sum g', sum g'*xhat per (b, c)
//   sa_fin_*    : tiny kernels turning sums into coefficients (and d gamma / d beta)
//   sa_ew_apply : d y = (c1*g' + c2*x + c3) [* (x > 0)]  -> + per-(b,c) sum for bias grads
// Reference semantics: nn.InstanceNorm1d(affine, eps 1e-5, biased var), nn.BatchNorm1d in
// train mode, x*sigmoid(x) (models/ConvAutoEncoder.py:119-120,146-169,33-43), GradReverse
// (:12-28) folded into the coefficients as a sign.
#include "sa_common.h"


#define SA_EW_ROWS 256

template <typename T, int C, bool APPLY>
__global__ __launch_bounds__(256) void sa_ew_kernel(SaEwArgs a) {
  typedef Tr<T> tr;
  constexpr int VEC = tr::VEC, CH = C / VEC, RPP = 256 / CH;
  __shared__ float red[RPP][C][2];
  const int tid = threadIdx.x, b = blockIdx.y, tile = blockIdx.x;
  const int c = tid % CH, r0 = tid / CH;
  const size_t cb1 = (size_t)b * C + c * VEC;                   // (b,c) indexed arrays (s1,t1)
  const size_t cbn = (size_t)b * a.bstride + c * VEC;           // mean/rstd/c1..c3
  float s1[VEC], t1[VEC], mu[VEC], rs[VEC], k1[VEC], k2[VEC], k3[VEC], acc0[VEC], acc1[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    s1[j] = a.s1 ? a.s1[cb1 + j] : 1.0f;
    t1[j] = a.t1 ? a.t1[cb1 + j] : 0.0f;
    mu[j] = a.mean ? a.mean[cbn + j] : 0.0f;
    rs[j] = a.rstd ? a.rstd[cbn + j] : 1.0f;
    k1[j] = (APPLY && a.c1) ? a.c1[cbn + j] : 1.0f;
    k2[j] = (APPLY && a.c2) ? a.c2[cbn + j] : 0.0f;
    k3[j] = (APPLY && a.c3) ? a.c3[cbn + j] : 0.0f;
    acc0[j] = 0.0f; acc1[j] = 0.0f;
  }
  const size_t base = (size_t)b * a.L * C + c * VEC;
  const T* gp = reinterpret_cast<const T*>(a.g) + base;
  const T* g2p = a.g2 ? reinterpret_cast<const T*>(a.g2) + base : nullptr;
  const T* xp = reinterpret_cast<const T*>(a.x) + base;
  T* op = a.out ? reinterpret_cast<T*>(a.out) + base : nullptr;
  const int lbeg = tile * SA_EW_ROWS;
  for (int r = r0; r < SA_EW_ROWS; r += RPP) {
    const int l = lbeg + r;
    if (l >= a.L) break;
    float g[VEC], x[VEC], o[VEC];
    tr::unpack(*reinterpret_cast<const uint4*>(gp + (size_t)l * C), g);
    tr::unpack(*reinterpret_cast<const uint4*>(xp + (size_t)l * C), x);
    if (g2p) {
      float g2[VEC];
      tr::unpack(*reinterpret_cast<const uint4*>(g2p + (size_t)l * C), g2);
#pragma unroll
      for (int j = 0; j < VEC; ++j) g[j] += g2[j];
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float z = fmaf(x[j], s1[j], t1[j]);
      const float xv = a.xp_is_act ? sa_swish(z) : x[j];
      if (APPLY) {
        float v = fmaf(k1[j], g[j], fmaf(k2[j], xv, k3[j]));
        if (a.relu_mask && !(x[j] > 0.0f)) v = 0.0f;
        o[j] = v;
      } else {
        o[j] = a.actbwd ? g[j] * sa_swish_grad(z) : g[j];
      }
    }
    uint4 u = tr::pack(o);
    if (op) *reinterpret_cast<uint4*>(op + (size_t)l * C) = u;
    if (a.stats) {
      tr::unpack(u, o);                                  // statistics of the STORED values
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        acc0[j] += o[j];
        if (!APPLY) {
          const float z = fmaf(x[j], s1[j], t1[j]);
          const float xv = a.xp_is_act ? sa_swish(z) : x[j];
          acc1[j] = fmaf(o[j], (xv - mu[j]) * rs[j], acc1[j]);
        }
      }
    }
  }
  if (a.stats) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) { red[r0][c * VEC + j][0] = acc0[j]; red[r0][c * VEC + j][1] = acc1[j]; }
    __syncthreads();
    if (tid < C) {
      float s = 0.f, q = 0.f;
      for (int r = 0; r < RPP; ++r) { s += red[r][tid][0]; q += red[r][tid][1]; }
      float* d = a.stats + (((size_t)b * a.ntiles + tile) * C + tid) * 2;
      d[0] = s; d[1] = q;
    }
  }
}

extern "C" int sa_ew_ntiles(int L) { return sa_div_up(L, SA_EW_ROWS); }

template <bool APPLY>
static int launch_ew(int dtype, int C, const SaEwArgs& a0, hipStream_t st) {
  SaEwArgs a = a0;
  a.ntiles = sa_div_up(a.L, SA_EW_ROWS);
  dim3 grid(a.ntiles, a.B);
#define SA_EW_CASE(CC)                                                                      \
  if (C == CC) {                                                                            \
    if (dtype == SA_BF16) hipLaunchKernelGGL((sa_ew_kernel<bf16_t, CC, APPLY>), grid, dim3(256), 0, st, a); \
    else hipLaunchKernelGGL((sa_ew_kernel<float, CC, APPLY>), grid, dim3(256), 0, st, a);   \
    hipError_t e = hipGetLastError();                                                       \
    return e == hipSuccess ? 0 : -(int)e;                                                   \
  }
  SA_EW_CASE(32) SA_EW_CASE(64) SA_EW_CASE(128)
  return -38;
}

extern "C" int sa_ew_stats(int dtype, int C, const SaEwArgs* a, void* stream) {
  if (!a || !a->g || !a->x || a->B <= 0 || a->L <= 0) return -22;
  return launch_ew<false>(dtype, C, *a, reinterpret_cast<hipStream_t>(stream));
}
extern "C" int sa_ew_apply(int dtype, int C, const SaEwArgs* a, void* stream) {
  if (!a || !a->g || !a->x || !a->out || a->B <= 0 || a->L <= 0) return -22;
  return launch_ew<true>(dtype, C, *a, reinterpret_cast<hipStream_t>(stream));
}

// ---------------------------------------------------------------------------------
// transform_stats: per-(b,c) sum / sumsq of swish(x*s1+t1) -- statistics of the encoder
// output for the classifier's first BatchNorm (`norm`, ConvAutoEncoder.py:44,59).
// ---------------------------------------------------------------------------------
template <typename T, int C>
__global__ __launch_bounds__(256) void sa_act_stats_kernel(const T* __restrict__ x,
                                                           const float* __restrict__ s1p,
                                                           const float* __restrict__ t1p,
                                                           int swish, float* __restrict__ stats,
                                                           int L, int ntiles) {
  typedef Tr<T> tr;
  constexpr int VEC = tr::VEC, CH = C / VEC, RPP = 256 / CH;
  __shared__ float red[RPP][C][2];
  const int tid = threadIdx.x, b = blockIdx.y, tile = blockIdx.x;
  const int c = tid % CH, r0 = tid / CH;
  float s1[VEC], t1[VEC], a0[VEC], a1[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    s1[j] = s1p ? s1p[(size_t)b * C + c * VEC + j] : 1.0f;
    t1[j] = t1p ? t1p[(size_t)b * C + c * VEC + j] : 0.0f;
    a0[j] = 0.f; a1[j] = 0.f;
  }
  const T* xp = x + (size_t)b * L * C + c * VEC;
  for (int r = r0; r < SA_EW_ROWS; r += RPP) {
    const int l = tile * SA_EW_ROWS + r;
    if (l >= L) break;
    float f[VEC];
    tr::unpack(*reinterpret_cast<const uint4*>(xp + (size_t)l * C), f);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float v = fmaf(f[j], s1[j], t1[j]);
      if (swish) v = sa_swish(v);
      a0[j] += v; a1[j] = fmaf(v, v, a1[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) { red[r0][c * VEC + j][0] = a0[j]; red[r0][c * VEC + j][1] = a1[j]; }
  __syncthreads();
  if (tid < C) {
    float s = 0.f, q = 0.f;
    for (int r = 0; r < RPP; ++r) { s += red[r][tid][0]; q += red[r][tid][1]; }
    float* d = stats + (((size_t)b * ntiles + tile) * C + tid) * 2;
    d[0] = s; d[1] = q;
  }
}

extern "C" int sa_act_stats(int dtype, int C, const void* x, const float* s1, const float* t1,
                            int swish, float* stats, int B, int L, void* stream) {
  if (!x || !stats || B <= 0 || L <= 0) return -22;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int nt = sa_div_up(L, SA_EW_ROWS);
  dim3 grid(nt, B);
  if (C == 128 && dtype == SA_BF16)
    hipLaunchKernelGGL((sa_act_stats_kernel<bf16_t, 128>), grid, dim3(256), 0, st,
                       reinterpret_cast<const bf16_t*>(x), s1, t1, swish, stats, L, nt);
  else if (C == 128)
    hipLaunchKernelGGL((sa_act_stats_kernel<float, 128>), grid, dim3(256), 0, st,
                       reinterpret_cast<const float*>(x), s1, t1, swish, stats, L, nt);
  else
    return -38;
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// ---------------------------------------------------------------------------------
// batched slab sum: dst[bb][i] = sum_k slabs[bb][k][i]   (fixed order, double accumulate)
// ---------------------------------------------------------------------------------
// 512 threads = 32 outputs x 16 slab lanes (128-byte coalesced segments): lane q sums slabs q,
// q+16, ... in order (16 loads in flight per thread: the launch is latency-bound, one workgroup
// per CU), then the 16 lane sums are added in lane order by one thread -> the same result on
// every run.
#define SA_SP_LANES 16
__global__ __launch_bounds__(32 * SA_SP_LANES) void sa_sum_partials_kernel(const float* __restrict__ slabs,
                                                                           double* __restrict__ dst,
                                                                           int nslab, int n) {
  __shared__ double part[SA_SP_LANES][33];
  const int bb = blockIdx.y, o = threadIdx.x & 31, q = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + o;
  double s = 0.0;
  if (i < n) {
    const float* p = slabs + (size_t)bb * nslab * n + i;
#pragma unroll 16
    for (int k = q; k < nslab; k += SA_SP_LANES) s += (double)p[(size_t)k * n];
  }
  part[q][o] = s;
  __syncthreads();
  if (q == 0 && i < n) {
    double t = 0.0;
#pragma unroll
    for (int r = 0; r < SA_SP_LANES; ++r) t += part[r][o];
    dst[(size_t)bb * n + i] = t;
  }
}

extern "C" int sa_sum_partials(const float* slabs, double* dst, int nbatch, int nslab, int n,
                               void* stream) {
  if (!slabs || !dst || nbatch <= 0 || nslab <= 0 || n <= 0) return -22;
  hipLaunchKernelGGL(sa_sum_partials_kernel, dim3(sa_div_up(n, 32), nbatch), dim3(32 * SA_SP_LANES), 0,
                     reinterpret_cast<hipStream_t>(stream), slabs, dst, nslab, n);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// dst[i] = sum_r src[r][i]  (fp64 in / out; second level of a two-level slab reduction)
__global__ void sa_sum_rows_d_kernel(const double* __restrict__ src, double* __restrict__ dst, int R, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = 0.0;
#pragma unroll 8
  for (int r = 0; r < R; ++r) s += src[(size_t)r * n + i];
  dst[i] = s;
}

extern "C" int sa_sum_rows_d(const double* src, double* dst, int R, int n, void* stream) {
  if (!src || !dst || R <= 0 || n <= 0) return -22;
  hipLaunchKernelGGL(sa_sum_rows_d_kernel, dim3(sa_div_up(n, 128)), dim3(128), 0,
                     reinterpret_cast<hipStream_t>(stream), src, dst, R, n);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// ---------------------------------------------------------------------------------
// finalisers.  sums layouts: IN  [B][C][2],  BN  [C][2]  (sum, sumsq) or (S1, S2).
// ---------------------------------------------------------------------------------
// InstanceNorm forward: mean, rstd, scale = gamma*rstd, shift = beta - mean*scale per (b,c)
__global__ void sa_fin_in_fwd_kernel(const double* __restrict__ sums, int BC, int C, float n,
                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                     float eps, float* mean, float* rstd, float* scale, float* shift) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= BC) return;
  const double m = (double)sums[2 * i] / n;
  double var = (double)sums[2 * i + 1] / n - m * m;
  if (var < 0.0) var = 0.0;
  const float r = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = gamma[i % C] * r;
  mean[i] = (float)m; rstd[i] = r; scale[i] = sc; shift[i] = beta[i % C] - (float)m * sc;
}

extern "C" int sa_fin_in_fwd(const double* sums, int B, int C, int n, const float* gamma,
                             const float* beta, float eps, float* mean, float* rstd, float* scale,
                             float* shift, void* stream) {
  if (!sums || !gamma || !beta || !mean || !rstd || !scale || !shift) return -22;
  hipLaunchKernelGGL(sa_fin_in_fwd_kernel, dim3(sa_div_up(B * C, 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), sums, B * C, C, (float)n, gamma, beta,
                     eps, mean, rstd, scale, shift);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// BatchNorm (train mode) forward from (possibly all-reduced) sums over `count` elements per
// channel; updates running_mean / running_var (unbiased) with `momentum` like nn.BatchNorm1d.
// sums may come as R partial rows [R][C][2] (the per-utterance level of the slab reduction): they
// are added here in row order, which saves a launch per BatchNorm
// count_dev != null: the element count is read on the device (the all-reduced count of a
// SyncBatchNorm whose ranks hold ragged batches) and `count` is ignored.
__global__ void sa_fin_bn_fwd_kernel(const double* __restrict__ sums, int R, int C, double count,
                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                     float eps, float momentum, float* run_mean, float* run_var,
                                     float* mean, float* rstd, float* scale, float* shift,
                                     const double* __restrict__ count_dev) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C) return;
  if (count_dev) count = *count_dev;
  double S = 0.0, Q = 0.0;
#pragma unroll 8
  for (int r = 0; r < R; ++r) { S += sums[2 * ((size_t)r * C + i)]; Q += sums[2 * ((size_t)r * C + i) + 1]; }
  const double m = S / count;
  double var = Q / count - m * m;
  if (var < 0.0) var = 0.0;
  const float r = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = gamma[i] * r;
  mean[i] = (float)m; rstd[i] = r; scale[i] = sc; shift[i] = beta[i] - (float)m * sc;
  if (run_mean) {
    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    run_mean[i] = (1.0f - momentum) * run_mean[i] + momentum * (float)m;
    run_var[i] = (1.0f - momentum) * run_var[i] + momentum * (float)unb;
  }
}

extern "C" int sa_fin_bn_fwd(const double* sums, int R, int C, double count, const float* gamma,
                             const float* beta, float eps, float momentum, float* run_mean,
                             float* run_var, float* mean, float* rstd, float* scale, float* shift,
                             const double* count_dev, void* stream) {
  if (!sums || !gamma || !beta || !mean || !rstd || !scale || !shift || (!count_dev && count <= 0) || R < 1) return -22;
  hipLaunchKernelGGL(sa_fin_bn_fwd_kernel, dim3(sa_div_up(C, 128)), dim3(128), 0,
                     reinterpret_cast<hipStream_t>(stream), sums, R, C, count, gamma, beta, eps,
                     momentum, run_mean, run_var, mean, rstd, scale, shift, count_dev);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// BatchNorm in eval mode: normalise with the running statistics.
__global__ void sa_fin_bn_eval_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                      float eps, const float* __restrict__ rm, const float* __restrict__ rv,
                                      float* mean, float* rstd, float* scale, float* shift) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C) return;
  const float r = 1.0f / sqrtf(rv[i] + eps), sc = gamma[i] * r;
  mean[i] = rm[i]; rstd[i] = r; scale[i] = sc; shift[i] = beta[i] - rm[i] * sc;
}

extern "C" int sa_fin_bn_eval(int C, const float* gamma, const float* beta, float eps,
                              const float* run_mean, const float* run_var, float* mean, float* rstd,
                              float* scale, float* shift, void* stream) {
  if (!gamma || !beta || !run_mean || !run_var || !mean || !rstd || !scale || !shift) return -22;
  hipLaunchKernelGGL(sa_fin_bn_eval_kernel, dim3(sa_div_up(C, 128)), dim3(128), 0,
                     reinterpret_cast<hipStream_t>(stream), C, gamma, beta, eps, run_mean, run_var,
                     mean, rstd, scale, shift);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// Normalisation backward coefficients.  sums = (S1 = sum g', S2 = sum g'*xhat) per group;
// groups = B*C (InstanceNorm, n = L) or C (BatchNorm, n = count).
//   d x = c1*g' + c2*xv + c3,  c1 = gamma*rstd, c2 = -c1*rstd*S2/n, c3 = c1*(-S1/n + mean*rstd*S2/n)
// sign = -1 folds the GradReverse layer in.  lsums (local sums, may equal sums) feed
// d gamma = sum_b S2, d beta = sum_b S1 (written, not accumulated, when dgamma != null).
__global__ void sa_fin_norm_bwd_kernel(const double* __restrict__ sums, const double* __restrict__ lsums,
                                       int R, int G, int C, double n, const float* __restrict__ gamma,
                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                       float sign, float* c1, float* c2, float* c3, float* dgamma,
                                       float* dbeta, int nb, int coef_blocks,
                                       const double* __restrict__ n_dev) {
  if (n_dev) n = *n_dev;                                // device-side (all-reduced) element count
  if ((int)blockIdx.x < coef_blocks) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < G) {
      double S1 = 0.0, S2 = 0.0;                        // R partial rows [R][G][2], added in row order
#pragma unroll 8
      for (int r = 0; r < R; ++r) { S1 += sums[2 * ((size_t)r * G + i)]; S2 += sums[2 * ((size_t)r * G + i) + 1]; }
      const double r = rstd[i], m = mean[i];
      const double k1 = (double)gamma[i % C] * r;
      c1[i] = sign * (float)k1;
      c2[i] = sign * (float)(-k1 * r * S2 / n);
      c3[i] = sign * (float)(k1 * (-S1 / n + m * r * S2 / n));
    }
    return;
  }
  // d gamma / d beta: 32 channels x 8 lanes per workgroup; lane q adds the (row, utterance) pairs
  // q, q+8, ... in order, the 8 lane sums are added in lane order
  __shared__ double part[8][32][2];
  const int ch = ((int)blockIdx.x - coef_blocks) * 32 + (threadIdx.x & 31), q = threadIdx.x >> 5;
  double a = 0.0, bsum = 0.0;
  if (ch < C) {
    const int tot = R * nb;
#pragma unroll 4
    for (int e = q; e < tot; e += 8) {
      const int r = e / nb, b = e % nb;
      const size_t at = 2 * ((size_t)r * G + (size_t)b * C + ch);
      bsum += lsums[at]; a += lsums[at + 1];
    }
  }
  part[q][threadIdx.x & 31][0] = bsum; part[q][threadIdx.x & 31][1] = a;
  __syncthreads();
  if (q == 0 && ch < C) {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { s0 += part[k][threadIdx.x][0]; s1 += part[k][threadIdx.x][1]; }
    dbeta[ch] = (float)s0; dgamma[ch] = (float)s1;
  }
}

extern "C" int sa_fin_norm_bwd(const double* sums, const double* lsums, int R, int groups, int C, double n,
                               const float* gamma, const float* mean, const float* rstd, float sign,
                               float* c1, float* c2, float* c3, float* dgamma, float* dbeta,
                               const double* n_dev, void* stream) {
  if (!sums || !gamma || !mean || !rstd || !c1 || !c2 || !c3 || groups % C || R < 1) return -22;
  if ((dgamma == nullptr) != (dbeta == nullptr)) return -22;
  const int coef_blocks = sa_div_up(groups, 256);
  const int grid = coef_blocks + (dgamma ? sa_div_up(C, 32) : 0);
  hipLaunchKernelGGL(sa_fin_norm_bwd_kernel, dim3(grid), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), sums, lsums ? lsums : sums, R, groups, C,
                     n, gamma, mean, rstd, sign, c1, c2, c3, dgamma, dbeta, groups / C, coef_blocks, n_dev);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// bias gradient from per-(b,c) sums: db[c] = sum_b sums[b][c][0]   (sums [B][C][ncomp])
__global__ void sa_fin_bias_kernel(const double* __restrict__ sums, int B, int C, int ncomp, float* db) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C) return;
  double s = 0.0;
#pragma unroll 8
  for (int b = 0; b < B; ++b) s += sums[(size_t)ncomp * ((size_t)b * C + i)];
  db[i] = (float)s;
}

extern "C" int sa_fin_bias(const double* sums, int B, int C, int ncomp, float* db, void* stream) {
  if (!sums || !db || ncomp < 1) return -22;
  hipLaunchKernelGGL(sa_fin_bias_kernel, dim3(sa_div_up(C, 128)), dim3(128), 0,
                     reinterpret_cast<hipStream_t>(stream), sums, B, C, ncomp, db);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// ---------------------------------------------------------------------------------
// sa_bias_multi: the bias gradients of all layers of a backward stage in two launches (they are
// needed by nobody before the stage's gradients are reduced / the optimizer runs, so their
// ~2 x 12 launches per step leave the critical chain of the backward).  Level 1 = sa_sum_partials
// restricted to component 0 of every channel, level 2 = sa_fin_bias: same lanes, same order, same bits.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(32 * SA_SP_LANES) void sa_bias_multi_l1_kernel(SaBiasMulti m) {
  __shared__ double part[SA_SP_LANES][33];
  const SaBiasDesc& d = m.d[blockIdx.z];
  const int bb = blockIdx.y, o = threadIdx.x & 31, q = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + o;
  if (bb >= d.nbatch || blockIdx.x * 32 >= d.C) return;          // (uniform per workgroup)
  double s = 0.0;
  const int n = d.C * d.ncomp;
  if (i < d.C) {
    const float* p = d.part + (size_t)bb * d.nslab * n + (size_t)i * d.ncomp;
#pragma unroll 16
    for (int k = q; k < d.nslab; k += SA_SP_LANES) s += (double)p[(size_t)k * n];
  }
  part[q][o] = s;
  __syncthreads();
  if (q == 0 && i < d.C) {
    double t = 0.0;
#pragma unroll
    for (int r = 0; r < SA_SP_LANES; ++r) t += part[r][o];
    d.rows[(size_t)bb * d.C + i] = t;
  }
}

__global__ void sa_bias_multi_l2_kernel(SaBiasMulti m) {
  const SaBiasDesc& d = m.d[blockIdx.y];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.C) return;
  double s = 0.0;
#pragma unroll 8
  for (int b = 0; b < d.nbatch; ++b) s += d.rows[(size_t)b * d.C + i];
  d.db[i] = (float)s;
}

extern "C" int sa_bias_multi(const SaBiasMulti* m, void* stream) {
  if (!m || m->n <= 0 || m->n > SA_BIAS_MAX) return -22;
  int maxC = 0, maxB = 0;
  for (int j = 0; j < m->n; ++j) {
    const SaBiasDesc& d = m->d[j];
    if (!d.part || !d.rows || !d.db || d.nbatch <= 0 || d.nslab <= 0 || d.C <= 0 || d.ncomp < 1) return -22;
    maxC = d.C > maxC ? d.C : maxC;
    maxB = d.nbatch > maxB ? d.nbatch : maxB;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(sa_bias_multi_l1_kernel, dim3(sa_div_up(maxC, 32), maxB, m->n), dim3(32 * SA_SP_LANES), 0, st, *m);
  hipLaunchKernelGGL(sa_bias_multi_l2_kernel, dim3(sa_div_up(maxC, 128), m->n), dim3(128), 0, st, *m);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// ---------------------------------------------------------------------------------
// sa_reduce_finalize: slab reduction AND the finaliser that consumes it in ONE launch.
//
// The statistics of a convolution come out of its epilogue as per-tile partial slabs; until
// round 2 every use was two or three tiny launches (sa_sum_partials -> [sa_sum_rows_d ->] sa_fin_*),
// ~70 per train step.  Here workgroup (chunk j, utterance b) sums its 32 outputs over the slabs
// exactly like sa_sum_partials (lane q adds slabs q, q+16, ... in order, the 16 lane sums are added
// in lane order: bit-identical sums), writes them to rows[b][.] (fp64) and
//   * SA_FIN_IN_FWD : finalises its own 16 channels (InstanceNorm forward is per utterance);
//   * the modes that need a sum over utterances (BatchNorm forward / backward, d gamma / d beta of
//     InstanceNorm, bias gradients) take a ticket per chunk; the LAST workgroup of a chunk to arrive
//     adds the B rows in utterance order (fixed order: deterministic) and runs the finaliser for
//     its channels.  Hand-off per cdna_hip_programming.md Guideline 16: stores -> vmcnt(0) ->
//     barrier -> agent-scope release -> vmcnt(0) -> relaxed agent-scope ticket; the last arriver:
//     agent-scope acquire -> vmcnt(0) -> barrier -> plain loads.  The ticket resets itself.
// Not used when the sums must be all-reduced first (SyncBatchNorm under DDP): that path keeps
// the separate launches.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(32 * SA_SP_LANES) void sa_reduce_finalize_kernel(SaFinArgs a) {
  __shared__ double part[SA_SP_LANES][33];
  __shared__ double fin[32];
  __shared__ int is_last;
  const int bb = blockIdx.y, o = threadIdx.x & 31, q = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + o, n = a.n;
  double s = 0.0;
  if (i < n) {
    const float* p = a.part + (size_t)bb * a.nslab * n + i;
#pragma unroll 16
    for (int k = q; k < a.nslab; k += SA_SP_LANES) s += (double)p[(size_t)k * n];
  }
  part[q][o] = s;
  __syncthreads();
  if (q == 0) {
    double t = 0.0;
#pragma unroll
    for (int r = 0; r < SA_SP_LANES; ++r) t += part[r][o];
    fin[o] = t;
    if (i < n && a.rows) a.rows[(size_t)bb * n + i] = t;
  }
  const int C = a.C, nc = a.ncomp;
  if (a.mode == SA_FIN_IN_FWD || a.mode == SA_FIN_IN_BWD) {
    __syncthreads();
    // per (utterance, channel): this workgroup holds (S, Q) / (S1, S2) of its 16 channels
    if (q == 0 && (o & 1) == 0 && i < n) {
      const int c = i >> 1, g = bb * C + c;
      if (a.mode == SA_FIN_IN_FWD) {
        const double m = fin[o] / a.count;
        double var = fin[o + 1] / a.count - m * m;
        if (var < 0.0) var = 0.0;
        const float r = (float)(1.0 / sqrt(var + (double)a.eps));
        const float sc = a.gamma[c] * r;
        a.o0[g] = (float)m; a.o1[g] = r; a.o2[g] = sc; a.o3[g] = a.beta[c] - (float)m * sc;
      } else {
        const double S1 = fin[o], S2 = fin[o + 1];
        const double r = a.rstd[g], m = a.mean[g];
        const double k1 = (double)a.gamma[c] * r;
        a.o0[g] = a.sign * (float)k1;
        a.o1[g] = a.sign * (float)(-k1 * r * S2 / a.count);
        a.o2[g] = a.sign * (float)(k1 * (-S1 / a.count + m * r * S2 / a.count));
      }
    }
    if (a.mode == SA_FIN_IN_FWD) return;
  }
  // ---- cross-utterance stage: publish the row, take a ticket, the last arriver finalises ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned old = __hip_atomic_fetch_add(&a.tickets[blockIdx.x], 1u, __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT);
    const int last = old == (unsigned)(gridDim.y - 1);
    if (last) {
      __hip_atomic_store(&a.tickets[blockIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    is_last = last;
  }
  __syncthreads();
  if (!is_last || q != 0 || i >= n) return;
  double tot = 0.0;                                   // sum over utterances, in utterance order
  const int B = gridDim.y;
#pragma unroll 8
  for (int b = 0; b < B; ++b) tot += a.rows[(size_t)b * n + i];
  fin[o] = tot;
  // the 32 finishing lanes are one half-wave: no barrier needed between writing fin[] and reading
  // the neighbour's entry, but the compiler must not reorder the LDS accesses
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  const int c = i / nc, comp = i % nc;
  if (a.mode == SA_FIN_IN_BWD) {                      // d beta = sum_b S1, d gamma = sum_b S2
    if (a.dgamma) { if (comp == 0) a.dbeta[c] = (float)tot; else a.dgamma[c] = (float)tot; }
  } else if (a.mode == SA_FIN_BIAS) {
    if (comp == 0) a.db[c] = (float)tot;
  } else if (comp == 0) {
    const double S = fin[o], Q = fin[o + 1];
    if (a.mode == SA_FIN_BN_FWD) {
      const double m = S / a.count;
      double var = Q / a.count - m * m;
      if (var < 0.0) var = 0.0;
      const float r = (float)(1.0 / sqrt(var + (double)a.eps));
      const float sc = a.gamma[c] * r;
      a.o0[c] = (float)m; a.o1[c] = r; a.o2[c] = sc; a.o3[c] = a.beta[c] - (float)m * sc;
      if (a.run_mean) {
        const double unb = a.count > 1.0 ? var * a.count / (a.count - 1.0) : var;
        a.run_mean[c] = (1.0f - a.momentum) * a.run_mean[c] + a.momentum * (float)m;
        a.run_var[c] = (1.0f - a.momentum) * a.run_var[c] + a.momentum * (float)unb;
      }
    } else {                                          // SA_FIN_BN_BWD: S = S1, Q = S2
      const double r = a.rstd[c], m = a.mean[c];
      const double k1 = (double)a.gamma[c] * r;
      a.o0[c] = a.sign * (float)k1;
      a.o1[c] = a.sign * (float)(-k1 * r * Q / a.count);
      a.o2[c] = a.sign * (float)(k1 * (-S / a.count + m * r * Q / a.count));
      if (a.dgamma) { a.dbeta[c] = (float)S; a.dgamma[c] = (float)Q; }
    }
  }
}

extern "C" int sa_reduce_finalize(const SaFinArgs* a, void* stream) {
  if (!a || !a->part || a->nbatch <= 0 || a->nslab <= 0 || a->n <= 0 || a->C <= 0 || a->ncomp < 1 ||
      a->n != a->C * a->ncomp)
    return -22;
  const int m = a->mode;
  if (m < SA_FIN_IN_FWD || m > SA_FIN_BIAS) return -22;
  if (m != SA_FIN_BIAS && a->ncomp != 2) return -22;
  if (m != SA_FIN_IN_FWD && (!a->tickets || !a->rows)) return -22;
  if ((m == SA_FIN_IN_FWD || m == SA_FIN_BN_FWD) && (!a->gamma || !a->beta || !a->o0 || !a->o1 || !a->o2 || !a->o3))
    return -22;
  if ((m == SA_FIN_IN_BWD || m == SA_FIN_BN_BWD) && (!a->gamma || !a->mean || !a->rstd || !a->o0 || !a->o1 || !a->o2))
    return -22;
  if ((a->dgamma == nullptr) != (a->dbeta == nullptr)) return -22;
  if (m == SA_FIN_BIAS && !a->db) return -22;
  if (m != SA_FIN_BIAS && !(a->count > 0.0)) return -22;
  hipLaunchKernelGGL(sa_reduce_finalize_kernel, dim3(sa_div_up(a->n, 32), a->nbatch), dim3(32 * SA_SP_LANES), 0,
                     reinterpret_cast<hipStream_t>(stream), *a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}


// ---------------------------------------------------------------------------------
// sa_clip_grads: torch.nn.utils.clip_grad_norm_(params, max_norm) (speechbrain's check_gradients,
// speechbrain_convae_train.py:249) on the flat gradient buckets of a backward: total = sqrt(sum g^2) over
// all of them, coef = min(1, max_norm / (total + 1e-6)), g *= coef -- two launches instead of the ~8 of the
// foreach implementation over 56 tensors.  Deterministic: fixed partial layout, fixed summation order.
// ---------------------------------------------------------------------------------
#define SA_CLIP_BLKS 64
__global__ __launch_bounds__(256) void sa_clip_sumsq_kernel(SaFlats f, double* __restrict__ partials) {
  __shared__ double red[4];
  const float* p = f.f[blockIdx.y].p;
  const long long n = f.f[blockIdx.y].n;
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)SA_CLIP_BLKS * 256) s = fmaf(p[i], p[i], s);
  double d = (double)s;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.y * SA_CLIP_BLKS + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void sa_clip_scale_kernel(SaFlats f, const double* __restrict__ partials, float max_norm,
                                                            float eps, float* __restrict__ total_norm) {
  __shared__ float coef_s;
  __shared__ double red[4];
  // every workgroup adds the f.n * 64 <= 256 partials itself, one per thread, in a fixed tree
  double d = (int)threadIdx.x < f.n * SA_CLIP_BLKS ? partials[threadIdx.x] : 0.0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float tn = (float)sqrt((red[0] + red[1]) + (red[2] + red[3]));
    float c = max_norm / (tn + eps);
    coef_s = c < 1.0f ? c : 1.0f;
    if (blockIdx.x == 0 && blockIdx.y == 0 && total_norm) *total_norm = tn;
  }
  __syncthreads();
  const float c = coef_s;
  if (c >= 1.0f) return;                                    // (g * 1 = g: nothing to write)
  float* p = f.f[blockIdx.y].p;
  const long long n = f.f[blockIdx.y].n;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)SA_CLIP_BLKS * 256) p[i] *= c;
}

extern "C" int sa_clip_grads(const SaFlats* f, float max_norm, float eps, double* partials, float* total_norm,
                             void* stream) {
  if (!f || f->n <= 0 || f->n > SA_FLATS_MAX || !partials) return -22;
  for (int j = 0; j < f->n; ++j)
    if (!f->f[j].p || f->f[j].n <= 0) return -22;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(sa_clip_sumsq_kernel, dim3(SA_CLIP_BLKS, f->n), dim3(256), 0, st, *f, partials);
  hipLaunchKernelGGL(sa_clip_scale_kernel, dim3(SA_CLIP_BLKS, f->n), dim3(256), 0, st, *f, partials, max_norm, eps, total_norm);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}
