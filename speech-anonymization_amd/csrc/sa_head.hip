// Classifier head: statistics pooling with the reference's reshape quirk, the dense FC head
// on f32 MFMA, log-softmax, and the loss reductions with fused gradients.
//
// Reference: models/ConvAutoEncoder.py:57-69 (TDNNSexClassifier.forward: reshape at :61 is a
// MEMORY REINTERPRETATION of the [B,128,L'] tensor as [B,L',128], then speechbrain
// StatisticsPooling mean/std over dim 1), :47-55 (classify), speechbrain_convae_train.py:105-108
// (recon / NLL / confusion losses), utils/cosine_similarity_loss.py:53-56.
#include "sa_common.h"

// ---------------------------------------------------------------------------------
// Statistics pooling.  The classifier's last BatchNorm output, in the reference's channel-
// major memory order, has flat index f = c*L + l; the reshape regroups it so that pooled
// column j collects every f with f % 128 == j.  Here the tensor is channels-last [B][L][128]
// (stored ReLU output r, BN affine applied on the fly), so element (l, c) belongs to column
// j = (c*L + l) % 128: for a fixed channel the column advances with l, so all rows with the same
// l mod 128 fall into the same column.  Stage 1 (sa_pool_fwd) is therefore a plain strided
// reduction with no transposition: thread (lm, 16-byte channel chunk) sums rows lm, lm+128, ...
// of its segment in registers (coalesced: a workgroup reads 8 or 16 consecutive rows per step)
// and leaves S[b][seg][c][lm] = (sum, sumsq); stage 2 (sa_pool_gather) rotates: column j
// collects S[..][c][(j - c*L) mod 128] over c and the segments in a fixed order (fp64).
// ---------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void sa_pool_fwd_kernel(const T* __restrict__ r,
                                                          const float* __restrict__ sc,
                                                          const float* __restrict__ sh,
                                                          float* __restrict__ part, int L,
                                                          int nseg) {
  constexpr int C = 128, VEC = Tr<T>::VEC, CH = C / VEC, RPB = 256 / CH, NG = 128 / RPB;
  __shared__ float tile[RPB][C][2];
  const int tid = threadIdx.x, b = blockIdx.y, grp = blockIdx.x % NG, seg = blockIdx.x / NG;
  const int c = tid % CH, rl = tid / CH, lm = grp * RPB + rl;
  float s[VEC], t[VEC], sum[VEC], sq[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) { s[j] = sc[c * VEC + j]; t[j] = sh[c * VEC + j]; sum[j] = 0.f; sq[j] = 0.f; }
  const int nk = (L + 127) / 128, per = (nk + nseg - 1) / nseg;
  const int kbeg = seg * per, kend = (kbeg + per < nk) ? kbeg + per : nk;
  const T* rb = r + (size_t)b * L * C + c * VEC;
  for (int k0 = kbeg; k0 < kend; k0 += 4) {
    uint4 raw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int l = lm + 128 * (k0 + i);
      raw[i] = make_uint4(0, 0, 0, 0);
      if (k0 + i < kend && l < L) raw[i] = *reinterpret_cast<const uint4*>(rb + (size_t)l * C);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int l = lm + 128 * (k0 + i);
      if (k0 + i < kend && l < L) {
        float f[VEC];
        Tr<T>::unpack(raw[i], f);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float v = fmaf(f[j], s[j], t[j]);
          sum[j] += v; sq[j] = fmaf(v, v, sq[j]);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) { tile[rl][c * VEC + j][0] = sum[j]; tile[rl][c * VEC + j][1] = sq[j]; }
  __syncthreads();
  // [c][lm] layout: RPB consecutive lm (x 2 values) per channel are contiguous
  float* dst = part + ((size_t)b * nseg + seg) * C * 128 * 2;
  for (int e = tid; e < RPB * C * 2; e += 256) {
    const int cc = e / (RPB * 2), w = e % (RPB * 2);
    dst[((size_t)cc * 128 + grp * RPB) * 2 + w] = tile[w >> 1][cc][w & 1];
  }
}

// sums[b][j] = sum_seg sum_c S[b][seg][c][(j - c*L) mod 128]   (fp64, fixed order)
// 1024 threads: eight groups of 16 channels per pooled column, added in group order (one workgroup per
// utterance: at B = 10 the 256-thread version spent 32 us on 512 dependent loads per thread)
__global__ __launch_bounds__(1024) void sa_pool_gather_kernel(const float* __restrict__ part, int nseg,
                                                              int L, double* __restrict__ sums) {
  __shared__ double grp[7][128][2];
  const int tid = threadIdx.x, b = blockIdx.x, j = tid & 127, h = tid >> 7, Lm = L % 128;
  double s = 0.0, q = 0.0;
  for (int seg = 0; seg < nseg; ++seg) {
    const float2* p = reinterpret_cast<const float2*>(part) + ((size_t)b * nseg + seg) * 128 * 128;
#pragma unroll 16
    for (int c = h * 16; c < h * 16 + 16; ++c) {
      const float2 v = p[(size_t)c * 128 + ((j - c * Lm) & 127)];
      s += v.x; q += v.y;
    }
  }
  if (h > 0) { grp[h - 1][j][0] = s; grp[h - 1][j][1] = q; }
  __syncthreads();
  if (h == 0) {
#pragma unroll
    for (int g = 0; g < 7; ++g) { s += grp[g][j][0]; q += grp[g][j][1]; }
    sums[((size_t)b * 128 + j) * 2 + 0] = s;
    sums[((size_t)b * 128 + j) * 2 + 1] = q;
  }
}

// segments per utterance: enough workgroups (16 or 8 row groups x B x nseg) to fill the chip
extern "C" int sa_pool_nseg(int B) {
  int n = 2048 / (B * 16);
  return n < 1 ? 1 : (n > 8 ? 8 : n);
}

extern "C" int sa_pool_fwd(int dtype, const void* r, const float* scale, const float* shift,
                           float* part, int B, int L, int nseg, void* stream) {
  if (!r || !scale || !shift || !part || B <= 0 || L <= 1 || nseg < 1) return -22;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == SA_BF16)
    hipLaunchKernelGGL(sa_pool_fwd_kernel<bf16_t>, dim3(8 * nseg, B), dim3(256), 0, st,
                       reinterpret_cast<const bf16_t*>(r), scale, shift, part, L, nseg);
  else
    hipLaunchKernelGGL(sa_pool_fwd_kernel<float>, dim3(16 * nseg, B), dim3(256), 0, st,
                       reinterpret_cast<const float*>(r), scale, shift, part, L, nseg);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

extern "C" int sa_pool_gather(const float* part, int B, int nseg, int L, double* sums, void* stream) {
  if (!part || !sums || B <= 0 || nseg < 1) return -22;
  hipLaunchKernelGGL(sa_pool_gather_kernel, dim3(B), dim3(1024), 0,
                     reinterpret_cast<hipStream_t>(stream), part, nseg, L, sums);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// sums [B][128][2] -> pooled [B][256] = (mean + eps*((1-9)*noise+9) if noise, std_unbiased + eps)
// and saves mean / raw std for the backward.
__global__ void sa_pool_fin_kernel(const double* __restrict__ sums, int B, int n,
                                   const float* __restrict__ noise, float eps, float* pooled,
                                   float* mean, float* stdraw) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * 128) return;
  const int b = i / 128, j = i % 128;
  const double S = sums[2 * i], Q = sums[2 * i + 1];
  const double m = S / n;
  double var = (Q - S * m) / (n - 1);
  if (var < 0.0) var = 0.0;
  const float sd = (float)sqrt(var);
  mean[i] = (float)m; stdraw[i] = sd;
  float mo = (float)m;
  if (noise) mo += eps * ((1.0f - 9.0f) * noise[i] + 9.0f);
  pooled[(size_t)b * 256 + j] = mo;
  pooled[(size_t)b * 256 + 128 + j] = sd + eps;
}

extern "C" int sa_pool_fin(const double* sums, int B, int n, const float* noise, float eps,
                           float* pooled, float* mean, float* stdraw, void* stream) {
  if (!sums || !pooled || !mean || !stdraw || n < 2) return -22;
  hipLaunchKernelGGL(sa_pool_fin_kernel, dim3(sa_div_up(B * 128, 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), sums, B, n, noise, eps, pooled, mean,
                     stdraw);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// Pooling backward: g[b][l][c] = dmean_j/n + dstd_j * (xbn - mean_j) / ((n-1) * std_j),
// j = (c*L + l) % 128, xbn = r*scale[c] + shift[c].   dpooled [B][256].
// With stats != null the kernel also leaves the partial sums the BatchNorm backward of the
// pooled tensor needs, (sum g, sum g*(r - bn_mean[c])*bn_rstd[c]) per channel, one slab per
// workgroup [B][ceil(L/256)][128][2] (a separate sa_ew_stats pass would re-read g and r).
template <typename T>
__global__ __launch_bounds__(256) void sa_pool_bwd_kernel(const T* __restrict__ r,
                                                          const float* __restrict__ sc,
                                                          const float* __restrict__ sh,
                                                          const float* __restrict__ dpooled,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ stdraw,
                                                          T* __restrict__ g, int L,
                                                          const float* __restrict__ bn_mean,
                                                          const float* __restrict__ bn_rstd,
                                                          float* __restrict__ stats) {
  constexpr int C = 128, VEC = Tr<T>::VEC, CH = C / VEC, RPP = 256 / CH;
  __shared__ float ka[128], kb[128];
  __shared__ float red[RPP][C][2];
  const int tid = threadIdx.x, b = blockIdx.y, l0 = blockIdx.x * SA_WAVE * 4;
  if (tid < 128) {
    const float dm = dpooled[(size_t)b * 256 + tid], ds = dpooled[(size_t)b * 256 + 128 + tid];
    const float sd = stdraw[(size_t)b * 128 + tid], m = mean[(size_t)b * 128 + tid];
    const float inv = sd > 0.0f ? 1.0f / ((float)(L - 1) * sd) : 0.0f;
    kb[tid] = ds * inv;
    ka[tid] = dm / (float)L - ds * inv * m;
  }
  __syncthreads();
  const int c = tid % CH, r0 = tid / CH;
  float s[VEC], t[VEC];
  int jb[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    s[j] = sc[c * VEC + j]; t[j] = sh[c * VEC + j];
    jb[j] = (int)(((long long)(c * VEC + j) * L) % 128);
  }
  float bm[VEC], br[VEC], s1[VEC], s2[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    bm[j] = stats ? bn_mean[c * VEC + j] : 0.0f;
    br[j] = stats ? bn_rstd[c * VEC + j] : 0.0f;
    s1[j] = 0.0f; s2[j] = 0.0f;
  }
  for (int rr = r0; rr < 256; rr += RPP) {
    const int l = l0 + rr;
    if (l >= L) break;
    float f[VEC], x[VEC];
    Tr<T>::unpack(*reinterpret_cast<const uint4*>(r + ((size_t)b * L + l) * C + c * VEC), x);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int col = (jb[j] + l) % 128;
      f[j] = fmaf(kb[col], fmaf(x[j], s[j], t[j]), ka[col]);
    }
    const uint4 u = Tr<T>::pack(f);
    *reinterpret_cast<uint4*>(g + ((size_t)b * L + l) * C + c * VEC) = u;
    if (stats) {
      Tr<T>::unpack(u, f);                             // the stored (rounded) gradient
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        s1[j] += f[j];
        s2[j] = fmaf(f[j], (x[j] - bm[j]) * br[j], s2[j]);
      }
    }
  }
  if (stats) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) { red[r0][c * VEC + j][0] = s1[j]; red[r0][c * VEC + j][1] = s2[j]; }
    __syncthreads();
    if (tid < C) {
      float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
      for (int q = 0; q < RPP; ++q) { a0 += red[q][tid][0]; a1 += red[q][tid][1]; }
      float* dst = stats + (((size_t)b * gridDim.x + blockIdx.x) * C + tid) * 2;
      dst[0] = a0; dst[1] = a1;
    }
  }
}

extern "C" int sa_pool_bwd(int dtype, const void* r, const float* scale, const float* shift,
                           const float* dpooled, const float* mean, const float* stdraw, void* g,
                           int B, int L, const float* bn_mean, const float* bn_rstd, float* stats,
                           void* stream) {
  if (!r || !scale || !shift || !dpooled || !mean || !stdraw || !g || L < 2) return -22;
  if (stats && (!bn_mean || !bn_rstd)) return -22;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(sa_div_up(L, 256), B);
  if (dtype == SA_BF16)
    hipLaunchKernelGGL(sa_pool_bwd_kernel<bf16_t>, grid, dim3(256), 0, st,
                       reinterpret_cast<const bf16_t*>(r), scale, shift, dpooled, mean, stdraw,
                       reinterpret_cast<bf16_t*>(g), L, bn_mean, bn_rstd, stats);
  else
    hipLaunchKernelGGL(sa_pool_bwd_kernel<float>, grid, dim3(256), 0, st,
                       reinterpret_cast<const float*>(r), scale, shift, dpooled, mean, stdraw,
                       reinterpret_cast<float*>(g), L, bn_mean, bn_rstd, stats);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// ---------------------------------------------------------------------------------
// Dense layers of the FC head on exact-f32 MFMA (v_mfma_f32_32x32x2_f32).
//   Y[m][n] = act( sum_k A(m,k) * Bm(k,n) + bias[n] ),   A(m,k) = P(X[m*lda + k])
//   P(v) = (v*ps[k] + pt[k]) (BatchNorm of the previous layer folded in), Bm(k,n) = W[k*sbk + n*sbn]
// One wave per 32x32 output tile; rows >= M / cols >= N are masked.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sa_dense_kernel(const float* __restrict__ X, int lda,
                                                       const float* __restrict__ ps,
                                                       const float* __restrict__ pt,
                                                       const float* __restrict__ W, int sbk, int sbn,
                                                       const float* __restrict__ bias, float* __restrict__ Y,
                                                       int ldy, int M, int N, int K, int relu) {
  // four waves share one 32x32 output tile: wave w takes the 16-deep k chunks w, w+4, ... (these
  // GEMMs are latency-bound, M = batch size), the four partial tiles are added in wave order
  __shared__ float part[4][16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
  const int m = m0 + (lane & 31), n = n0 + (lane & 31), kh = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
  for (int k0 = wave * 16; k0 < K; k0 += 64) {
    float av[8], bv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = k0 + 2 * u + kh;
      av[u] = 0.0f; bv[u] = 0.0f;
      if (k < K) {
        if (m < M) {
          av[u] = X[(size_t)m * lda + k];
          if (ps) av[u] = fmaf(av[u], ps[k], pt[k]);
        }
        if (n < N) bv[u] = W[(size_t)k * sbk + (size_t)n * sbn];
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (k0 + 2 * u < K) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) part[wave][i][lane] = acc[i];
  __syncthreads();
  if (wave == 0 && n < N) {
    const float bb = bias ? bias[n] : 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int mr = m0 + sa_acc_row(i, lane);
      if (mr < M) {
        float v = ((part[0][i][lane] + part[1][i][lane]) + part[2][i][lane]) + part[3][i][lane] + bb;
        if (relu) v = fmaxf(v, 0.0f);
        Y[(size_t)mr * ldy + n] = v;
      }
    }
  }
}

extern "C" int sa_dense(const float* X, int lda, const float* ps, const float* pt, const float* W,
                        int sbk, int sbn, const float* bias, float* Y, int ldy, int M, int N, int K,
                        int relu, void* stream) {
  if (!X || !W || !Y || M <= 0 || N <= 0 || K <= 0) return -22;
  dim3 grid(sa_div_up(N, 32), sa_div_up(M, 32));
  hipLaunchKernelGGL(sa_dense_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), X,
                     lda, ps, pt, W, sbk, sbn, bias, Y, ldy, M, N, K, relu);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// column sums of a small [M][N] matrix: sums[n][0] = sum_m X, sums[n][1] = sum_m X*Xh
// where Xh = X (sumsq, Xh null) or a second matrix (e.g. normalised activations).
__global__ void sa_colsums_kernel(const float* __restrict__ X, const float* __restrict__ H,
                                  const float* __restrict__ hm, const float* __restrict__ hr,
                                  int M, int N, double* sums, float* __restrict__ out0,
                                  float* __restrict__ out1) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  double s = 0.0, q = 0.0;
#pragma unroll 8
  for (int m = 0; m < M; ++m) {
    const float x = X[(size_t)m * N + n];
    float h = H ? H[(size_t)m * N + n] : x;
    if (hm) h = (h - hm[n]) * hr[n];
    s += x; q += (double)x * h;
  }
  sums[2 * n] = s; sums[2 * n + 1] = q;
  if (out0) out0[n] = (float)s;
  if (out1) out1[n] = (float)q;
}

extern "C" int sa_colsums(const float* X, const float* H, const float* hmean, const float* hrstd,
                          int M, int N, double* sums, float* out0, float* out1, void* stream) {
  if (!X || !sums) return -22;
  hipLaunchKernelGGL(sa_colsums_kernel, dim3(sa_div_up(N, 64)), dim3(64), 0,
                     reinterpret_cast<hipStream_t>(stream), X, H, hmean, hrstd, M, N, sums, out0,
                     out1);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// BatchNorm backward on a small [M][N] activation (batch statistics over `count` rows,
// possibly global): dH = gamma*rstd*(G - S1/count - hhat*S2/count) [* (H > 0)]
__global__ void sa_bn2d_bwd_kernel(const float* __restrict__ G, const float* __restrict__ H,
                                   const double* __restrict__ sums, double count,
                                   const float* __restrict__ gamma, const float* __restrict__ mean,
                                   const float* __restrict__ rstd, int relu_mask, int M, int N,
                                   float* dH, const double* __restrict__ count_dev) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * N) return;
  if (count_dev) count = *count_dev;                    // device-side (all-reduced) row count
  const int n = i % N;
  const float hh = (H[i] - mean[n]) * rstd[n];
  float v = gamma[n] * rstd[n] * (G[i] - (float)(sums[2 * n] / count) - hh * (float)(sums[2 * n + 1] / count));
  if (relu_mask && !(H[i] > 0.0f)) v = 0.0f;
  dH[i] = v;
}

extern "C" int sa_bn2d_bwd(const float* G, const float* H, const double* sums, double count,
                           const float* gamma, const float* mean, const float* rstd, int relu_mask,
                           int M, int N, float* dH, const double* count_dev, void* stream) {
  if (!G || !H || !sums || !gamma || !mean || !rstd || !dH) return -22;
  hipLaunchKernelGGL(sa_bn2d_bwd_kernel, dim3(sa_div_up(M * N, 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), G, H, sums, count, gamma, mean, rstd,
                     relu_mask, M, N, dH, count_dev);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// dW[n][k] = sum_m dY[m][n] * P(X[m][k])  (P = BatchNorm affine of the layer input)
__global__ void sa_dense_wgrad_kernel(const float* __restrict__ dY, const float* __restrict__ X,
                                      const float* __restrict__ ps, const float* __restrict__ pt,
                                      int M, int N, int K, float* dW) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * K) return;
  const int n = i / K, k = i % K;
  float s = 0.0f;
#pragma unroll 8
  for (int m = 0; m < M; ++m) {
    float x = X[(size_t)m * K + k];
    if (ps) x = fmaf(x, ps[k], pt[k]);
    s = fmaf(dY[(size_t)m * N + n], x, s);
  }
  dW[i] = s;
}

extern "C" int sa_dense_wgrad(const float* dY, const float* X, const float* ps, const float* pt,
                              int M, int N, int K, float* dW, void* stream) {
  if (!dY || !X || !dW) return -22;
  hipLaunchKernelGGL(sa_dense_wgrad_kernel, dim3(sa_div_up(N * K, 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), dY, X, ps, pt, M, N, K, dW);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// log_softmax over the last dim of [M][N] (N small), and its backward
__global__ void sa_log_softmax_kernel(const float* __restrict__ X, float* __restrict__ Y, int M, int N) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  float mx = -INFINITY;
  for (int n = 0; n < N; ++n) mx = fmaxf(mx, X[(size_t)m * N + n]);
  float s = 0.0f;
  for (int n = 0; n < N; ++n) s += expf(X[(size_t)m * N + n] - mx);
  const float lse = mx + logf(s);
  for (int n = 0; n < N; ++n) Y[(size_t)m * N + n] = X[(size_t)m * N + n] - lse;
}
__global__ void sa_log_softmax_bwd_kernel(const float* __restrict__ dY, const float* __restrict__ Y,
                                          float* __restrict__ dX, int M, int N) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  float s = 0.0f;
  for (int n = 0; n < N; ++n) s += dY[(size_t)m * N + n];
  for (int n = 0; n < N; ++n) dX[(size_t)m * N + n] = dY[(size_t)m * N + n] - expf(Y[(size_t)m * N + n]) * s;
}
extern "C" int sa_log_softmax(const float* X, float* Y, int M, int N, void* stream) {
  if (!X || !Y) return -22;
  hipLaunchKernelGGL(sa_log_softmax_kernel, dim3(sa_div_up(M, 64)), dim3(64), 0,
                     reinterpret_cast<hipStream_t>(stream), X, Y, M, N);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}
extern "C" int sa_log_softmax_bwd(const float* dY, const float* Y, float* dX, int M, int N, void* stream) {
  if (!dY || !Y || !dX) return -22;
  hipLaunchKernelGGL(sa_log_softmax_bwd_kernel, dim3(sa_div_up(M, 64)), dim3(64), 0,
                     reinterpret_cast<hipStream_t>(stream), dY, Y, dX, M, N);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// ---------------------------------------------------------------------------------
// Loss reductions with fused gradients.
// ---------------------------------------------------------------------------------
// recon loss: mean over N of |a-b| (kind 0, nn.L1Loss) or (a-b)^2 (kind 1, nn.MSELoss);
// grad[i] = d loss / d a[i].  Two-level deterministic reduction: part[block] then one thread.
__global__ __launch_bounds__(256) void sa_recon_loss_kernel(const float* __restrict__ a,
                                                            const float* __restrict__ b, size_t n,
                                                            int kind, float* __restrict__ grad,
                                                            double* __restrict__ part) {
  __shared__ double wsum[4];
  const float inv = 1.0f / (float)n;
  double s = 0.0;
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 x = reinterpret_cast<const float4*>(a)[i], y = reinterpret_cast<const float4*>(b)[i];
    float d[4] = {x.x - y.x, x.y - y.y, x.z - y.z, x.w - y.w}, g[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (kind == 0) { s += fabsf(d[j]); g[j] = d[j] > 0.f ? inv : (d[j] < 0.f ? -inv : 0.f); }
      else { s += (double)d[j] * d[j]; g[j] = 2.0f * d[j] * inv; }
    }
    if (grad) reinterpret_cast<float4*>(grad)[i] = make_float4(g[0], g[1], g[2], g[3]);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (size_t i = n4 * 4; i < n; ++i) {
      const float d = a[i] - b[i];
      if (kind == 0) { s += fabsf(d); if (grad) grad[i] = d > 0.f ? inv : (d < 0.f ? -inv : 0.f); }
      else { s += (double)d * d; if (grad) grad[i] = 2.0f * d * inv; }
    }
  s = sa_wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
// 64 threads: thread i adds partials i, i+64, ... in order, thread 0 adds the 64 lane sums in order
__global__ void sa_recon_loss_fin_kernel(const double* part, int nb, double n, float* loss) {
  __shared__ double lane_sum[64];
  double s = 0.0;
#pragma unroll 8
  for (int i = threadIdx.x; i < nb; i += 64) s += part[i];
  lane_sum[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < 64; ++i) t += lane_sum[i];
    loss[0] = (float)(t / n);
  }
}
#define SA_LOSS_BLOCKS 512
extern "C" int sa_loss_workspace_bytes() { return SA_LOSS_BLOCKS * (int)sizeof(double); }
extern "C" int sa_recon_loss(const float* a, const float* b, long long n, int kind, float* grad,
                             float* loss, void* workspace, void* stream) {
  if (!a || !b || !loss || !workspace || n <= 0) return -22;
  if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) |
       reinterpret_cast<uintptr_t>(grad)) & 15) return -22;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int nb = (int)((n / 4 + 255) / 256);
  if (nb < 1) nb = 1;
  if (nb > SA_LOSS_BLOCKS) nb = SA_LOSS_BLOCKS;
  hipLaunchKernelGGL(sa_recon_loss_kernel, dim3(nb), dim3(256), 0, st, a, b, (size_t)n, kind, grad,
                     reinterpret_cast<double*>(workspace));
  hipLaunchKernelGGL(sa_recon_loss_fin_kernel, dim3(1), dim3(64), 0, st,
                     reinterpret_cast<const double*>(workspace), nb, (double)n, loss);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// NLLLoss(mean) and the confusion MSE against log(0.5) = -0.6931 (the reference's literal,
// speechbrain_convae_train.py:107-108) on logp [B][2]; out = (nll, conf); grads per element.
__global__ __launch_bounds__(64) void sa_cls_losses_kernel(const float* __restrict__ logp,
                                                           const long long* __restrict__ label, int B, int NC,
                                                           float* out, float* dnll, float* dconf) {
  // lane t takes elements t, t+64, ...; the 64 lane sums are added in lane order
  __shared__ double pn[64], pc[64];
  const int t = threadIdx.x;
  double nll = 0.0, conf = 0.0;
  for (int e = t; e < B * NC; e += 64) {
    const int b = e / NC, c = e % NC;
    const float lp = logp[e];
    const float d = lp - (-0.6931f);
    conf += (double)d * d;
    if (dconf) dconf[e] = 2.0f * d / (float)(B * NC);
    const bool hit = (long long)c == label[b];
    if (hit) nll -= lp;
    if (dnll) dnll[e] = hit ? -1.0f / (float)B : 0.0f;
  }
  pn[t] = nll; pc[t] = conf;
  __syncthreads();
  if (t == 0) {
    double a = 0.0, q = 0.0;
    for (int i = 0; i < 64; ++i) { a += pn[i]; q += pc[i]; }
    out[0] = (float)(a / B);
    out[1] = (float)(q / (B * NC));
  }
}
extern "C" int sa_cls_losses(const float* logp, const long long* label, int B, int NC, float* out,
                             float* dnll, float* dconf, void* stream) {
  if (!logp || !label || !out) return -22;
  hipLaunchKernelGGL(sa_cls_losses_kernel, dim3(1), dim3(64), 0,
                     reinterpret_cast<hipStream_t>(stream), logp, label, B, NC, out, dnll, dconf);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// CosineSimilarityLoss (utils/cosine_similarity_loss.py:53-56):
//   loss = sum_{b,s} (1 - cos(x1[b,s,:], x2[b,s,:]; eps 1e-6)) / S.     One wave per (b,s) row.
// torch.cosine_similarity: dot / max(|x1|*|x2|, eps)  [ATen: (x1.x2) / sqrt(max(|x1|^2*|x2|^2, eps^2))]
// dx1 (optional) = d loss / d x1 = -(1/S) * (x2/(|x1||x2|) - cos * x1/|x1|^2)
__global__ __launch_bounds__(256) void sa_cosine_rows_kernel(const float* __restrict__ x1,
                                                             const float* __restrict__ x2,
                                                             int rows, int D, int S, float eps,
                                                             float* __restrict__ rowloss,
                                                             float* __restrict__ dx1) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* a = x1 + (size_t)row * D;
  const float* b = x2 + (size_t)row * D;
  float dot = 0.f, na = 0.f, nb = 0.f;
  for (int i = lane; i < D; i += 64) { dot = fmaf(a[i], b[i], dot); na = fmaf(a[i], a[i], na); nb = fmaf(b[i], b[i], nb); }
  dot = sa_wave_sum(dot); na = sa_wave_sum(na); nb = sa_wave_sum(nb);
  const float den2 = fmaxf(na * nb, eps * eps);
  const float cs = dot / sqrtf(den2);
  if (lane == 0) rowloss[row] = 1.0f - cs;
  if (dx1) {
    const bool clamped = na * nb < eps * eps;
    const float inv = 1.0f / sqrtf(den2), k = -1.0f / (float)S;
    for (int i = lane; i < D; i += 64) {
      float g = b[i] * inv;
      if (!clamped) g -= cs * a[i] / na;
      dx1[(size_t)row * D + i] = k * g;
    }
  }
}
__global__ void sa_cosine_fin_kernel(const float* rowloss, int rows, int S, float* loss) {
  double s = 0.0;
  for (int i = 0; i < rows; ++i) s += rowloss[i];
  loss[0] = (float)(s / S);
}
extern "C" int sa_cosine_loss(const float* x1, const float* x2, int B, int S, int D, float* rowloss,
                              float* loss, float* dx1, void* stream) {
  if (!x1 || !x2 || !rowloss || !loss || B <= 0 || S <= 0 || D <= 0) return -22;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int rows = B * S;
  hipLaunchKernelGGL(sa_cosine_rows_kernel, dim3(sa_div_up(rows, 4)), dim3(256), 0, st, x1, x2, rows,
                     D, S, 1e-6f, rowloss, dx1);
  hipLaunchKernelGGL(sa_cosine_fin_kernel, dim3(1), dim3(1), 0, st, rowloss, rows, S, loss);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// ---------------------------------------------------------------------------------
// x-vector TDNN layer (forward, eval): speechbrain Conv1d ("same" reflect padding, dilation)
// -> LeakyReLU -> BatchNorm1d(eval), the block of models/external_gender_classifiers.py:71-87
// (Xvector, used through evaluator_inference.yaml:34-41).  Activations are [B][T][C] already.
//   y[b][t][co] = bn_s[co] * leaky( bias[co] + sum_{k,ci} x[b][refl(t + k*dil - pad)][ci] * w[co][ci][k] ) + bn_t[co]
// Tiled GEMM on the bf16 MFMA with split (hi/lo) operands, like the conv kernels: one 4-wave
// workgroup = 128 frames x 128 output channels of one utterance; the input channels go through
// LDS in chunks of <= 64 (frames + reflect halo staged and split once per chunk, every tap is a
// row offset into the staged tile), weights come as the fragment-major image of sa_pack_weights
// (SA_BF16X3, N padded to a multiple of 128) straight from L2.  Evaluation-only path.
// ---------------------------------------------------------------------------------
#define SA_TD_BM 128
#define SA_TD_BN 128
#define SA_TD_CK 64
#define SA_TD_HALO 8                      // >= dil*(K-1) for the x-vector layers (k5 d1, k3 d2, k3 d3)
__global__ __launch_bounds__(256, 2) void sa_tdnn_fwd_kernel(const float* __restrict__ x,
                                                             const bf16x8* __restrict__ wp,
                                                             const float* __restrict__ bias,
                                                             const float* __restrict__ bn_s,
                                                             const float* __restrict__ bn_t,
                                                             float* __restrict__ y, int T, int Cin,
                                                             int Cout, int Npad, int K, int dil,
                                                             float slope, unsigned char* __restrict__ mask) {
  constexpr int PITCH = SA_TD_CK + 8, ROWS = SA_TD_BM + SA_TD_HALO, PLANE = ROWS * PITCH;
  __shared__ __attribute__((aligned(16))) bf16_t As[2 * PLANE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int t0 = blockIdx.x * SA_TD_BM, n0 = blockIdx.y * SA_TD_BN, b = blockIdx.z;
  const int pad = dil * (K - 1) / 2, nrows = SA_TD_BM + 2 * pad;
  const int KSTEPS = Cin / 16, NT = Npad / 32;
  const size_t lo_off = (size_t)K * KSTEPS * NT * 64;            // fragments per weight plane
  f32x16 acc[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.0f;
  const float* xb = x + (size_t)b * T * Cin;
  for (int c0 = 0; c0 < Cin; c0 += SA_TD_CK) {
    const int ck = Cin - c0 < SA_TD_CK ? Cin - c0 : SA_TD_CK, chunks = ck / 4;
    // ---- stage rows t0-pad .. t0+BM+pad (reflected at the utterance ends), split hi / lo ----
    for (int e = tid; e < nrows * chunks; e += 256) {
      const int r = e / chunks, c = e % chunks;
      int tt = t0 + r - pad;
      if (tt < 0) tt = -tt;                                     // reflect (no edge repeat)
      if (tt >= T) tt = 2 * (T - 1) - tt;
      float f[4] = {0.f, 0.f, 0.f, 0.f};
      if (tt >= 0 && tt < T) {
        const float4 v = *reinterpret_cast<const float4*>(xb + (size_t)tt * Cin + c0 + c * 4);
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
      }
      uint2 hi, lo;
      sa_split4(f, hi, lo);
      *reinterpret_cast<uint2*>(As + r * PITCH + c * 4) = hi;
      *reinterpret_cast<uint2*>(As + PLANE + r * PITCH + c * 4) = lo;
    }
    __syncthreads();
    const int ksteps = ck / 16;
    for (int k = 0; k < K; ++k) {
      for (int ks = 0; ks < ksteps; ++ks) {
        const bf16x8* wt = wp + (((size_t)k * KSTEPS + c0 / 16 + ks) * NT + n0 / 32 + wn * 2) * 64 + lane;
        bf16x8 bh[2], bl[2], ah[2], al[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) { bh[nt] = wt[nt * 64]; bl[nt] = wt[lo_off + nt * 64]; }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const bf16_t* ap = As + (wm * 64 + mt * 32 + (lane & 31) + k * dil) * PITCH + ks * 16 + 8 * (lane >> 5);
          ah[mt] = *reinterpret_cast<const bf16x8*>(ap);
          al[mt] = *reinterpret_cast<const bf16x8*>(ap + PLANE);
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
          }
      }
    }
    __syncthreads();
  }
  // ---- bias -> LeakyReLU -> BatchNorm(eval) affine -> store ----
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int n = n0 + (wn * 2 + nt) * 32 + (lane & 31);
    if (n < Cout) {
      const float bb = bias ? bias[n] : 0.0f, sc = bn_s ? bn_s[n] : 1.0f, sh = bn_t ? bn_t[n] : 0.0f;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int t = t0 + wm * 64 + mt * 32 + sa_acc_row(i, lane);
          if (t < T) {
            float v = acc[mt][nt][i] + bb;
            if (mask) mask[((size_t)b * T + t) * Cout + n] = v > 0.0f;   // LeakyReLU branch, for the backward
            v = v > 0.0f ? v : v * slope;
            y[((size_t)b * T + t) * Cout + n] = fmaf(v, sc, sh);
          }
        }
    }
  }
}

// wp: sa_pack_weights(SA_BF16X3, w padded to Npad output channels, ntaps = K, K = Cin, N = Npad,
// sk = K, sn = Cin*K, st = 1); Npad % 128 == 0, Cin % 16 == 0.
extern "C" int sa_tdnn_fwd(const float* x, const void* wp, const float* bias, const float* bn_s,
                           const float* bn_t, float* y, int B, int T, int Cin, int Cout, int Npad, int K,
                           int dil, float slope, unsigned char* mask, void* stream) {
  if (!x || !wp || !y || B <= 0 || T <= 0 || Cin <= 0 || Cout <= 0 || K < 1 || !(K & 1) || dil < 1 ||
      dil * (K - 1) / 2 >= T || dil * (K - 1) > SA_TD_HALO || Cin % 16 || Npad % SA_TD_BN || Npad < Cout)
    return -22;
  dim3 grid(sa_div_up(T, SA_TD_BM), Npad / SA_TD_BN, B);
  hipLaunchKernelGGL(sa_tdnn_fwd_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x,
                     reinterpret_cast<const bf16x8*>(wp), bias, bn_s, bn_t, y, T, Cin, Cout, Npad, K, dil,
                     slope, mask);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// ---------------------------------------------------------------------------------
// Input gradient of a (frozen, eval-mode) TDNN block  y = BN_eval(LeakyReLU(conv_same_reflect(x)))
// (models/EndToEnd.py:57-61,81: the pretrained x-vector classifier sits in the training graph
// with requires_grad off, so only d loss / d x is needed; blocks as in
// models/external_gender_classifiers.py:71-100).
//   d z[t][c] = d y[t][c] * bn_s[c] * (z > 0 ? 1 : slope),   z > 0 from the branch mask the forward
//               wrote (recovering the sign from the stored y = s*leaky(z) + t cancels for small z)
//   d xe[p]   = sum_k W_k^T d z[p + pad - k*dil]              on the EXTENDED range p in [-pad, T + pad)
//   d x[j]    = d xe[j] + d xe[-j] (1 <= j <= pad) + d xe[2(T-1) - j] (T-1-pad <= j <= T-2)
// (the last line is the adjoint of the reflect padding: sa_tdnn_fold).  Same tiling as the forward
// kernel: 128 rows x 128 output channels per 4-wave workgroup, reduction channels through LDS in
// chunks of 64 with the tap halo, split-bf16 operands, taps walked in reverse.
// wp: sa_pack_weights(SA_BF16X3, ...) image of the Conv1d weight as a DATA-GRADIENT operand
// (reduction = the block's output channels, zero-padded to Cred % 16 == 0; produced = its input
// channels, zero-padded to Npad % 128 == 0): ntaps = K, K = Cred, N = Npad, sk = Cin_w*K, sn = K, st = 1.
__global__ __launch_bounds__(256, 2) void sa_tdnn_bwd_kernel(const float* __restrict__ dy,
                                                              const unsigned char* __restrict__ mask,
                                                              const float* __restrict__ bn_s,
                                                              const bf16x8* __restrict__ wp,
                                                              float* __restrict__ dxe, int T, int Cy,
                                                              int Cred, int Cin, int Npad, int K, int dil,
                                                              float slope) {
  constexpr int PITCH = SA_TD_CK + 8, ROWS = SA_TD_BM + SA_TD_HALO, PLANE = ROWS * PITCH;
  __shared__ __attribute__((aligned(16))) bf16_t As[2 * PLANE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int t0 = blockIdx.x * SA_TD_BM, n0 = blockIdx.y * SA_TD_BN, b = blockIdx.z;
  const int pad = dil * (K - 1) / 2, nrows = SA_TD_BM + 2 * pad, Te = T + 2 * pad;
  const int KSTEPS = Cred / 16, NT = Npad / 32;
  const size_t lo_off = (size_t)K * KSTEPS * NT * 64;
  f32x16 acc[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.0f;
  const float* dyb = dy + (size_t)b * T * Cy;
  const unsigned char* mb = mask + (size_t)b * T * Cy;
  for (int c0 = 0; c0 < Cred; c0 += SA_TD_CK) {
    const int ck = Cred - c0 < SA_TD_CK ? Cred - c0 : SA_TD_CK, chunks = ck / 4;
    // ---- stage d z rows (extended output row t0 + r uses d z[t0 + r - 2*pad + k'*dil]): zero
    // outside the utterance, LeakyReLU / BatchNorm(eval) backward applied on the fly ----
    for (int e = tid; e < nrows * chunks; e += 256) {
      const int r = e / chunks, c = e % chunks, ch = c0 + c * 4;
      const int tt = t0 + r - 2 * pad;
      float f[4] = {0.f, 0.f, 0.f, 0.f};
      if (tt >= 0 && tt < T && ch < Cy) {
        const float4 g = *reinterpret_cast<const float4*>(dyb + (size_t)tt * Cy + ch);
        const uchar4 m4 = *reinterpret_cast<const uchar4*>(mb + (size_t)tt * Cy + ch);
        const float gg[4] = {g.x, g.y, g.z, g.w};
        const unsigned char mm[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) f[j] = gg[j] * (bn_s ? bn_s[ch + j] : 1.0f) * (mm[j] ? 1.0f : slope);
      }
      uint2 hi, lo;
      sa_split4(f, hi, lo);
      *reinterpret_cast<uint2*>(As + r * PITCH + c * 4) = hi;
      *reinterpret_cast<uint2*>(As + PLANE + r * PITCH + c * 4) = lo;
    }
    __syncthreads();
    const int ksteps = ck / 16;
    for (int k = 0; k < K; ++k) {
      for (int ks = 0; ks < ksteps; ++ks) {
        const bf16x8* wt = wp + (((size_t)(K - 1 - k) * KSTEPS + c0 / 16 + ks) * NT + n0 / 32 + wn * 2) * 64 + lane;
        bf16x8 bh[2], bl[2], ah[2], al[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) { bh[nt] = wt[nt * 64]; bl[nt] = wt[lo_off + nt * 64]; }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const bf16_t* ap = As + (wm * 64 + mt * 32 + (lane & 31) + k * dil) * PITCH + ks * 16 + 8 * (lane >> 5);
          ah[mt] = *reinterpret_cast<const bf16x8*>(ap);
          al[mt] = *reinterpret_cast<const bf16x8*>(ap + PLANE);
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
          }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int n = n0 + (wn * 2 + nt) * 32 + (lane & 31);
    if (n < Cin) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int t = t0 + wm * 64 + mt * 32 + sa_acc_row(i, lane);
          if (t < Te) dxe[((size_t)b * Te + t) * Cin + n] = acc[mt][nt][i];
        }
    }
  }
}

// dxe: [B][T + 2*pad][Cin] (pad = dil*(K-1)/2), caller-allocated; feed it to sa_tdnn_fold.
extern "C" int sa_tdnn_bwd_input(const float* dy, const unsigned char* mask, const float* bn_s,
                                 const void* wp, float* dxe, int B, int T, int Cy, int Cred, int Cin,
                                 int Npad, int K, int dil, float slope, void* stream) {
  if (!dy || !mask || !wp || !dxe || B <= 0 || T <= 0 || Cy <= 0 || Cin <= 0 || K < 1 || !(K & 1) || dil < 1 ||
      dil * (K - 1) / 2 >= T || dil * (K - 1) > SA_TD_HALO || Cred % 16 || Cred < Cy || Cy % 4 ||
      Npad % SA_TD_BN || Npad < Cin)
    return -22;
  const int Te = T + dil * (K - 1);
  dim3 grid(sa_div_up(Te, SA_TD_BM), Npad / SA_TD_BN, B);
  hipLaunchKernelGGL(sa_tdnn_bwd_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dy, mask,
                     bn_s, reinterpret_cast<const bf16x8*>(wp), dxe, T, Cy, Cred, Cin, Npad, K, dil,
                     slope);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// adjoint of the reflect padding: dx[j] = dxe[j + pad] + dxe[pad - j] (1 <= j <= pad)
//                                       + dxe[2(T-1) - j + pad] (T-1-pad <= j <= T-2)
__global__ void sa_tdnn_fold_kernel(const float* __restrict__ dxe, float* __restrict__ dx, int T, int C,
                                    int pad, size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C);
  const size_t bt = i / C;
  const int j = (int)(bt % T);
  const size_t b = bt / T;
  const float* e = dxe + (b * (size_t)(T + 2 * pad)) * C + c;
  float v = e[(size_t)(j + pad) * C];
  if (j >= 1 && j <= pad) v += e[(size_t)(pad - j) * C];
  if (j >= T - 1 - pad && j <= T - 2) v += e[(size_t)(2 * (T - 1) - j + pad) * C];
  dx[i] = v;
}

extern "C" int sa_tdnn_fold(const float* dxe, float* dx, int B, int T, int C, int pad, void* stream) {
  if (!dxe || !dx || B <= 0 || T <= 0 || C <= 0 || pad < 0 || pad >= T) return -22;
  const size_t total = (size_t)B * T * C;
  hipLaunchKernelGGL(sa_tdnn_fold_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), dxe, dx, T, C, pad, total);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// backward of sa_time_pool (statistics over the first n = round(len*T) frames):
//   dx[t][c] = g_mean[c]/n + g_std[c] * (x[t][c] - mean[c]) / ((n-1) * std[c])   for t < n, else 0
// pooled = the forward output [B][2C] WITHOUT the noise offset on the mean half (mean, std + eps).
__global__ void sa_time_pool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ lens,
                                        const float* __restrict__ g, const float* __restrict__ pooled,
                                        int T, int C, float eps, float* __restrict__ dx, size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C);
  const size_t bt = i / C;
  const int t = (int)(bt % T);
  const size_t b = bt / T;
  int n = lens ? (int)rintf(lens[b] * (float)T) : T;
  if (n > T) n = T;
  float v = 0.0f;
  if (t < n) {
    const float m = pooled[b * 2 * C + c], sd = pooled[b * 2 * C + C + c] - eps;
    v = g[b * 2 * C + c] / (float)n;
    if (n > 1 && sd > 0.0f) v += g[b * 2 * C + C + c] * (x[i] - m) / ((float)(n - 1) * sd);
  }
  dx[i] = v;
}

extern "C" int sa_time_pool_bwd(const float* x, const float* lens, const float* g, const float* pooled,
                                int B, int T, int C, float eps, float* dx, void* stream) {
  if (!x || !g || !pooled || !dx || B <= 0 || T <= 0 || C <= 0) return -22;
  const size_t total = (size_t)B * T * C;
  hipLaunchKernelGGL(sa_time_pool_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, lens, g, pooled, T, C, eps, dx, total);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// backward of sa_leaky_affine: dx = dy * s[c] * (x > 0 ? 1 : slope)
__global__ void sa_leaky_affine_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                           const float* __restrict__ s, float slope, int M, int C, float* dx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * C) return;
  dx[i] = dy[i] * (s ? s[i % C] : 1.0f) * (x[i] > 0.0f ? 1.0f : slope);
}

extern "C" int sa_leaky_affine_bwd(const float* dy, const float* x, const float* s, float slope, int M, int C,
                                   float* dx, void* stream) {
  if (!dy || !x || !dx) return -22;
  hipLaunchKernelGGL(sa_leaky_affine_bwd_kernel, dim3(sa_div_up(M * C, 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), dy, x, s, slope, M, C, dx);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// speechbrain StatisticsPooling over time with relative lengths: mean and unbiased std (+eps) of
// the first round(len*T) frames per (utterance, channel); out [B][2C] = (mean (+noise), std).
__global__ void sa_time_pool_kernel(const float* __restrict__ x, const float* __restrict__ lens,
                                    const float* __restrict__ noise, int T, int C, float eps,
                                    float* __restrict__ out) {
  const int b = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  int n = lens ? (int)rintf(lens[b] * (float)T) : T;
  if (n > T) n = T;
  double s = 0.0, q = 0.0;
  for (int t = 0; t < n; ++t) {
    const double v = x[((size_t)b * T + t) * C + c];
    s += v; q += v * v;
  }
  const double m = s / n;
  double var = n > 1 ? (q - s * m) / (n - 1) : 0.0;
  if (var < 0.0) var = 0.0;
  float mo = (float)m;
  if (noise) mo += eps * ((1.0f - 9.0f) * noise[(size_t)b * C + c] + 9.0f);
  out[(size_t)b * 2 * C + c] = mo;
  out[(size_t)b * 2 * C + C + c] = (float)sqrt(var) + eps;
}

extern "C" int sa_time_pool(const float* x, const float* lens, const float* noise, int B, int T, int C,
                            float eps, float* out, void* stream) {
  if (!x || !out || B <= 0 || T <= 0 || C <= 0) return -22;
  hipLaunchKernelGGL(sa_time_pool_kernel, dim3(sa_div_up(C, 128), B), dim3(128), 0,
                     reinterpret_cast<hipStream_t>(stream), x, lens, noise, T, C, eps, out);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// y = x > 0 ? x : slope*x, then *s[c] + t[c]  (LeakyReLU -> BatchNorm(eval) on a small [M][C])
__global__ void sa_leaky_affine_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                       const float* __restrict__ t, float slope, int M, int C, float* y) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * C) return;
  float v = x[i];
  v = v > 0.0f ? v : v * slope;
  y[i] = s ? fmaf(v, s[i % C], t[i % C]) : v;
}

extern "C" int sa_leaky_affine(const float* x, const float* s, const float* t, float slope, int M, int C,
                               float* y, void* stream) {
  if (!x || !y) return -22;
  hipLaunchKernelGGL(sa_leaky_affine_kernel, dim3(sa_div_up(M * C, 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, s, t, slope, M, C, y);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}
