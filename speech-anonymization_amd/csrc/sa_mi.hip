// k-NN mutual-information estimate between a continuous batch and class labels
// (Ross 2014), as utils/ClusterMI.py:88-121 computes it, batched over the resampled index
// sets of utils/GroupSamplingMI.py:49-61 (one workgroup per resampling iteration).
//   d[i][j] = 1 - cos(x_i, x_j)   (eps 1e-8; diagonal 0)              ClusterMI.py:9-10,12-65
//   anchor_i = (k+1)-th smallest same-class distance (self included)   :108-112
//   m_i = #{j : d[i][j] <= anchor_i} - 1                               :115
//   MI = (psi(N) - sum_c N_c/N psi(N_c) + psi(k) - mean_i psi(m_i)) / ln 2      :120-121
// The estimate is non-differentiable (counts + digamma): forward only, like the reference.
#include "sa_common.h"

__device__ static inline double sa_digamma(double x) {
  if (x == 0.0) return -INFINITY;                 // torch.digamma(0) = -inf
  double r = 0.0;
  while (x < 6.0) { r -= 1.0 / x; x += 1.0; }
  const double f = 1.0 / (x * x);
  return r + log(x) - 0.5 / x - f * (1.0 / 12.0 - f * (1.0 / 120.0 - f * (1.0 / 252.0 - f * (1.0 / 240.0))));
}

#define SA_MI_MAXK 8

__global__ __launch_bounds__(256) void sa_cluster_mi_kernel(const float* __restrict__ X,
                                                            const long long* __restrict__ y,
                                                            const long long* __restrict__ idx,
                                                            int n, int D, int ncls, int k,
                                                            float* __restrict__ mi) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* dist = reinterpret_cast<float*>(smem);          // [n][n]
  float* nrm = dist + (size_t)n * n;                     // [n]
  int* lab = reinterpret_cast<int*>(nrm + n);            // [n]
  int* row = lab + n;                                    // [n] source row of each sample
  double* psum = reinterpret_cast<double*>(row + n);   // [256]; n*(n+3) ints is even -> 8-byte aligned
  const int tid = threadIdx.x, it = blockIdx.x;
  for (int i = tid; i < n; i += 256) {
    const int r = idx ? (int)idx[(size_t)it * n + i] : i;
    row[i] = r; lab[i] = (int)y[r];
  }
  __syncthreads();
  for (int i = tid; i < n; i += 256) {
    const float* a = X + (size_t)row[i] * D;
    float s = 0.f;
    for (int t = 0; t < D; ++t) s = fmaf(a[t], a[t], s);
    nrm[i] = s;
  }
  __syncthreads();
  for (int p = tid; p < n * n; p += 256) {
    const int i = p / n, j = p % n;
    float d = 0.0f;
    if (i != j) {
      const int lo = i < j ? i : j, hi = i < j ? j : i;    // identical arithmetic for (i,j),(j,i)
      const float* a = X + (size_t)row[lo] * D;
      const float* b = X + (size_t)row[hi] * D;
      float dot = 0.f;
      for (int t = 0; t < D; ++t) dot = fmaf(a[t], b[t], dot);
      d = 1.0f - dot / sqrtf(fmaxf(nrm[lo] * nrm[hi], 1e-16f));
    }
    dist[p] = d;
  }
  __syncthreads();
  double acc = 0.0;
  for (int i = tid; i < n; i += 256) {
    float best[SA_MI_MAXK + 1];
    for (int q = 0; q <= k; ++q) best[q] = 10e6f;
    for (int j = 0; j < n; ++j) {
      float d = lab[j] == lab[i] ? dist[(size_t)i * n + j] : 10e6f;
      for (int q = 0; q <= k; ++q)
        if (d < best[q]) { const float t = best[q]; best[q] = d; d = t; }
    }
    const float anchor = best[k];
    int m = -1;
    for (int j = 0; j < n; ++j) m += dist[(size_t)i * n + j] <= anchor ? 1 : 0;
    acc += sa_digamma((double)m);
  }
  psum[tid] = acc;
  __syncthreads();
  if (tid == 0) {
    double s = 0.0;
    for (int i = 0; i < 256; ++i) s += psum[i];
    double avg_nx = 0.0;
    for (int c = 0; c < ncls; ++c) {
      int cnt = 0;
      for (int i = 0; i < n; ++i) cnt += lab[i] == c;
      avg_nx += (double)cnt / n * sa_digamma((double)cnt);
    }
    const double v = sa_digamma((double)n) - avg_nx + sa_digamma((double)k) - s / n;
    mi[it] = (float)(v / log(2.0));
  }
}

// X [N][D] fp32, y [N] int64, idx [iters][n] int64 (or null: one iteration over rows 0..n-1)
extern "C" int sa_cluster_mi(const float* X, const long long* y, const long long* idx, int iters,
                             int n, int D, int ncls, int k, float* mi, void* stream) {
  if (!X || !y || !mi || iters <= 0 || n <= k || n > 160 || k < 1 || k > SA_MI_MAXK) return -22;
  const size_t lds = ((size_t)n * n + n) * sizeof(float) + (size_t)(2 * n) * sizeof(int)
                     + 256 * sizeof(double) + 16;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sa_cluster_mi_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    if (e != hipSuccess) return -(int)e;
    attr = true;
  }
  hipLaunchKernelGGL(sa_cluster_mi_kernel, dim3(iters), dim3(256), lds,
                     reinterpret_cast<hipStream_t>(stream), X, y, idx, n, D, ncls, k, mi);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}
