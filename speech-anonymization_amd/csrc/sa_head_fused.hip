// The FC head of the sex classifier as ONE forward and ONE backward launch (models/ConvAutoEncoder.py:47-55
// `classify` = Linear(256,128) -> ReLU -> BatchNorm1d(128) -> Linear(128,64) -> ReLU -> BatchNorm1d(64) ->
// Linear(64,2), :68 log_softmax; backward via speechbrain_convae_train.py:241).
//
// The head works on [B, 256] pooled rows: a few hundred kFLOP.  As separate launches (3 dense, 2 column
// sums, 2 finalisers, log-softmax forward; ~25 launches backward) every one of them is a 5 us kernel
// behind a 1.5 us boundary on the step's critical path -- 37 launches, ~0.25 ms of a 8.7 ms step at
// B = 32 and of a 3.8 ms step at B = 10.  Here one workgroup walks the whole chain with the
// activations in LDS: train-mode BatchNorm needs the statistics of ALL rows between two layers, which
// is a workgroup barrier in one launch and a kernel boundary otherwise.  Forward: the two wide layers on
// the exact-fp32 MFMA with LDS-staged operands; backward: fp32 FMA loops (its products are M-deep).  fp64
// statistics like the separate finalisers.  The launches of sa_head.hip stay for B > SA_HEAD_MAXB, eval
// mode and SyncBatchNorm, where the sums are all-reduced between the layers.
#include "sa_common.h"

#define SA_HEAD_MAXB 64

namespace {

constexpr int K0 = 256, N1 = 128, N2 = 64, NC = 2;

// train-mode BatchNorm statistics of column n of H [M][N] (LDS): the operations of sa_colsums +
// sa_fin_bn_fwd (fp64 sums; running statistics with the unbiased variance)
__device__ __forceinline__ void head_bn_fwd(const float* H, int M, int N, int n, const float* gamma, const float* beta,
                                            float eps, float momentum, float* run_mean, float* run_var,
                                            float* f /* [4][N] global */, float* s_lds, float* t_lds) {
  double S = 0.0, Q = 0.0;
  for (int m = 0; m < M; ++m) { const float x = H[m * N + n]; S += x; Q += (double)x * x; }
  const double cnt = (double)M, mu = S / cnt;
  double var = Q / cnt - mu * mu;
  if (var < 0.0) var = 0.0;
  const float r = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = gamma[n] * r, sh = beta[n] - (float)mu * sc;
  f[n] = (float)mu; f[N + n] = r; f[2 * N + n] = sc; f[3 * N + n] = sh;
  s_lds[n] = sc; t_lds[n] = sh;
  if (run_mean) {
    const double unb = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
    run_mean[n] = (1.0f - momentum) * run_mean[n] + momentum * (float)mu;
    run_var[n] = (1.0f - momentum) * run_var[n] + momentum * (float)unb;
  }
}

}  // namespace

struct SaHeadFwdArgs {
  const float* pooled;                                     // [M][256]
  const float *w1, *b1, *g1, *be1; float *rm1, *rv1;        // Linear(256,128), BatchNorm1d(128)
  const float *w2, *b2, *g2, *be2; float *rm2, *rv2;        // Linear(128,64), BatchNorm1d(64)
  const float *w3, *b3;                                    // Linear(64,2)
  float *h1, *f1, *h2, *f2, *logp;                         // outputs: [M][128], [4][128], [M][64], [4][64], [M][2]
  int M; float eps, momentum;
};

// GEMM on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32, the FC head's matrix core path) with both
// operands staged in LDS: out[m][n] += sum_{k < KC} A[m][k] * W[n][k].  A, W: LDS tiles with a pitch of
// PITCH floats (33 sixteen-byte slots: an odd slot pitch makes the ds_read_b128 of 32 consecutive rows
// conflict-free).  The reduction index is permuted so that every lane reads four CONSECUTIVE k of its
// row per load: lanes 0..31 walk k in [0, KC/2), lanes 32..63 k in [KC/2, KC) -- the same map for both
// operands, which is all a sum over k needs.
constexpr int PITCH = 132;
template <int KC>
__device__ __forceinline__ void head_mfma_tile(f32x16& acc, const float* At, const float* Wt, int lane) {
  const float* ap = At + (lane & 31) * PITCH + (lane >> 5) * (KC / 2);
  const float* wp = Wt + (lane & 31) * PITCH + (lane >> 5) * (KC / 2);
#pragma unroll 4
  for (int j = 0; j < KC / 8; ++j) {
    const float4 av = *reinterpret_cast<const float4*>(ap + 4 * j), bv = *reinterpret_cast<const float4*>(wp + 4 * j);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
  }
}

// stage `rows` rows of KC floats (global row pitch ld, first column k0) into an LDS tile; rows >= nvalid are
// zero; xf(v, k): per-column transform (the BatchNorm affine of the previous layer)
template <int KC, class XF>
__device__ __forceinline__ void head_stage(float* tile, const float* src, int ld, int k0, int rows, int nvalid, int tid,
                                           XF xf) {
  for (int i = tid; i < rows * (KC / 4); i += 256) {
    const int r = i / (KC / 4), q = i % (KC / 4);
    float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (r < nvalid) v = xf(*reinterpret_cast<const float4*>(src + (size_t)r * ld + k0 + 4 * q), k0 + 4 * q);
    *reinterpret_cast<float4*>(tile + r * PITCH + 4 * q) = v;
  }
}

__global__ __launch_bounds__(256) void sa_head_fwd_kernel(SaHeadFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int M = a.M, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int MT = (M + 31) / 32;                             // 32-row tiles (1 or 2)
  float* At = reinterpret_cast<float*>(smem);               // [64][PITCH] A operand tile
  float* Wt = At + 64 * PITCH;                              // [128][PITCH] weight tile
  float* H1 = Wt + 128 * PITCH;                             // [M][128] relu(Linear 1)
  float* H2 = H1 + M * N1;                                  // [M][64]
  float* s1 = H2 + M * N2; float* t1 = s1 + N1; float* s2 = t1 + N1; float* t2 = s2 + N2;
  float* lg = t2 + N2;                                      // [M][2] logits
  auto ident = [](float4 v, int) { return v; };
  // ---- H1 = relu(pooled W1^T + b1): 4 column tiles x MT row tiles, K = 256 in two chunks of 128 ----
  {
    f32x16 acc[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mt][i] = 0.0f;
    for (int c = 0; c < 2; ++c) {
      head_stage<128>(At, a.pooled, K0, c * 128, 32 * MT, M, tid, ident);
      head_stage<128>(Wt, a.w1, K0, c * 128, N1, N1, tid, ident);
      __syncthreads();
      head_mfma_tile<128>(acc[0], At, Wt + wave * 32 * PITCH, lane);
      if (MT > 1) head_mfma_tile<128>(acc[1], At + 32 * PITCH, Wt + wave * 32 * PITCH, lane);
      __syncthreads();
    }
    const int n = wave * 32 + (lane & 31);
    const float bb = a.b1[n];
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = mt * 32 + sa_acc_row(i, lane);
        if (m < M) { const float v = fmaxf(acc[mt][i] + bb, 0.0f); H1[m * N1 + n] = v; a.h1[(size_t)m * N1 + n] = v; }
      }
  }
  __syncthreads();
  if (tid < N1) head_bn_fwd(H1, M, N1, tid, a.g1, a.be1, a.eps, a.momentum, a.rm1, a.rv1, a.f1, s1, t1);
  __syncthreads();
  // ---- H2 = relu(bn1(H1) W2^T + b2): 2 column tiles x MT row tiles over the four waves, K = 128 ----
  {
    head_stage<128>(At, H1, N1, 0, 32 * MT, M, tid, [&](float4 v, int k) {
      return make_float4(fmaf(v.x, s1[k], t1[k]), fmaf(v.y, s1[k + 1], t1[k + 1]), fmaf(v.z, s1[k + 2], t1[k + 2]),
                         fmaf(v.w, s1[k + 3], t1[k + 3]));
    });
    head_stage<128>(Wt, a.w2, N1, 0, N2, N2, tid, ident);
    __syncthreads();
    const int nt = wave & 1, mt = wave >> 1;                // (waves 2, 3: the second row tile)
    if (mt < MT) {
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
      head_mfma_tile<128>(acc, At + mt * 32 * PITCH, Wt + nt * 32 * PITCH, lane);
      const int n = nt * 32 + (lane & 31);
      const float bb = a.b2[n];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = mt * 32 + sa_acc_row(i, lane);
        if (m < M) { const float v = fmaxf(acc[i] + bb, 0.0f); H2[m * N2 + n] = v; a.h2[(size_t)m * N2 + n] = v; }
      }
    }
  }
  __syncthreads();
  if (tid < N2) head_bn_fwd(H2, M, N2, tid, a.g2, a.be2, a.eps, a.momentum, a.rm2, a.rv2, a.f2, s2, t2);
  __syncthreads();
  if (tid < M * NC) {                                       // logits = bn2(H2) W3^T + b3
    const int m = tid / NC, c = tid % NC;
    float acc = 0.0f;
    for (int k = 0; k < N2; ++k) acc = fmaf(fmaf(H2[m * N2 + k], s2[k], t2[k]), a.w3[c * N2 + k], acc);
    lg[tid] = acc + a.b3[c];
  }
  __syncthreads();
  if (tid < M * NC) {                                       // log_softmax over the two classes (sa_log_softmax's operations)
    const int m = tid / NC;
    const float x0 = lg[m * NC], x1 = lg[m * NC + 1], mx = fmaxf(x0, x1);
    const float lse = mx + logf(expf(x0 - mx) + expf(x1 - mx));
    a.logp[tid] = lg[tid] - lse;
  }
}

extern "C" int sa_head_fwd(const float* pooled, const float* w1, const float* b1, const float* g1, const float* be1,
                           float* rm1, float* rv1, const float* w2, const float* b2, const float* g2,
                           const float* be2, float* rm2, float* rv2, const float* w3, const float* b3,
                           float* h1, float* f1, float* h2, float* f2, float* logp, int M, float eps,
                           float momentum, void* stream) {
  if (!pooled || !w1 || !b1 || !g1 || !be1 || !w2 || !b2 || !g2 || !be2 || !w3 || !b3 || !h1 || !f1 || !h2 || !f2 ||
      !logp || M < 1 || M > SA_HEAD_MAXB)
    return -22;
  SaHeadFwdArgs a{pooled, w1, b1, g1, be1, rm1, rv1, w2, b2, g2, be2, rm2, rv2, w3, b3, h1, f1, h2, f2, logp, M, eps, momentum};
  const size_t lds = ((size_t)(64 + 128) * PITCH + (size_t)M * (N1 + N2 + NC) + 2 * (N1 + N2)) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sa_head_fwd_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return -(int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(sa_head_fwd_kernel, dim3(1), dim3(256), lds, reinterpret_cast<hipStream_t>(stream), a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

struct SaHeadBwdArgs {
  const float *dlogp, *logp, *pooled, *h1, *f1, *h2, *f2;
  const float *w1, *g1, *w2, *g2, *w3;
  float *dw1, *db1, *dg1, *dbe1, *dw2, *db2, *dg2, *dbe2, *dw3, *db3, *dpooled;
  int M;
};

// BatchNorm backward of column n over the M rows: sums (fp64) of dN and dN * hhat -> d gamma, d beta, and
// dH = gamma*rstd*(dN - S1/M - hhat*S2/M) * (H > 0): sa_colsums + sa_bn2d_bwd; dN [M][N] in LDS is
// overwritten with dH
__device__ __forceinline__ void head_bn_bwd(float* dN, const float* H, int M, int N, int n, const float* f,
                                            const float* gamma, float* dgamma, float* dbeta) {
  const float mean = f[n], rstd = f[N + n];
  double S1 = 0.0, S2 = 0.0;
  for (int m = 0; m < M; ++m) {
    const float g = dN[m * N + n], hh = (H[m * N + n] - mean) * rstd;
    S1 += g; S2 += (double)g * hh;
  }
  if (dbeta) dbeta[n] = (float)S1;
  if (dgamma) dgamma[n] = (float)S2;
  const float a1 = (float)(S1 / (double)M), a2 = (float)(S2 / (double)M), c = gamma[n] * rstd;
  for (int m = 0; m < M; ++m) {
    const float h = H[m * N + n], hh = (h - mean) * rstd;
    float v = c * (dN[m * N + n] - a1 - hh * a2);
    if (!(h > 0.0f)) v = 0.0f;
    dN[m * N + n] = v;
  }
}

__global__ __launch_bounds__(1024) void sa_head_bwd_kernel(SaHeadBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int M = a.M, tid = threadIdx.x;
  const float* X = a.pooled;                                // [M][256]: read in place (L1 / L2), 64 KB of LDS saved
  float* H1 = reinterpret_cast<float*>(smem);               // [M][128] forward activation (post-ReLU)
  float* H2 = H1 + M * N1;                                  // [M][64]
  float* D1 = H2 + M * N2;                                  // [M][128] d N1 -> d H1
  float* D2 = D1 + M * N1;                                  // [M][64]  d N2 -> d H2
  float* DL = D2 + M * N2;                                  // [M][2]   d logits
  for (int i = tid; i < M * N1 / 4; i += 1024) reinterpret_cast<float4*>(H1)[i] = reinterpret_cast<const float4*>(a.h1)[i];
  for (int i = tid; i < M * N2 / 4; i += 1024) reinterpret_cast<float4*>(H2)[i] = reinterpret_cast<const float4*>(a.h2)[i];
  if (tid < M) {                                            // d logits = d logp - exp(logp) * sum_c d logp
    const float d0 = a.dlogp[tid * NC], d1 = a.dlogp[tid * NC + 1], s = d0 + d1;
    DL[tid * NC] = d0 - expf(a.logp[tid * NC]) * s;
    DL[tid * NC + 1] = d1 - expf(a.logp[tid * NC + 1]) * s;
  }
  __syncthreads();
  const float* s2 = a.f2 + 2 * N2; const float* t2 = a.f2 + 3 * N2;
  const float* s1 = a.f1 + 2 * N1; const float* t1 = a.f1 + 3 * N1;
  // d W3[c][k] = sum_m dL[m][c] * bn2(H2)[m][k]; d b3[c] = sum_m dL[m][c]; d N2[m][k] = sum_c dL[m][c] W3[c][k]
  if (tid < NC * N2) {
    const int c = tid / N2, k = tid % N2;
    float acc = 0.0f;
    for (int m = 0; m < M; ++m) acc = fmaf(DL[m * NC + c], fmaf(H2[m * N2 + k], s2[k], t2[k]), acc);
    if (a.dw3) a.dw3[tid] = acc;
  } else if (tid < NC * N2 + NC) {
    const int c = tid - NC * N2;
    float acc = 0.0f;
    for (int m = 0; m < M; ++m) acc += DL[m * NC + c];
    if (a.db3) a.db3[c] = acc;
  }
  for (int i = tid; i < M * N2; i += 1024) {
    const int m = i / N2, k = i % N2;
    D2[i] = fmaf(DL[m * NC], a.w3[k], DL[m * NC + 1] * a.w3[N2 + k]);
  }
  __syncthreads();
  if (tid < N2) head_bn_bwd(D2, H2, M, N2, tid, a.f2, a.g2, a.dg2, a.dbe2);
  __syncthreads();
  // d W2[n][k] = sum_m dH2[m][n] * bn1(H1)[m][k] (64 x 128 outputs); d b2[n] = sum_m dH2[m][n]
  for (int i = tid; i < N2 * N1; i += 1024) {
    const int n = i / N1, k = i % N1;
    float acc = 0.0f;
    const float sk = s1[k], tk = t1[k];
    for (int m = 0; m < M; ++m) acc = fmaf(D2[m * N2 + n], fmaf(H1[m * N1 + k], sk, tk), acc);
    if (a.dw2) a.dw2[i] = acc;
  }
  if (tid < N2 && a.db2) {
    float acc = 0.0f;
    for (int m = 0; m < M; ++m) acc += D2[m * N2 + tid];
    a.db2[tid] = acc;
  }
  // d N1[m][k] = sum_n dH2[m][n] * W2[n][k]
  for (int i = tid; i < M * N1; i += 1024) {
    const int m = i / N1, k = i % N1;
    float acc = 0.0f;
#pragma unroll 8
    for (int n = 0; n < N2; ++n) acc = fmaf(D2[m * N2 + n], a.w2[n * N1 + k], acc);
    D1[i] = acc;
  }
  __syncthreads();
  if (tid < N1) head_bn_bwd(D1, H1, M, N1, tid, a.f1, a.g1, a.dg1, a.dbe1);
  __syncthreads();
  // d W1[n][k] = sum_m dH1[m][n] * pooled[m][k] (128 x 256 outputs, 32 per thread: 8 n x 4 k);
  // d b1[n] = sum_m dH1[m][n]
  {
    const int kq = tid & 63, ng = tid >> 6;                 // k = 4*kq .. +3, n = 8*ng .. +7
    float acc[8][4];
#pragma unroll
    for (int r = 0; r < 8; ++r) { acc[r][0] = acc[r][1] = acc[r][2] = acc[r][3] = 0.0f; }
    for (int m = 0; m < M; ++m) {
      const float4 x = *reinterpret_cast<const float4*>(X + m * K0 + 4 * kq);
      const float4 d0 = *reinterpret_cast<const float4*>(D1 + m * N1 + 8 * ng);
      const float4 d1 = *reinterpret_cast<const float4*>(D1 + m * N1 + 8 * ng + 4);
      const float dv[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        acc[r][0] = fmaf(dv[r], x.x, acc[r][0]); acc[r][1] = fmaf(dv[r], x.y, acc[r][1]);
        acc[r][2] = fmaf(dv[r], x.z, acc[r][2]); acc[r][3] = fmaf(dv[r], x.w, acc[r][3]);
      }
    }
    if (a.dw1) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
        *reinterpret_cast<float4*>(a.dw1 + (size_t)(8 * ng + r) * K0 + 4 * kq) = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
    }
  }
  if (tid < N1 && a.db1) {
    float acc = 0.0f;
    for (int m = 0; m < M; ++m) acc += D1[m * N1 + tid];
    a.db1[tid] = acc;
  }
  // d pooled[m][k] = sum_n dH1[m][n] * W1[n][k]: thread = (k quad, row group)
  if (a.dpooled) {
    const int kq = tid & 63, m0 = tid >> 6;                 // 16 row groups
    float acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { acc[r][0] = acc[r][1] = acc[r][2] = acc[r][3] = 0.0f; }
#pragma unroll 2
    for (int n = 0; n < N1; ++n) {
      const float4 w = *reinterpret_cast<const float4*>(a.w1 + (size_t)n * K0 + 4 * kq);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + 16 * r;
        const float d = m < M ? D1[m * N1 + n] : 0.0f;
        acc[r][0] = fmaf(d, w.x, acc[r][0]); acc[r][1] = fmaf(d, w.y, acc[r][1]);
        acc[r][2] = fmaf(d, w.z, acc[r][2]); acc[r][3] = fmaf(d, w.w, acc[r][3]);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + 16 * r;
      if (m < M) *reinterpret_cast<float4*>(a.dpooled + (size_t)m * K0 + 4 * kq) = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
    }
  }
}

extern "C" int sa_head_bwd(const float* dlogp, const float* logp, const float* pooled, const float* h1,
                           const float* f1, const float* h2, const float* f2, const float* w1, const float* g1,
                           const float* w2, const float* g2, const float* w3, float* dw1, float* db1, float* dg1,
                           float* dbe1, float* dw2, float* db2, float* dg2, float* dbe2, float* dw3, float* db3,
                           float* dpooled, int M, void* stream) {
  if (!dlogp || !logp || !pooled || !h1 || !f1 || !h2 || !f2 || !w1 || !g1 || !w2 || !g2 || !w3 || M < 1 ||
      M > SA_HEAD_MAXB)
    return -22;
  SaHeadBwdArgs a{dlogp, logp, pooled, h1, f1, h2, f2, w1, g1, w2, g2, w3,
                  dw1, db1, dg1, dbe1, dw2, db2, dg2, dbe2, dw3, db3, dpooled, M};
  const size_t lds = (size_t)M * (2 * N1 + 2 * N2 + NC) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sa_head_bwd_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return -(int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(sa_head_bwd_kernel, dim3(1), dim3(1024), lds, reinterpret_cast<hipStream_t>(stream), a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

extern "C" int sa_head_max_rows(void) { return SA_HEAD_MAXB; }
