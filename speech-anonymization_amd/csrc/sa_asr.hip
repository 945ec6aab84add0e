// Element-wise passes around the GEMMs of the frozen recogniser (models/SpeechBrain_ASR.py:16-30,
// speechbrain_configs/convae.yaml:139-158 -- ConvolutionFrontEnd + post-norm TransformerASR, d_model 768).
// The GEMMs are library calls (hipBLASLt through torch); what surrounds them was ~30 % of the
// branch as separate torch launches.  Here, bf16 storage / fp32 arithmetic, HBM-bound by design:
//   sa_add_layernorm_fwd   s = bf16(x + r);  y = LayerNorm(s) * gamma + beta      (one pass: x, r in; y, s out)
//   sa_layernorm_bwd       d s from d y, s, (mean, rstd)                            (one pass; d s is the
//                          gradient of BOTH addends -- the residual add has no backward kernel)
//   sa_reflect_pad_fwd/bwd "same" padding of the front end's 3x3 convolutions on [B][T][F][C] rows:
//                          one gather each way (torch: two concatenations forward, four narrow adds back)
// One wave per row for the LayerNorm (d = 768: three 8-byte chunks per lane, the statistics by DPP /
// permute reductions inside the wave, no LDS, no barrier).
#include "sa_common.h"

namespace {

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ inline void unpack4(const uint2& u, float* f) {
  f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
  f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
}
__device__ inline uint2 pack4(const float* f) {
  return make_uint2(sa_pack_bf16x2(f[0], f[1]), sa_pack_bf16x2(f[2], f[3]));
}

// NCH chunks of 256 elements per row (d = 256 * NCH); lane l holds elements [256 j + 4 l, +4)
template <int NCH>
__global__ __launch_bounds__(256) void add_ln_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ r,
                                                         const bf16_t* __restrict__ gamma,
                                                         const bf16_t* __restrict__ beta, bf16_t* __restrict__ y,
                                                         bf16_t* __restrict__ s_out, float* __restrict__ stat,
                                                         int rows, float eps) {
  constexpr int D = 256 * NCH;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const size_t base = (size_t)row * D + lane * 4;
  float v[NCH][4];
  uint2 xv[NCH], rv[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) xv[j] = *reinterpret_cast<const uint2*>(x + base + 256 * j);
  if (r) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) rv[j] = *reinterpret_cast<const uint2*>(r + base + 256 * j);
  }
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    unpack4(xv[j], v[j]);
    if (r) {
      float t[4];
      unpack4(rv[j], t);
      // the sum is rounded to bf16 first: the normalisation (and its backward, from the stored s)
      // sees exactly the tensor an unfused bf16 add would have produced
      const uint2 p = pack4((float[4]){v[j][0] + t[0], v[j][1] + t[1], v[j][2] + t[2], v[j][3] + t[3]});
      if (s_out) *reinterpret_cast<uint2*>(s_out + base + 256 * j) = p;
      unpack4(p, v[j]);
    } else if (s_out) {
      *reinterpret_cast<uint2*>(s_out + base + 256 * j) = xv[j];
    }
    sum += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
  }
  const float mean = wave_sum(sum) * (1.0f / D);
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < NCH; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[j][i] -= mean;
      sq = fmaf(v[j][i], v[j][i], sq);
    }
  const float rstd = rsqrtf(wave_sum(sq) * (1.0f / D) + eps);
  if (stat && lane == 0) {
    stat[2 * row] = mean;
    stat[2 * row + 1] = rstd;
  }
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    float g[4], b[4], o[4];
    unpack4(*reinterpret_cast<const uint2*>(gamma + lane * 4 + 256 * j), g);
    unpack4(*reinterpret_cast<const uint2*>(beta + lane * 4 + 256 * j), b);
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = fmaf(v[j][i] * rstd, g[i], b[i]);
    *reinterpret_cast<uint2*>(y + base + 256 * j) = pack4(o);
  }
}

// d s = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = d y * gamma,  xhat = (s - mean) * rstd
template <int NCH>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ s,
                                                     const float* __restrict__ stat,
                                                     const bf16_t* __restrict__ gamma, bf16_t* __restrict__ ds,
                                                     int rows) {
  constexpr int D = 256 * NCH;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const size_t base = (size_t)row * D + lane * 4;
  const float mean = stat[2 * row], rstd = stat[2 * row + 1];
  float g[NCH][4], xh[NCH][4];
  uint2 dv[NCH], sv[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    dv[j] = *reinterpret_cast<const uint2*>(dy + base + 256 * j);
    sv[j] = *reinterpret_cast<const uint2*>(s + base + 256 * j);
  }
  float sg = 0.f, sgx = 0.f;
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    float gm[4];
    unpack4(dv[j], g[j]);
    unpack4(sv[j], xh[j]);
    unpack4(*reinterpret_cast<const uint2*>(gamma + lane * 4 + 256 * j), gm);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      g[j][i] *= gm[i];
      xh[j][i] = (xh[j][i] - mean) * rstd;
      sg += g[j][i];
      sgx = fmaf(g[j][i], xh[j][i], sgx);
    }
  }
  const float mg = wave_sum(sg) * (1.0f / D), mgx = wave_sum(sgx) * (1.0f / D);
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    float o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = rstd * (g[j][i] - mg - xh[j][i] * mgx);
    *reinterpret_cast<uint2*>(ds + base + 256 * j) = pack4(o);
  }
}

// ---- front end: LayerNorm over (frequency, channel) + LeakyReLU, rows of d = 1024 * NCH elements, one
// 256-thread workgroup per row (thread t holds elements [1024 j + 4 t, +4))
__device__ inline float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();                       // (red is reused by the next reduction)
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

template <int NCH>
__global__ __launch_bounds__(256) void ln_leaky_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ gamma,
                                                           const bf16_t* __restrict__ beta, bf16_t* __restrict__ y,
                                                           float* __restrict__ stat, float eps, float slope) {
  constexpr int D = 1024 * NCH;
  __shared__ float red[4];
  const size_t base = (size_t)blockIdx.x * D + threadIdx.x * 4;
  float v[NCH][4];
  uint2 xv[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) xv[j] = *reinterpret_cast<const uint2*>(x + base + 1024 * j);
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    unpack4(xv[j], v[j]);
    sum += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
  }
  const float mean = block_sum(sum, red) * (1.0f / D);
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < NCH; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[j][i] -= mean;
      sq = fmaf(v[j][i], v[j][i], sq);
    }
  const float rstd = rsqrtf(block_sum(sq, red) * (1.0f / D) + eps);
  if (stat && threadIdx.x == 0) {
    stat[2 * blockIdx.x] = mean;
    stat[2 * blockIdx.x + 1] = rstd;
  }
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    float g[4], b[4], o[4];
    unpack4(*reinterpret_cast<const uint2*>(gamma + threadIdx.x * 4 + 1024 * j), g);
    unpack4(*reinterpret_cast<const uint2*>(beta + threadIdx.x * 4 + 1024 * j), b);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float z = fmaf(v[j][i] * rstd, g[i], b[i]);
      o[i] = z > 0.f ? z : z * slope;
    }
    *reinterpret_cast<uint2*>(y + base + 1024 * j) = pack4(o);
  }
}

// d x of y = leaky(LayerNorm(x) * gamma + beta): the activation's branch is recomputed from x and the statistics
template <int NCH>
__global__ __launch_bounds__(256) void ln_leaky_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                           const float* __restrict__ stat,
                                                           const bf16_t* __restrict__ gamma,
                                                           const bf16_t* __restrict__ beta, bf16_t* __restrict__ dx,
                                                           float slope) {
  constexpr int D = 1024 * NCH;
  __shared__ float red[4];
  const size_t base = (size_t)blockIdx.x * D + threadIdx.x * 4;
  const float mean = stat[2 * blockIdx.x], rstd = stat[2 * blockIdx.x + 1];
  float g[NCH][4], xh[NCH][4];
  uint2 dv[NCH], sv[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    dv[j] = *reinterpret_cast<const uint2*>(dy + base + 1024 * j);
    sv[j] = *reinterpret_cast<const uint2*>(x + base + 1024 * j);
  }
  float sg = 0.f, sgx = 0.f;
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    float gm[4], bt[4];
    unpack4(dv[j], g[j]);
    unpack4(sv[j], xh[j]);
    unpack4(*reinterpret_cast<const uint2*>(gamma + threadIdx.x * 4 + 1024 * j), gm);
    unpack4(*reinterpret_cast<const uint2*>(beta + threadIdx.x * 4 + 1024 * j), bt);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      xh[j][i] = (xh[j][i] - mean) * rstd;
      const float z = fmaf(xh[j][i], gm[i], bt[i]);
      g[j][i] *= (z > 0.f ? 1.0f : slope) * gm[i];
      sg += g[j][i];
      sgx = fmaf(g[j][i], xh[j][i], sgx);
    }
  }
  const float mg = block_sum(sg, red) * (1.0f / D), mgx = block_sum(sgx, red) * (1.0f / D);
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    float o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = rstd * (g[j][i] - mg - xh[j][i] * mgx);
    *reinterpret_cast<uint2*>(dx + base + 1024 * j) = pack4(o);
  }
}

__device__ inline int reflect1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// y[b][i][j][:] = x[b][refl(i-1)][refl(j-1)][:]   ([B][T][F][C] -> [B][T+2][F+2][C]; VEC elements per thread)
template <int VEC>
__global__ void reflect_pad_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int T, int F, int CV,
                                       long long total) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % CV);
  long long q = idx / CV;
  const int j = (int)(q % (F + 2));
  q /= (F + 2);
  const int i = (int)(q % (T + 2));
  const long long b = q / (T + 2);
  const size_t src = (((size_t)b * T + reflect1(i - 1, T)) * F + reflect1(j - 1, F)) * CV + c;
  if constexpr (VEC == 8) reinterpret_cast<uint4*>(y)[idx] = reinterpret_cast<const uint4*>(x)[src];
  else y[idx] = x[src];
}

// adjoint: dx[b][t][f] = sum of dy over the padded positions that read (t, f)
template <int VEC>
__global__ void reflect_pad_bwd_kernel(const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx, int T, int F, int CV,
                                       long long total) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % CV);
  long long q = idx / CV;
  const int f = (int)(q % F);
  q /= F;
  const int t = (int)(q % T);
  const long long b = q / T;
  int ti[3], fj[3], nt = 0, nf = 0;
  ti[nt++] = t + 1;
  if (t == 1) ti[nt++] = 0;
  if (t == T - 2) ti[nt++] = T + 1;
  fj[nf++] = f + 1;
  if (f == 1) fj[nf++] = 0;
  if (f == F - 2) fj[nf++] = F + 1;
  float acc[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
  for (int a = 0; a < nt; ++a)
    for (int e = 0; e < nf; ++e) {
      const size_t src = (((size_t)b * (T + 2) + ti[a]) * (F + 2) + fj[e]) * CV + c;
      if constexpr (VEC == 8) {
        float v[8];
        Tr<bf16_t>::unpack(reinterpret_cast<const uint4*>(dy)[src], v);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += v[k];
      } else {
        acc[0] += (float)dy[src];
      }
    }
  if constexpr (VEC == 8) reinterpret_cast<uint4*>(dx)[idx] = Tr<bf16_t>::pack(acc);
  else dx[idx] = (bf16_t)acc[0];
}

}  // namespace

extern "C" int sa_add_layernorm_fwd(const void* x, const void* r, const void* gamma, const void* beta, void* y,
                                    void* s_out, float* stat, int rows, int d, float eps, void* stream) {
  if (!x || !gamma || !beta || !y || rows <= 0) return -22;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid(sa_div_up(rows, 4)), block(256);
#define SA_LN_FWD(N)                                                                                        \
  hipLaunchKernelGGL(add_ln_fwd_kernel<N>, grid, block, 0, st, (const bf16_t*)x, (const bf16_t*)r,          \
                     (const bf16_t*)gamma, (const bf16_t*)beta, (bf16_t*)y, (bf16_t*)s_out, stat, rows, eps)
  switch (d) {
    case 256: SA_LN_FWD(1); break;
    case 512: SA_LN_FWD(2); break;
    case 768: SA_LN_FWD(3); break;
    case 1024: SA_LN_FWD(4); break;
    default: return -38;
  }
#undef SA_LN_FWD
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

extern "C" int sa_layernorm_bwd(const void* dy, const void* s, const float* stat, const void* gamma, void* ds,
                                int rows, int d, void* stream) {
  if (!dy || !s || !stat || !gamma || !ds || rows <= 0) return -22;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid(sa_div_up(rows, 4)), block(256);
#define SA_LN_BWD(N)                                                                                   \
  hipLaunchKernelGGL(ln_bwd_kernel<N>, grid, block, 0, st, (const bf16_t*)dy, (const bf16_t*)s, stat, \
                     (const bf16_t*)gamma, (bf16_t*)ds, rows)
  switch (d) {
    case 256: SA_LN_BWD(1); break;
    case 512: SA_LN_BWD(2); break;
    case 768: SA_LN_BWD(3); break;
    case 1024: SA_LN_BWD(4); break;
    default: return -38;
  }
#undef SA_LN_BWD
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

extern "C" int sa_reflect_pad_fwd(const void* x, void* y, int B, int T, int F, int C, void* stream) {
  if (!x || !y || B <= 0 || T < 3 || F < 3 || C <= 0) return -22;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (C % 8 == 0) {
    const long long total = (long long)B * (T + 2) * (F + 2) * (C / 8);
    hipLaunchKernelGGL(reflect_pad_fwd_kernel<8>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const bf16_t*)x, (bf16_t*)y, T, F, C / 8, total);
  } else {
    const long long total = (long long)B * (T + 2) * (F + 2) * C;
    hipLaunchKernelGGL(reflect_pad_fwd_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const bf16_t*)x, (bf16_t*)y, T, F, C, total);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

extern "C" int sa_reflect_pad_bwd(const void* dy, void* dx, int B, int T, int F, int C, void* stream) {
  if (!dy || !dx || B <= 0 || T < 3 || F < 3 || C <= 0) return -22;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (C % 8 == 0) {
    const long long total = (long long)B * T * F * (C / 8);
    hipLaunchKernelGGL(reflect_pad_bwd_kernel<8>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const bf16_t*)dy, (bf16_t*)dx, T, F, C / 8, total);
  } else {
    const long long total = (long long)B * T * F * C;
    hipLaunchKernelGGL(reflect_pad_bwd_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const bf16_t*)dy, (bf16_t*)dx, T, F, C, total);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

extern "C" int sa_ln_leaky_fwd(const void* x, const void* gamma, const void* beta, void* y, float* stat, int rows,
                               int d, float eps, float slope, void* stream) {
  if (!x || !gamma || !beta || !y || rows <= 0) return -22;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define SA_LL_FWD(N)                                                                                          \
  hipLaunchKernelGGL(ln_leaky_fwd_kernel<N>, dim3(rows), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)gamma, \
                     (const bf16_t*)beta, (bf16_t*)y, stat, eps, slope)
  switch (d) {
    case 5120: SA_LL_FWD(5); break;
    case 10240: SA_LL_FWD(10); break;
    default: return -38;
  }
#undef SA_LL_FWD
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

extern "C" int sa_ln_leaky_bwd(const void* dy, const void* x, const float* stat, const void* gamma, const void* beta,
                               void* dx, int rows, int d, float slope, void* stream) {
  if (!dy || !x || !stat || !gamma || !beta || !dx || rows <= 0) return -22;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define SA_LL_BWD(N)                                                                                            \
  hipLaunchKernelGGL(ln_leaky_bwd_kernel<N>, dim3(rows), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)x, stat, \
                     (const bf16_t*)gamma, (const bf16_t*)beta, (bf16_t*)dx, slope)
  switch (d) {
    case 5120: SA_LL_BWD(5); break;
    case 10240: SA_LL_BWD(10); break;
    default: return -38;
  }
#undef SA_LL_BWD
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// ---------------------------------------------------------------------------------------------------
// Block 0 of the convolutional front end as ONE pass each way: Conv2d(1 -> 128, 3 x 3, stride 2, reflect
// "same" padding) -> LayerNorm over (frequency, channel) -> LeakyReLU on [B][T][F] features
// (speechbrain_configs/convae.yaml:139-146; F = 80 -> F' = 40, 40 x 128 = 5120 outputs per output row).
// The convolution is 9 multiply-adds per output: the block is bound by its 165 MB of bf16 output, which the
// library path wrote (and re-read) four times (convolution 0.5 ms, layout copy, LayerNorm, LeakyReLU).
// One 256-thread workgroup per output row (b, i): thread t owns channels [4 (t % 32), +4) of the columns
// j = 8 k + t / 32 (k = 0..4), i.e. elements [1024 k + 4 t, +4) of the row -- the layout of ln_leaky above.
//   forward : the three padded input rows in LDS, z in registers, row statistics, y stored once
//   backward: z recomputed from the input (nothing but the statistics is kept), LayerNorm / LeakyReLU
//             backward, then the input gradient in two deterministic steps: per output row the sums over the
//             128 channels per tap -> a [3][F + 2] gradient of its padded input rows (part), and a gather
//             kernel that folds the overlap of neighbouring rows and the reflection into d x.
// ---------------------------------------------------------------------------------------------------
#define SA_B0_F 80
#define SA_B0_FO 40
#define SA_B0_C 128
#define SA_B0_D (SA_B0_FO * SA_B0_C)

__device__ inline void b0_stage_rows(const bf16_t* __restrict__ x, int b, int i, int T, float (*xp)[SA_B0_F + 2]) {
  // padded rows 2i .. 2i+2 = input rows reflect(2i - 1 + kh), padded columns q = reflect(q - 1)
  for (int e = threadIdx.x; e < 3 * (SA_B0_F + 2); e += 256) {
    const int kh = e / (SA_B0_F + 2), q = e % (SA_B0_F + 2);
    const int t = reflect1(2 * i - 1 + kh, T), f = reflect1(q - 1, SA_B0_F);
    xp[kh][q] = (float)x[((size_t)b * T + t) * SA_B0_F + f];
  }
}

__device__ inline void b0_conv(const float (*xp)[SA_B0_F + 2], const float (*wr)[9], const float* br, float (*z)[4]) {
  const int jb = threadIdx.x >> 5;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int j = 8 * k + jb;
    float v[9];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) v[3 * kh + kw] = xp[kh][2 * j + kw];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float s = br[c];
#pragma unroll
      for (int u = 0; u < 9; ++u) s = fmaf(wr[c][u], v[u], s);
      z[k][c] = s;
    }
  }
}

__global__ __launch_bounds__(256) void asr_block0_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                             const bf16_t* __restrict__ bias,
                                                             const bf16_t* __restrict__ gamma,
                                                             const bf16_t* __restrict__ beta, bf16_t* __restrict__ y,
                                                             float* __restrict__ stat, int T, int To, float eps,
                                                             float slope) {
  __shared__ float xp[3][SA_B0_F + 2];
  __shared__ float red[4];
  const int row = blockIdx.x, b = row / To, i = row % To;
  const int c0 = 4 * (threadIdx.x & 31);
  float wr[4][9], br[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    br[c] = (float)bias[c0 + c];
#pragma unroll
    for (int u = 0; u < 9; ++u) wr[c][u] = (float)w[(c0 + c) * 9 + u];
  }
  b0_stage_rows(x, b, i, T, xp);
  __syncthreads();
  float z[5][4];
  b0_conv(xp, wr, br, z);
  // the convolution's output is a bf16 tensor in the unfused graph: round it, so that the normalisation sees
  // the values the library path would have stored
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    unpack4(pack4(z[k]), z[k]);
    sum += (z[k][0] + z[k][1]) + (z[k][2] + z[k][3]);
  }
  const float mean = block_sum(sum, red) * (1.0f / SA_B0_D);
  float sq = 0.f;
#pragma unroll
  for (int k = 0; k < 5; ++k)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      z[k][c] -= mean;
      sq = fmaf(z[k][c], z[k][c], sq);
    }
  const float rstd = rsqrtf(block_sum(sq, red) * (1.0f / SA_B0_D) + eps);
  if (stat && threadIdx.x == 0) {
    stat[2 * row] = mean;
    stat[2 * row + 1] = rstd;
  }
  const size_t base = (size_t)row * SA_B0_D + threadIdx.x * 4;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    float g[4], bt[4], o[4];
    unpack4(*reinterpret_cast<const uint2*>(gamma + threadIdx.x * 4 + 1024 * k), g);
    unpack4(*reinterpret_cast<const uint2*>(beta + threadIdx.x * 4 + 1024 * k), bt);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float v = fmaf(z[k][c] * rstd, g[c], bt[c]);
      o[c] = v > 0.f ? v : v * slope;
    }
    *reinterpret_cast<uint2*>(y + base + 1024 * k) = pack4(o);
  }
}

// part[row][kh][q]: gradient of padded input row 2i + kh, padded column q, from output row `row` alone
__global__ __launch_bounds__(256) void asr_block0_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                             const bf16_t* __restrict__ w,
                                                             const bf16_t* __restrict__ bias,
                                                             const bf16_t* __restrict__ gamma,
                                                             const bf16_t* __restrict__ beta,
                                                             const float* __restrict__ stat, float* __restrict__ part,
                                                             int T, int To, float slope) {
  __shared__ float xp[3][SA_B0_F + 2];
  __shared__ float red[4];
  __shared__ float s9[SA_B0_FO][9];
  const int row = blockIdx.x, b = row / To, i = row % To;
  const int c0 = 4 * (threadIdx.x & 31), jb = threadIdx.x >> 5;
  float wr[4][9], br[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    br[c] = (float)bias[c0 + c];
#pragma unroll
    for (int u = 0; u < 9; ++u) wr[c][u] = (float)w[(c0 + c) * 9 + u];
  }
  b0_stage_rows(x, b, i, T, xp);
  __syncthreads();
  float z[5][4], g[5][4];
  b0_conv(xp, wr, br, z);
  const float mean = stat[2 * row], rstd = stat[2 * row + 1];
  const size_t base = (size_t)row * SA_B0_D + threadIdx.x * 4;
  float sg = 0.f, sgx = 0.f;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    float gm[4], bt[4];
    unpack4(pack4(z[k]), z[k]);                                     // (the forward's rounding)
    unpack4(*reinterpret_cast<const uint2*>(dy + base + 1024 * k), g[k]);
    unpack4(*reinterpret_cast<const uint2*>(gamma + threadIdx.x * 4 + 1024 * k), gm);
    unpack4(*reinterpret_cast<const uint2*>(beta + threadIdx.x * 4 + 1024 * k), bt);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      z[k][c] = (z[k][c] - mean) * rstd;                             // xhat
      const float v = fmaf(z[k][c], gm[c], bt[c]);
      g[k][c] *= (v > 0.f ? 1.0f : slope) * gm[c];
      sg += g[k][c];
      sgx = fmaf(g[k][c], z[k][c], sgx);
    }
  }
  const float mg = block_sum(sg, red) * (1.0f / SA_B0_D), mgx = block_sum(sgx, red) * (1.0f / SA_B0_D);
  // d z, then per column j and tap the sum over the 128 channels: 4 in the thread, 32 lanes by butterfly
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    float t9[9];
#pragma unroll
    for (int u = 0; u < 9; ++u) t9[u] = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      // the unfused graph holds d z as a bf16 tensor between the LayerNorm backward and the convolution's
      const float dz = (float)(bf16_t)(rstd * (g[k][c] - mg - z[k][c] * mgx));
#pragma unroll
      for (int u = 0; u < 9; ++u) t9[u] = fmaf(dz, wr[c][u], t9[u]);
    }
#pragma unroll
    for (int u = 0; u < 9; ++u) {
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) t9[u] += __shfl_xor(t9[u], o, 64);       // within the 32 lanes of a column
    }
    if ((threadIdx.x & 31) == 0) {
#pragma unroll
      for (int u = 0; u < 9; ++u) s9[8 * k + jb][u] = t9[u];
    }
  }
  __syncthreads();
  // padded column q of padded row kh collects (j, kw) with 2 j + kw = q: at most two terms, in j order
  for (int e = threadIdx.x; e < 3 * (SA_B0_F + 2); e += 256) {
    const int kh = e / (SA_B0_F + 2), q = e % (SA_B0_F + 2);
    float s = 0.f;
#pragma unroll
    for (int kw = 2; kw >= 0; --kw) {
      const int j2 = q - kw;
      if (j2 >= 0 && !(j2 & 1) && (j2 >> 1) < SA_B0_FO) s += s9[j2 >> 1][3 * kh + kw];
    }
    part[((size_t)row * 3 + kh) * (SA_B0_F + 2) + q] = s;
  }
}

// d x[b][t][f] = sum over the padded positions (p, q) that read (t, f), over the output rows i and taps kh
// with 2 i + kh = p, of part[b][i][kh][q] -- a fixed order, no atomics
__global__ void asr_block0_fold_kernel(const float* __restrict__ part, bf16_t* __restrict__ dx, int B, int T, int To) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)B * T * SA_B0_F) return;
  const int f = (int)(idx % SA_B0_F);
  const int t = (int)((idx / SA_B0_F) % T);
  const int b = (int)(idx / ((long long)SA_B0_F * T));
  int ps[3], qs[3], np = 0, nq = 0;
  ps[np++] = t + 1;
  if (t == 1) ps[np++] = 0;
  if (t == T - 2) ps[np++] = T + 1;
  qs[nq++] = f + 1;
  if (f == 1) qs[nq++] = 0;
  if (f == SA_B0_F - 2) qs[nq++] = SA_B0_F + 1;
  float s = 0.f;
  for (int a = 0; a < np; ++a)
    for (int kh = 0; kh < 3; ++kh) {
      const int i2 = ps[a] - kh;
      if (i2 < 0 || (i2 & 1) || (i2 >> 1) >= To) continue;
      const float* pr = part + (((size_t)b * To + (i2 >> 1)) * 3 + kh) * (SA_B0_F + 2);
      for (int e = 0; e < nq; ++e) s += pr[qs[e]];
    }
  dx[idx] = (bf16_t)s;
}

extern "C" int sa_asr_block0_fwd(const void* x, const void* w, const void* bias, const void* gamma, const void* beta,
                                 void* y, float* stat, int B, int T, int F, int C, float eps, float slope,
                                 void* stream) {
  if (!x || !w || !bias || !gamma || !beta || !y || B <= 0 || T < 3) return -22;
  if (F != SA_B0_F || C != SA_B0_C) return -38;
  const int To = (T - 1) / 2 + 1;
  hipLaunchKernelGGL(asr_block0_fwd_kernel, dim3(B * To), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     (const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)bias, (const bf16_t*)gamma, (const bf16_t*)beta,
                     (bf16_t*)y, stat, T, To, eps, slope);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

extern "C" int sa_asr_block0_bwd(const void* dy, const void* x, const void* w, const void* bias, const void* gamma,
                                 const void* beta, const float* stat, float* part, void* dx, int B, int T, int F,
                                 int C, float slope, void* stream) {
  if (!dy || !x || !w || !bias || !gamma || !beta || !stat || !part || !dx || B <= 0 || T < 3) return -22;
  if (F != SA_B0_F || C != SA_B0_C) return -38;
  const int To = (T - 1) / 2 + 1;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(asr_block0_bwd_kernel, dim3(B * To), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)x,
                     (const bf16_t*)w, (const bf16_t*)bias, (const bf16_t*)gamma, (const bf16_t*)beta, stat, part, T, To,
                     slope);
  const long long total = (long long)B * T * SA_B0_F;
  hipLaunchKernelGGL(asr_block0_fold_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, part,
                     (bf16_t*)dx, B, T, To);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}
