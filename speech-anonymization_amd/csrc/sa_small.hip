// The two single-channel ends of the ConvAutoencoder (HBM-bound, plain VALU + LDS):
//   encoder.0  Conv1d(1 -> 32, k15, p7)   models/ConvAutoEncoder.py:142
//   decoder.8  Conv1d(32 -> 1, k15, p7)   models/ConvAutoEncoder.py:171
// and their gradients.  Algorithmic traffic is 80 -> 2560 (or 2560 -> 80) elements per frame;
// every element crosses HBM once with 16-byte coalesced accesses.
#include "sa_common.h"

#define SA_K15 15
#define SA_C32 32

// y[b][l][c] = bias[c] + sum_k x[b][l+k-7] * w[c][flip ? 14-k : k]      (x, w fp32; y T)
// Used as encoder.0 forward (flip=0) and as decoder.8 dgrad (x = d recon, flip=1).
// Backward epilogue (ep_x != null; the decoder.8 dgrad feeds [InstanceNorm -> x*sigmoid(x)]
// backward): what is stored is g' = y * swish'(z), z = ep_x*ep_s1[b][c] + ep_t1[b][c], and stats
// becomes (sum g', sum g'*(ep_x - ep_mean[b][c])*ep_rstd[b][c]) -- the sa_ew_stats pass fused in.
template <typename T>
__global__ __launch_bounds__(256) void sa_conv1toC_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ w,
                                                          const float* __restrict__ bias,
                                                          T* __restrict__ y, int L, int flip,
                                                          float* __restrict__ stats, int ntiles,
                                                          const T* __restrict__ ep_x,
                                                          const float* __restrict__ ep_s1,
                                                          const float* __restrict__ ep_t1,
                                                          const float* __restrict__ ep_mean,
                                                          const float* __restrict__ ep_rstd) {
  constexpr int VEC = Tr<T>::VEC, CH = SA_C32 / VEC, PPP = 256 / CH, TILE = 512;
  __shared__ float xs[TILE + SA_K15 - 1];
  __shared__ __attribute__((aligned(16))) float ws[SA_K15][SA_C32];
  __shared__ float red[PPP][SA_C32][2];
  const int tid = threadIdx.x, b = blockIdx.y, l0 = blockIdx.x * TILE;
  for (int i = tid; i < TILE + SA_K15 - 1; i += 256) {
    const int g = l0 + i - 7;
    xs[i] = (g >= 0 && g < L) ? x[(size_t)b * L + g] : 0.0f;
  }
  for (int i = tid; i < SA_K15 * SA_C32; i += 256) {
    const int k = i / SA_C32, c = i % SA_C32;
    ws[k][c] = w[c * SA_K15 + (flip ? SA_K15 - 1 - k : k)];
  }
  __syncthreads();
  const int c = tid % CH, p0 = tid / CH;
  float bv[VEC], ssum[VEC], ssq[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) { bv[j] = bias ? bias[c * VEC + j] : 0.0f; ssum[j] = 0.f; ssq[j] = 0.f; }
  float es[VEC], et[VEC], em[VEC], er[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const size_t bc = (size_t)b * SA_C32 + c * VEC + j;
    es[j] = ep_x ? ep_s1[bc] : 1.0f; et[j] = ep_x ? ep_t1[bc] : 0.0f;
    em[j] = ep_x ? ep_mean[bc] : 0.0f; er[j] = ep_x ? ep_rstd[bc] : 0.0f;
  }
  for (int p = p0; p < TILE; p += PPP) {
    const int l = l0 + p;
    if (l >= L) break;
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = bv[j];
#pragma unroll
    for (int k = 0; k < SA_K15; ++k) {
      const float xv = xs[p + k];
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] = fmaf(xv, ws[k][c * VEC + j], acc[j]);
    }
    float xn[VEC];
    if (ep_x) {
      float xv[VEC];
      Tr<T>::unpack(*reinterpret_cast<const uint4*>(ep_x + ((size_t)b * L + l) * SA_C32 + c * VEC), xv);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        acc[j] *= sa_swish_grad(fmaf(xv[j], es[j], et[j]));
        xn[j] = (xv[j] - em[j]) * er[j];
      }
    }
    const uint4 u = Tr<T>::pack(acc);
    *reinterpret_cast<uint4*>(y + ((size_t)b * L + l) * SA_C32 + c * VEC) = u;
    if (stats) {
      float f[VEC];
      Tr<T>::unpack(u, f);
#pragma unroll
      for (int j = 0; j < VEC; ++j) { ssum[j] += f[j]; ssq[j] = fmaf(f[j], ep_x ? xn[j] : f[j], ssq[j]); }
    }
  }
  if (stats) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) { red[p0][c * VEC + j][0] = ssum[j]; red[p0][c * VEC + j][1] = ssq[j]; }
    __syncthreads();
    if (tid < SA_C32) {
      float s = 0.f, q = 0.f;
      for (int r = 0; r < PPP; ++r) { s += red[r][tid][0]; q += red[r][tid][1]; }
      float* d = stats + (((size_t)b * ntiles + blockIdx.x) * SA_C32 + tid) * 2;
      d[0] = s; d[1] = q;
    }
  }
}

extern "C" int sa_conv1toC_ntiles(int L) { return sa_div_up(L, 512); }

extern "C" int sa_conv1toC(int dtype, const float* x, const float* w, const float* bias, void* y,
                           int B, int L, int flip, float* stats, const void* ep_x, const float* ep_s1,
                           const float* ep_t1, const float* ep_mean, const float* ep_rstd,
                           void* stream) {
  if (!x || !w || !y || B <= 0 || L <= 0) return -22;
  if (ep_x && (!ep_s1 || !ep_t1 || !ep_mean || !ep_rstd)) return -22;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int nt = sa_div_up(L, 512);
  dim3 grid(nt, B);
  if (dtype == SA_BF16)
    hipLaunchKernelGGL(sa_conv1toC_kernel<bf16_t>, grid, dim3(256), 0, st, x, w, bias,
                       reinterpret_cast<bf16_t*>(y), L, flip, stats, nt,
                       reinterpret_cast<const bf16_t*>(ep_x), ep_s1, ep_t1, ep_mean, ep_rstd);
  else
    hipLaunchKernelGGL(sa_conv1toC_kernel<float>, grid, dim3(256), 0, st, x, w, bias,
                       reinterpret_cast<float*>(y), L, flip, stats, nt,
                       reinterpret_cast<const float*>(ep_x), ep_s1, ep_t1, ep_mean, ep_rstd);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// y[b][l] = bias + sum_k sum_c P(x[b][l+k-7][c]) * w[c][k]       (x T, y fp32)
// P = prologue: v*s1[b][c]+t1[b][c], then x*sigmoid(x) if swish.   decoder.8 forward
// (flip=0) and encoder.0 dgrad (flip=1: taps reversed).
// Two steps per 128-output tile (30 KB of LDS: five workgroups per CU overlap each other's loads): Tp[p][k] = sum_c P(x)[p][c] * w[c][k] for the 142 staged
// positions on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32: M = positions, N = 16 tap slots,
// K = 32 channels), then y[l] = sum_k Tp[l + k][k] (15 LDS reads per output instead of 240).
template <typename T>
__global__ __launch_bounds__(256) void sa_convCto1_kernel(const T* __restrict__ x,
                                                          const float* __restrict__ w,
                                                          const float* __restrict__ bias,
                                                          float* __restrict__ y, int L,
                                                          const float* __restrict__ s1,
                                                          const float* __restrict__ t1, int swish,
                                                          int flip) {
  constexpr int VEC = Tr<T>::VEC, CH = SA_C32 / VEC, RPP = 256 / CH, TILE = 128, PITCH = 36;
  constexpr int NPOS = TILE + SA_K15 - 1, NBLK = (NPOS + 15) / 16, TP = 17;
  constexpr int NIT = (NBLK * 16 + RPP - 1) / RPP;
  __shared__ __attribute__((aligned(16))) float xs[NBLK * 16 * PITCH];
  __shared__ float tp[NBLK * 16 * TP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.y, l0 = blockIdx.x * TILE;
  {
    const int c = tid % CH, r0 = tid / CH;
    float sc[VEC], sh[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      sc[j] = s1 ? s1[(size_t)b * SA_C32 + c * VEC + j] : 1.0f;
      sh[j] = t1 ? t1[(size_t)b * SA_C32 + c * VEC + j] : 0.0f;
    }
    uint4 raw[NIT];                                   // all row loads in flight before the first use
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int r = r0 + i * RPP, g = l0 + r - 7;
      raw[i] = make_uint4(0, 0, 0, 0);
      if (r < NPOS && g >= 0 && g < L)
        raw[i] = *reinterpret_cast<const uint4*>(x + ((size_t)b * L + g) * SA_C32 + c * VEC);
    }
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int r = r0 + i * RPP, g = l0 + r - 7;
      if (r < NBLK * 16) {
        float f[VEC];
        Tr<T>::unpack(raw[i], f);
        const bool valid = r < NPOS && g >= 0 && g < L;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          float v = fmaf(f[j], sc[j], sh[j]);
          v = swish ? sa_swish(v) : v;
          xs[r * PITCH + c * VEC + j] = valid ? v : 0.0f;
        }
      }
    }
  }
  // B operand: w[c = 4*ks + (lane>>4)][tap slot n = lane&15] (slot 15 is zero)
  const int n = lane & 15, kq = lane >> 4;
  float wb[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks)
    wb[ks] = n < SA_K15 ? w[(ks * 4 + kq) * SA_K15 + (flip ? SA_K15 - 1 - n : n)] : 0.0f;
  __syncthreads();
  for (int blk = wave; blk < NBLK; blk += 4) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* ap = xs + (blk * 16 + n) * PITCH + kq;        // A[m = position][k = channel]
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[ks * 4], wb[ks], acc, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) tp[(blk * 16 + 4 * kq + i) * TP + n] = acc[i];
  }
  __syncthreads();
  const int l = l0 + tid;
  if (tid < TILE && l < L) {
    float acc = bias ? bias[0] : 0.0f;
#pragma unroll
    for (int k = 0; k < SA_K15; ++k) acc += tp[(tid + k) * TP + k];
    y[(size_t)b * L + l] = acc;
  }
}

extern "C" int sa_convCto1(int dtype, const void* x, const float* w, const float* bias, float* y,
                           int B, int L, const float* s1, const float* t1, int swish, int flip,
                           void* stream) {
  if (!x || !w || !y || B <= 0 || L <= 0) return -22;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(sa_div_up(L, 128), B);             // TILE of sa_convCto1_kernel
  if (dtype == SA_BF16)
    hipLaunchKernelGGL(sa_convCto1_kernel<bf16_t>, grid, dim3(256), 0, st,
                       reinterpret_cast<const bf16_t*>(x), w, bias, y, L, s1, t1, swish, flip);
  else
    hipLaunchKernelGGL(sa_convCto1_kernel<float>, grid, dim3(256), 0, st,
                       reinterpret_cast<const float*>(x), w, bias, y, L, s1, t1, swish, flip);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// slab[wg][c][k] = sum_{l in chunk} u[b][l + kk - 7] * P(v[b][l][c]),  kk = flip ? 14-k : k
// encoder.0 wgrad: u = feats, v = d y0 (flip 0);  decoder.8 wgrad: u = d recon, v = y8 with the
// InstanceNorm+swish prologue (flip 1).  Bias gradients come from the per-(b,c) sums of the
// producing kernels.
// A [taps x positions] x [positions x channels] GEMM on the exact-fp32 MFMA
// (v_mfma_f32_16x16x4_f32): A[m][q] = u[q + kk(m)] is read straight from the staged u row (a
// Toeplitz operand needs no copy), B[q][c] = P(v)[q][c] from the staged tile.  Each wave owns 64
// of the tile's 256 positions; k-slot j of step t is position t + 16*j, which puts the four rows
// of a B fragment 16*33 floats apart = disjoint 16-bank windows (pitch 33).
template <typename T>
__global__ __launch_bounds__(256) void sa_wgrad1C_kernel(const float* __restrict__ u,
                                                         const T* __restrict__ v,
                                                         float* __restrict__ slabs, int L,
                                                         int chunk, int flip,
                                                         const float* __restrict__ s1,
                                                         const float* __restrict__ t1, int swish) {
  constexpr int VEC = Tr<T>::VEC, CH = SA_C32 / VEC, RPP = 256 / CH, TILE = 256, PITCH = 33;
  __shared__ float us[TILE + SA_K15 - 1 + 1];
  __shared__ float vs[TILE * PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.y;
  const int lbeg = blockIdx.x * chunk;
  int lend = lbeg + chunk; if (lend > L) lend = L;
  const int c = tid % CH, r0 = tid / CH;
  float sc[VEC], sh[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    sc[j] = s1 ? s1[(size_t)b * SA_C32 + c * VEC + j] : 1.0f;
    sh[j] = t1 ? t1[(size_t)b * SA_C32 + c * VEC + j] : 0.0f;
  }
  const bool tr_on = (s1 != nullptr) || swish;
  const int m = lane & 15, j4 = lane >> 4;                 // MFMA row (tap) / k-slot of this lane
  const bool mval = m < SA_K15;
  const int kk = mval ? (flip ? SA_K15 - 1 - m : m) : 0;
  const int qa = wave * 64 + 16 * j4;                      // first position of this lane's k-slot
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  // the rows of tile i+1 are in flight (registers) while tile i is staged and multiplied
  constexpr int NIT = TILE / RPP, NU = (TILE + SA_K15 - 1 + 255) / 256;
  uint4 raw[NIT];
  float rawu[NU];
  auto issue = [&](int l0) {
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int g = l0 + r0 + i * RPP;
      raw[i] = make_uint4(0, 0, 0, 0);
      if (g < lend) raw[i] = *reinterpret_cast<const uint4*>(v + ((size_t)b * L + g) * SA_C32 + c * VEC);
    }
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int g = l0 + tid + i * 256 - 7;
      rawu[i] = (l0 < lend && tid + i * 256 < TILE + SA_K15 - 1 && g >= 0 && g < L) ? u[(size_t)b * L + g] : 0.0f;
    }
  };
  issue(lbeg);
  for (int l0 = lbeg; l0 < lend; l0 += TILE) {
#pragma unroll
    for (int i = 0; i < NU; ++i)
      if (tid + i * 256 < TILE + SA_K15 - 1) us[tid + i * 256] = rawu[i];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int r = r0 + i * RPP, g = l0 + r;
      float f[VEC];
      Tr<T>::unpack(raw[i], f);
      if (tr_on && g < lend) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          float z = fmaf(f[j], sc[j], sh[j]);
          f[j] = swish ? sa_swish(z) : z;
        }
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) vs[r * PITCH + c * VEC + j] = f[j];
    }
    __syncthreads();
    issue(l0 + TILE);
#pragma unroll 4
    for (int t = 0; t < 16; ++t) {
      const int q = qa + t;
      const float av = mval ? us[q + kk] : 0.0f;
      const float b0 = vs[q * PITCH + m], b1 = vs[q * PITCH + 16 + m];
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1, acc1, 0, 0, 0);
    }
    __syncthreads();
  }
  // D[tap = 4*(lane>>4) + i][channel = nt*16 + (lane&15)]: add the four waves' partials in wave order
  float* part = vs;                                        // [4][16][32]
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    part[(wave * 16 + 4 * j4 + i) * 32 + m] = acc0[i];
    part[(wave * 16 + 4 * j4 + i) * 32 + 16 + m] = acc1[i];
  }
  __syncthreads();
  for (int o = tid; o < SA_K15 * SA_C32; o += 256) {
    const int k = o / SA_C32, cc = o % SA_C32;
    const float sum = ((part[(0 * 16 + k) * 32 + cc] + part[(1 * 16 + k) * 32 + cc]) +
                       part[(2 * 16 + k) * 32 + cc]) + part[(3 * 16 + k) * 32 + cc];
    slabs[((size_t)b * gridDim.x + blockIdx.x) * (SA_C32 * SA_K15) + cc * SA_K15 + k] = sum;
  }
}

extern "C" int sa_wgrad1C_nchunk(int L, int chunk) { return sa_div_up(L, chunk); }

extern "C" int sa_wgrad1C(int dtype, const float* u, const void* v, float* slabs, int B, int L,
                          int chunk, int flip, const float* s1, const float* t1, int swish,
                          void* stream) {
  if (!u || !v || !slabs || B <= 0 || L <= 0 || chunk <= 0 || chunk % 256) return -22;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(sa_div_up(L, chunk), B);
  if (dtype == SA_BF16)
    hipLaunchKernelGGL(sa_wgrad1C_kernel<bf16_t>, grid, dim3(256), 0, st, u,
                       reinterpret_cast<const bf16_t*>(v), slabs, L, chunk, flip, s1, t1, swish);
  else
    hipLaunchKernelGGL(sa_wgrad1C_kernel<float>, grid, dim3(256), 0, st, u,
                       reinterpret_cast<const float*>(v), slabs, L, chunk, flip, s1, t1, swish);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// dst[i] (=|+=) sum_k slabs[k][i]   (fixed order, double accumulate; 16 outputs x 16 slab lanes)
__global__ __launch_bounds__(256) void sa_sum_slabs_kernel(const float* __restrict__ slabs,
                                                           float* __restrict__ dst, int nslab, int n,
                                                           int accumulate) {
  __shared__ double part[16][17];
  const int o = threadIdx.x & 15, q = threadIdx.x >> 4, i = blockIdx.x * 16 + o;
  double s = 0.0;
  if (i < n) {
#pragma unroll 4
    for (int k = q; k < nslab; k += 16) s += (double)slabs[(size_t)k * n + i];
  }
  part[q][o] = s;
  __syncthreads();
  if (q == 0 && i < n) {
    double t = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += part[r][o];
    dst[i] = accumulate ? dst[i] + (float)t : (float)t;
  }
}

extern "C" int sa_sum_slabs(const float* slabs, float* dst, int nslab, int n, int accumulate,
                            void* stream) {
  if (!slabs || !dst || nslab <= 0 || n <= 0) return -22;
  hipLaunchKernelGGL(sa_sum_slabs_kernel, dim3(sa_div_up(n, 16)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), slabs, dst, nslab, n, accumulate);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}
