// Weight packing, weight-gradient GEMM (split-K over rows, MFMA) and its reducer.
//
// Replaces what autograd does for the reference's Conv1d / ConvTranspose1d weights
// (the backward of models/ConvAutoEncoder.py:141-172, :33-43 reached from
// speechbrain_convae_train.py:241).
//
//   dW[t][ci][co] = sum_b sum_m  A[b, m*SA + off_t, ci] * dY[b, m*U + ph_t, co]
//
// Rows are the GEMM reduction dimension and the strided one in the channels-last layout,
// so the bf16 MFMA operand fragments (8 consecutive k per lane) are fetched from untransposed
// [row][channel] LDS tiles with the gfx950 hardware transpose read ds_read_b64_tr_b16
// (cdna_hip_programming.md T10); a tap shift is then just a row offset.  The f32 path
// (v_mfma_f32_32x32x2_f32, one k per lane) reads the same tiles with ds_read_b32.
// Each workgroup owns one (utterance, row chunk, tap) and writes an fp32 partial slab;
// sa_wgrad_reduce sums the slabs in a fixed order (deterministic) straight into the
// PyTorch weight layout.
#include "sa_common.h"

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------
// sa_pack_weights: fp32 master weights (PyTorch layout) -> fragment-major operand image
//   image[((t*KSTEPS + ks)*NT + nt)*64 + lane] = Frag of W(t, k, n)
//     bf16: k = ks*16 + 8*(lane>>5) + j (j = 0..7), n = nt*32 + (lane&31)
//     f32 : k = ks*2 + (lane>>5),                  n = nt*32 + (lane&31)
//   source element W(t,k,n) = src[k*sk + n*sn + t*st]
// ---------------------------------------------------------------------------------
template <typename T>
__global__ void sa_pack_weights_kernel(const float* __restrict__ src, T* __restrict__ dst,
                                       int ntaps, int K, int N, int sk, int sn, int st) {
  constexpr int KS = Tr<T>::KS;
  constexpr int PER = KS / 2;                     // k values per lane
  const int total = ntaps * K * N;
  const int NT = N / 32, KSTEPS = K / KS;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int j = i % PER;
    int r = i / PER;
    const int lane = r % 64; r /= 64;
    const int nt = r % NT; r /= NT;
    const int ks = r % KSTEPS;
    const int t = r / KSTEPS;
    const int k = ks * KS + PER * (lane >> 5) + j;
    const int n = nt * 32 + (lane & 31);
    dst[i] = Tr<T>::from_f(src[(size_t)k * sk + (size_t)n * sn + (size_t)t * st]);
  }
}

// split image: hi = bf16(w) image followed by lo = bf16(w - hi) image (SA_BF16X3)
__global__ void sa_pack_weights_split_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst,
                                             int ntaps, int K, int N, int sk, int sn, int st) {
  constexpr int KS = 16, PER = 8;
  const int total = ntaps * K * N;
  const int NT = N / 32, KSTEPS = K / KS;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int j = i % PER;
    int r = i / PER;
    const int lane = r % 64; r /= 64;
    const int nt = r % NT; r /= NT;
    const int ks = r % KSTEPS;
    const int t = r / KSTEPS;
    const int k = ks * KS + PER * (lane >> 5) + j;
    const int n = nt * 32 + (lane & 31);
    const float w = src[(size_t)k * sk + (size_t)n * sn + (size_t)t * st];
    const bf16_t hi = (bf16_t)w;
    dst[i] = hi;
    dst[(size_t)total + i] = (bf16_t)(w - (float)hi);
  }
}

extern "C" int sa_pack_weights(int dtype, const float* src, void* dst, int ntaps, int K, int N,
                               int sk, int sn, int st, void* stream) {
  if (!src || !dst || K % 16 || N % 32) return -22;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int total = ntaps * K * N;
  const int grid = sa_div_up(total, 256) < 1024 ? sa_div_up(total, 256) : 1024;
  if (dtype == SA_BF16)
    hipLaunchKernelGGL(sa_pack_weights_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, src,
                       reinterpret_cast<bf16_t*>(dst), ntaps, K, N, sk, sn, st);
  else if (dtype == SA_BF16X3)
    hipLaunchKernelGGL(sa_pack_weights_split_kernel, dim3(grid), dim3(256), 0, s, src,
                       reinterpret_cast<bf16_t*>(dst), ntaps, K, N, sk, sn, st);
  else
    hipLaunchKernelGGL(sa_pack_weights_kernel<float>, dim3(grid), dim3(256), 0, s, src,
                       reinterpret_cast<float*>(dst), ntaps, K, N, sk, sn, st);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// ---------------------------------------------------------------------------------
// wgrad GEMM
// ---------------------------------------------------------------------------------

template <typename LT, int C> struct WgPitch {
  // bf16: (pitch bytes / 4) mod 64 in {16, 48} makes the 4 rows of a tr16_b64 half-wave
  // block fall on disjoint bank windows; f32: any pitch is conflict free for ds_read_b32.
  static constexpr int value = sizeof(LT) == 2 ? (C >= 64 ? C + 32 : C) : C;
};

// 8 consecutive k (rows) of one channel column per lane, from an untransposed [row][channel]
// bf16 LDS tile: two ds_read_b64_tr_b16 (rows +0..3, +4..7).  `p` = this lane's block address.
__device__ static inline bf16x8 sa_tr_frag(const bf16_t* p, int pitch) {
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p));
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
      (__attribute__((address_space(3))) bf16x4*)(p + 4 * pitch));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <typename T, int CIN, int COUT, int SA, int U>
__global__ __launch_bounds__(256) void sa_wgrad_kernel(SaWgradArgs a) {
  typedef Pol<T> P;
  typedef typename P::store_t S;
  typedef typename P::lds_t LT;
  typedef typename P::Frag Frag;
  typedef Tr<S> tr;
  constexpr int VEC = P::VEC, KS = P::KS, NPL = P::NPL;
  constexpr int KT = NPL == 2 ? 32 : 64;              // rows per LDS tile
  constexpr int PA = WgPitch<LT, CIN>::value, PB = WgPitch<LT, COUT>::value;
  constexpr int MT = CIN / 32, NT = COUT / 32;
  constexpr int WN = NT >= 4 ? 4 : NT, WM = 4 / WN;   // waves over n-tiles / m-tiles
  constexpr int MPW = (MT + WM - 1) / WM;             // m-tiles per wave
  constexpr int NPW = NT / WN;
  constexpr int CHA = CIN / VEC, RPA = 256 / CHA, CHB = COUT / VEC, RPB = 256 / CHB;
  __shared__ __attribute__((aligned(16))) LT At[NPL * KT * PA];
  __shared__ __attribute__((aligned(16))) LT Bt[NPL * KT * PB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int chunk = blockIdx.x, tap = blockIdx.y, b = blockIdx.z;
  const int off = a.off[tap], ph = a.ph[tap];
  const int mbeg = chunk * a.chunk;
  int mend = mbeg + a.chunk; if (mend > a.Mrows) mend = a.Mrows;

  const int wn = wave % WN, wm = wave / WN;
  f32x16 acc[MPW][NPW];
#pragma unroll
  for (int i = 0; i < MPW; ++i)
#pragma unroll
    for (int j = 0; j < NPW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int ca = tid % CHA, ra0 = tid / CHA, cb = tid % CHB, rb0 = tid / CHB;
  float s1[VEC], t1[VEC], s2[VEC], t2[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    s1[j] = a.s1 ? a.s1[(size_t)b * CIN + ca * VEC + j] : 1.0f;
    t1[j] = a.t1 ? a.t1[(size_t)b * CIN + ca * VEC + j] : 0.0f;
    s2[j] = a.s2 ? a.s2[ca * VEC + j] : 1.0f;
    t2[j] = a.t2 ? a.t2[ca * VEC + j] : 0.0f;
  }
  const bool has1 = a.s1 != nullptr, has2 = a.s2 != nullptr, sw = a.swish != 0;
  const S* xb = reinterpret_cast<const S*>(a.x) + (size_t)b * a.Lin * CIN + ca * VEC;
  const S* yb = reinterpret_cast<const S*>(a.dy) + (size_t)b * a.Ldy * COUT + cb * VEC;

  for (int mt0 = mbeg; mt0 < mend; mt0 += KT) {
    // ---- stage KT rows of A (transformed) and dY; out-of-range rows are zero ----
    for (int r = ra0; r < KT; r += RPA) {
      const int m = mt0 + r, g = m * SA + off;
      float f[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) f[j] = 0.0f;
      if (m < mend && g >= 0 && g < a.Lin) {
        tr::unpack(*reinterpret_cast<const uint4*>(xb + (size_t)g * CIN), f);
        if (has1 || has2 || sw) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            float v = f[j];
            if (has1) v = fmaf(v, s1[j], t1[j]);
            if (sw) v = sa_swish(v);
            if (has2) v = fmaf(v, s2[j], t2[j]);
            f[j] = v;
          }
        }
      }
      LT* dst = At + (size_t)r * PA + ca * VEC;
      if constexpr (NPL == 2) {
        uint2 hi, lo;
        sa_split4(f, hi, lo);
        *reinterpret_cast<uint2*>(dst) = hi;
        *reinterpret_cast<uint2*>(dst + KT * PA) = lo;
      } else {
        *reinterpret_cast<uint4*>(dst) = tr::pack(f);
      }
    }
    for (int r = rb0; r < KT; r += RPB) {
      const int m = mt0 + r, g = m * U + ph;
      uint4 u = make_uint4(0, 0, 0, 0);
      if (m < mend && g < a.Ldy) u = *reinterpret_cast<const uint4*>(yb + (size_t)g * COUT);
      LT* dst = Bt + (size_t)r * PB + cb * VEC;
      if constexpr (NPL == 2) {
        float f[VEC];
        tr::unpack(u, f);
        uint2 hi, lo;
        sa_split4(f, hi, lo);
        *reinterpret_cast<uint2*>(dst) = hi;
        *reinterpret_cast<uint2*>(dst + KT * PB) = lo;
      } else {
        *reinterpret_cast<uint4*>(dst) = u;
      }
    }
    __syncthreads();
    // ---- MFMA over the KT rows ----
    const int g4 = lane >> 4, i16 = lane & 15;
    const int troff = (8 * (g4 >> 1) + (i16 >> 2));       // tr-read: row within the k-step block
    const int tcoff = 16 * (g4 & 1) + 4 * (i16 & 3);      //          column within the 32-wide tile
#pragma unroll 2
    for (int k0 = 0; k0 < KT; k0 += KS) {
      Frag bfr[NPW], bfl[NPW];
#pragma unroll
      for (int j = 0; j < NPW; ++j) {
        const int nt = wn + j * WN;
        if constexpr (sizeof(LT) == 2) {
          const LT* p = Bt + (size_t)(k0 + troff) * PB + nt * 32 + tcoff;
          bfr[j] = sa_tr_frag(p, PB);
          if constexpr (NPL == 2) bfl[j] = sa_tr_frag(p + KT * PB, PB);
        } else {
          bfr[j] = Bt[(size_t)(k0 + (lane >> 5)) * PB + nt * 32 + (lane & 31)];
        }
      }
#pragma unroll
      for (int i2 = 0; i2 < MPW; ++i2) {
        const int mt = wm + i2 * WM;
        if (mt < MT) {
          if constexpr (sizeof(LT) == 2) {
            const LT* p = At + (size_t)(k0 + troff) * PA + mt * 32 + tcoff;
            const Frag af = sa_tr_frag(p, PA);
            if constexpr (NPL == 2) {
              const Frag al = sa_tr_frag(p + KT * PA, PA);
#pragma unroll
              for (int j = 0; j < NPW; ++j) {
                acc[i2][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bfr[j], acc[i2][j], 0, 0, 0);
                acc[i2][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfl[j], acc[i2][j], 0, 0, 0);
                acc[i2][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr[j], acc[i2][j], 0, 0, 0);
              }
            } else {
#pragma unroll
              for (int j = 0; j < NPW; ++j)
                acc[i2][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr[j], acc[i2][j], 0, 0, 0);
            }
          } else {
            const Frag af = At[(size_t)(k0 + (lane >> 5)) * PA + mt * 32 + (lane & 31)];
#pragma unroll
            for (int j = 0; j < NPW; ++j)
              acc[i2][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bfr[j], acc[i2][j], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();
  }

  // ---- write the fp32 partial slab [CIN][COUT] ----
  float* slab = a.slabs + ((((size_t)b * a.nchunk + chunk) * a.ntaps + tap) * CIN) * COUT;
#pragma unroll
  for (int i2 = 0; i2 < MPW; ++i2) {
    const int mt = wm + i2 * WM;
    if (mt < MT) {
#pragma unroll
      for (int j = 0; j < NPW; ++j) {
        const int nt = wn + j * WN;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          slab[(size_t)(mt * 32 + sa_acc_row(r, lane)) * COUT + nt * 32 + (lane & 31)] = acc[i2][j][r];
      }
    }
  }
}

template <typename T, int CIN, int COUT, int SA, int U>
static int launch_wgrad(const SaWgradArgs& a, hipStream_t st) {
  dim3 grid(a.nchunk, a.ntaps, a.B);
  hipLaunchKernelGGL((sa_wgrad_kernel<T, CIN, COUT, SA, U>), grid, dim3(256), 0, st, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

#define SA_WG_CASE(CI, CO, S, UU)                                               \
  if (cin == CI && cout == CO && sa == S && u == UU)                            \
    return dtype == SA_BF16 ? launch_wgrad<bf16_t, CI, CO, S, UU>(*a, st)       \
           : dtype == SA_BF16X3 ? launch_wgrad<bf16x3_t, CI, CO, S, UU>(*a, st) \
                                : launch_wgrad<float, CI, CO, S, UU>(*a, st);

extern "C" int sa_wgrad(int dtype, int cin, int cout, int sa, int u, const SaWgradArgs* a,
                        void* stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!a || !a->x || !a->dy || !a->slabs || a->ntaps < 1 || a->ntaps > SA_MAX_TAPS ||
      a->chunk <= 0 || a->chunk % 64 || a->nchunk * a->chunk < a->Mrows)
    return -22;
  SA_WG_CASE(32, 64, 2, 1)
  SA_WG_CASE(64, 64, 1, 1)
  SA_WG_CASE(64, 128, 2, 1)
  SA_WG_CASE(128, 128, 1, 1)
  SA_WG_CASE(128, 64, 1, 2)
  SA_WG_CASE(64, 32, 1, 2)
  return -38;
}

// ---------------------------------------------------------------------------------
// sa_wgrad_reduce: dst[ci*sk + co*sn + t*st] (=|+=) sum over nslab slabs [t][ci][co]
// ---------------------------------------------------------------------------------
__global__ void sa_wgrad_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dst,
                                       int nslab, int ntaps, int CIN, int COUT, int sk, int sn,
                                       int st, int accumulate) {
  const int per = ntaps * CIN * COUT;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < per; i += gridDim.x * blockDim.x) {
    double s = 0.0;
    for (int k = 0; k < nslab; ++k) s += (double)slabs[(size_t)k * per + i];
    const int co = i % COUT, ci = (i / COUT) % CIN, t = i / (COUT * CIN);
    const size_t d = (size_t)ci * sk + (size_t)co * sn + (size_t)t * st;
    dst[d] = accumulate ? dst[d] + (float)s : (float)s;
  }
}

extern "C" int sa_wgrad_reduce(const float* slabs, float* dst, int nslab, int ntaps, int cin,
                               int cout, int sk, int sn, int st, int accumulate, void* stream) {
  if (!slabs || !dst || nslab <= 0) return -22;
  const int per = ntaps * cin * cout;
  hipLaunchKernelGGL(sa_wgrad_reduce_kernel, dim3(sa_div_up(per, 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), slabs, dst, nslab, ntaps, cin, cout,
                     sk, sn, st, accumulate);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}
