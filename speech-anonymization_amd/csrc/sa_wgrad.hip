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
// Each workgroup owns one (utterance, row chunk) with the whole CIN x COUT block and all taps
// and writes fp32 partial slabs; sa_wgrad_reduce sums the slabs in a fixed order
// (deterministic) straight into the PyTorch weight layout.  In the training step both operands
// arrive as bf16 caches written by the convolution launches (x_pre, dy_pre): the kernel is then
// pure copy-to-LDS + MFMA and HBM-bound.
#include "sa_common.h"
#include <type_traits>

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

// ---- SA_FP8 images: e4m3 fragments (8 per lane, same k order as the bf16 image) of w * scale,
// scale = 2^floor(log2(448 / max|w|)) per tensor, stored as a float right behind the image
// (byte offset ntaps*K*N, a multiple of 512)
__device__ static inline float sa_pow2_scale(float amax) {
  if (!(amax > 0.0f) || !(amax < 3.0e38f)) return 1.0f;
  return exp2f(floorf(log2f(448.0f / amax)));
}

__global__ void sa_pack_scale_kernel(const float* __restrict__ src, float* __restrict__ scale, int ntaps,
                                     int K, int N, int sk, int sn, int st) {
  __shared__ float red[256];
  float m = 0.0f;
  const int total = ntaps * K * N;
  for (int i = threadIdx.x; i < total; i += blockDim.x) {
    const int n = i % N, k = (i / N) % K, t = i / (N * K);
    m = fmaxf(m, fabsf(src[(size_t)k * sk + (size_t)n * sn + (size_t)t * st]));
  }
  red[threadIdx.x] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) scale[0] = sa_pow2_scale(red[0]);
}

__device__ static inline unsigned char sa_to_fp8(float v) {
  v = __builtin_amdgcn_fmed3f(v, -448.0f, 448.0f);
  return (unsigned char)(__builtin_amdgcn_cvt_pk_fp8_f32(v, 0.0f, 0, false) & 0xff);
}

__global__ void sa_pack_weights_fp8_kernel(const float* __restrict__ src, unsigned char* __restrict__ dst,
                                           const float* __restrict__ scale, int ntaps, int K, int N,
                                           int sk, int sn, int st) {
  constexpr int KS = 16, PER = 8;
  const int total = ntaps * K * N;
  const int NT = N / 32, KSTEPS = K / KS;
  const float sc = scale[0];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int j = i % PER;
    int r = i / PER;
    const int lane = r % 64; r /= 64;
    const int nt = r % NT; r /= NT;
    const int ks = r % KSTEPS;
    const int t = r / KSTEPS;
    const int k = ks * KS + PER * (lane >> 5) + j;
    const int n = nt * 32 + (lane & 31);
    dst[i] = sa_to_fp8(src[(size_t)k * sk + (size_t)n * sn + (size_t)t * st] * sc);
  }
}

__global__ void sa_pack_scales_multi_kernel(const SaPackDesc* __restrict__ descs) {
  const SaPackDesc d = descs[blockIdx.x];
  if (d.dtype != SA_FP8 || !d.scale) return;
  __shared__ float red[256];
  float m = 0.0f;
  const int total = d.ntaps * d.K * d.N;
  for (int i = threadIdx.x; i < total; i += blockDim.x) {
    const int n = i % d.N, k = (i / d.N) % d.K, t = i / (d.N * d.K);
    m = fmaxf(m, fabsf(d.src[(size_t)k * d.sk + (size_t)n * d.sn + (size_t)t * d.st]));
  }
  red[threadIdx.x] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) d.scale[0] = sa_pow2_scale(red[0]);
}

extern "C" int sa_pack_scales_multi(const SaPackDesc* descs, int n, void* stream) {
  if (!descs || n <= 0) return -22;
  hipLaunchKernelGGL(sa_pack_scales_multi_kernel, dim3(n), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), descs);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

extern "C" int sa_pack_weights(int dtype, const float* src, void* dst, int ntaps, int K, int N,
                               int sk, int sn, int st, void* stream) {
  if (!src || !dst || K % 16 || N % 32) return -22;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int total = ntaps * K * N;
  const int grid = sa_div_up(total, 256) < 1024 ? sa_div_up(total, 256) : 1024;
  if (dtype == SA_FP8) {
    float* scale = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(dst) + total);
    hipLaunchKernelGGL(sa_pack_scale_kernel, dim3(1), dim3(256), 0, s, src, scale, ntaps, K, N, sk, sn, st);
    hipLaunchKernelGGL(sa_pack_weights_fp8_kernel, dim3(grid), dim3(256), 0, s, src,
                       reinterpret_cast<unsigned char*>(dst), scale, ntaps, K, N, sk, sn, st);
  } else if (dtype == SA_BF16 || dtype == SA_BF16X1F)
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

// All weight images of a model in one launch: descs (device memory) holds n SaPackDesc records,
// workgroup column y packs image y with the same element mapping as sa_pack_weights.
__global__ void sa_pack_weights_multi_kernel(const SaPackDesc* __restrict__ descs) {
  const SaPackDesc d = descs[blockIdx.y];
  const bool f32 = d.dtype == SA_F32, split = d.dtype == SA_BF16X3;
  const int KS = f32 ? 2 : 16, PER = KS / 2;
  const int total = d.ntaps * d.K * d.N;
  const int NT = d.N / 32, KSTEPS = d.K / KS;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int j = i % PER;
    int r = i / PER;
    const int lane = r % 64; r /= 64;
    const int nt = r % NT; r /= NT;
    const int ks = r % KSTEPS;
    const int t = r / KSTEPS;
    const int k = ks * KS + PER * (lane >> 5) + j;
    const int n = nt * 32 + (lane & 31);
    const float w = d.src[(size_t)k * d.sk + (size_t)n * d.sn + (size_t)t * d.st];
    if (d.dtype == SA_FP8) {
      reinterpret_cast<unsigned char*>(d.dst)[i] = sa_to_fp8(w * d.scale[0]);
    } else if (f32) {
      reinterpret_cast<float*>(d.dst)[i] = w;
    } else {
      bf16_t* o = reinterpret_cast<bf16_t*>(d.dst);
      const bf16_t hi = (bf16_t)w;
      o[i] = hi;
      if (split) o[(size_t)total + i] = (bf16_t)(w - (float)hi);
    }
  }
}

extern "C" int sa_pack_weights_multi(const SaPackDesc* descs, int n, int blocks_per_image,
                                     void* stream) {
  if (!descs || n <= 0 || blocks_per_image <= 0) return -22;
  hipLaunchKernelGGL(sa_pack_weights_multi_kernel, dim3(blocks_per_image, n), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), descs);
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

// 8 consecutive k (tile rows k, k+1, .. at row stride `rs` elements) of one channel column per
// lane, from an untransposed [row][channel] bf16 LDS tile: two ds_read_b64_tr_b16.
// `p` = this lane's block address (its row q = (lane&15)>>2, its 4-column group).
__device__ static inline bf16x8 sa_tr_frag(const bf16_t* p, int rs) {
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p));
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
      (__attribute__((address_space(3))) bf16x4*)(p + 4 * rs));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// One workgroup = (utterance b, row chunk), the FULL CIN x COUT block and ALL taps: every A row
// and dY row is fetched from HBM exactly once (an earlier layout that split the channel block
// over four workgroups re-read each row twice: 1.32 GB instead of 0.66 GB per 128x128 launch).
// The (transformed) rows are staged once per K-tile and reused by every tap (a tap is a row
// offset into the A tile).  The 32x32 output pairs (mt, nt) are dealt over 4 or 8 waves; a
// wave owns PPW pairs of one mt for all taps (5*PPW accumulators) and, when there are fewer
// pairs than waves, the k-steps ks = kw (mod KW).  LDS holds two tile buffers: while the
// MFMAs run on buffer i the rows of K-tile i+1 (loaded one iteration earlier, in registers)
// are transformed and written to the other buffer and the loads of K-tile i+2 are issued;
// one barrier per K-tile.  Half of the waves stage first and multiply second, the other half
// the other way round, so that the VALU and MFMA phases of the two waves on a SIMD overlap.
// XPRE: the A rows come pre-transformed in bf16 (SaConvArgs.a_out of the forward launch): 8 elements
// per 16-byte chunk, no prologue arithmetic, and the registers that frees hold a second in-flight
// K-tile (loads are issued two tiles ahead).
// YPRE: likewise the dY rows come as bf16 (SaConvArgs.a_out of the data-gradient launch whose
// normalisation-backward prologue formed them).
template <typename T, int CIN, int COUT, int SA, int U, bool XPRE = false, bool YPRE = false>
struct WgCfg {
  typedef Pol<T> P;
  typedef typename P::lds_t LT;
  static constexpr int VECA = XPRE ? 8 : P::VEC;
  static constexpr int VECB = YPRE ? 8 : P::VEC;
  static constexpr int NSETS = XPRE ? 2 : 1;
  static constexpr int MT = CIN / 32, NT = COUT / 32, NP = MT * NT;
  static constexpr int NW = NP >= 8 ? 8 : 4;                       // waves per workgroup
  static constexpr int NTHR = NW * 64;
  static constexpr int PPW = NP > NW ? NP / NW : 1;                // pairs per wave (same mt)
  static constexpr int NPG = NP / PPW;                             // pair groups
  static constexpr int KW = NW / NPG;                              // k-step split
  static constexpr int KT = (P::NPL == 2 || sizeof(typename P::store_t) == 4) ? 32 : 64;
  static constexpr int HALO = 8;                                   // max tap offset spread
  static constexpr int RA = KT * SA + HALO, RB = KT * U;           // staged rows
  static constexpr int PA = WgPitch<LT, CIN>::value, PB = WgPitch<LT, COUT>::value;
  static constexpr int CHA = CIN / VECA, CHB = COUT / VECB;        // 16-byte chunks per row
  static constexpr int NITA = (RA * CHA + NTHR - 1) / NTHR, NITB = (RB * CHB + NTHR - 1) / NTHR;
  static constexpr int AEL = P::NPL * RA * PA, BEL = P::NPL * RB * PB;     // lds_t elements
  static constexpr int BUFEL = (AEL + BEL + 7) & ~7;
  static constexpr size_t LDS = 2 * (size_t)BUFEL * sizeof(LT) + 4 * CIN * sizeof(float);
  static_assert(NW % NPG == 0 && NT % PPW == 0, "pair dealing");
  static_assert(NTHR % CHA == 0 && NTHR % CHB == 0, "chunk column must be fixed per thread");
  static_assert(!XPRE || (sizeof(LT) == 2 && P::NPL == 1), "pre-transformed A is a bf16 single-plane operand");
  static_assert(!YPRE || XPRE, "bf16 dY comes with the bf16 A cache");
};

template <typename T, int CIN, int COUT, int SA, int U, bool XPRE, bool YPRE>
__global__ __launch_bounds__((WgCfg<T, CIN, COUT, SA, U, XPRE, YPRE>::NTHR)) void sa_wgrad_kernel(SaWgradArgs a) {
  typedef WgCfg<T, CIN, COUT, SA, U, XPRE, YPRE> C;
  typedef Pol<T> P;
  typedef typename P::store_t S;
  typedef typename P::lds_t LT;
  typedef typename P::Frag Frag;
  typedef Tr<S> tr;
  constexpr int VEC = P::VEC, KS = P::KS, NPL = P::NPL, KT = C::KT, PPW = C::PPW;
  constexpr int PA = C::PA, PB = C::PB, NTHR = C::NTHR, VECA = C::VECA, VECB = C::VECB, NSETS = C::NSETS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  LT* tiles = reinterpret_cast<LT*>(smem);
  float* coef = reinterpret_cast<float*>(smem + 2 * (size_t)C::BUFEL * sizeof(LT));   // s1 t1 s2 t2

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile = blockIdx.x;
  const int chunk = tile % a.nchunk, b = tile / a.nchunk;
  const int mbeg = chunk * a.chunk;
  int mend = mbeg + a.chunk; if (mend > a.Mrows) mend = a.Mrows;
  int offmin = a.off[0];
  for (int t = 1; t < a.ntaps; ++t) offmin = a.off[t] < offmin ? a.off[t] : offmin;

  const int grp = wave % C::NPG, kw = wave / C::NPG;
  const int mt = (grp * PPW) / C::NT, nt0 = (grp * PPW) % C::NT;
  f32x16 acc[PPW][SA_MAX_TAPS];
#pragma unroll
  for (int q = 0; q < PPW; ++q)
#pragma unroll
    for (int t = 0; t < SA_MAX_TAPS; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][t][r] = 0.0f;

  // prologue coefficients of this utterance: LDS (read back per staged chunk, so that they do
  // not hold registers across the MFMA phase)
  const bool has1 = !XPRE && a.s1 != nullptr, has2 = !XPRE && a.s2 != nullptr, sw = !XPRE && a.swish != 0;
  for (int ch = tid; ch < CIN; ch += NTHR) {
    coef[ch] = has1 ? a.s1[(size_t)b * CIN + ch] : 1.0f;
    coef[CIN + ch] = has1 ? a.t1[(size_t)b * CIN + ch] : 0.0f;
    coef[2 * CIN + ch] = has2 ? a.s2[ch] : 1.0f;
    coef[3 * CIN + ch] = has2 ? a.t2[ch] : 0.0f;
  }
  typedef typename std::conditional<XPRE, bf16_t, S>::type SX;        // storage type of the A rows
  const SX* xb = reinterpret_cast<const SX*>(a.x) + (size_t)b * a.Lin * CIN;
  typedef typename std::conditional<YPRE, bf16_t, S>::type SY;        // storage type of the dY rows
  const SY* yb = reinterpret_cast<const SY*>(a.dy) + (size_t)b * a.Ldy * COUT;
  const int dyend = mend * U < a.Ldy ? mend * U : a.Ldy;
  const int ca = tid % C::CHA, ra0 = tid / C::CHA, cb = tid % C::CHB, rb0 = tid / C::CHB;
  constexpr int RSA = NTHR / C::CHA, RSB = NTHR / C::CHB;           // row step per iteration slot

  struct RegSet { uint4 a[C::NITA]; uint4 b[C::NITB]; };
  RegSet set0, set1;
  auto issue = [&](RegSet& rs, int m0) {
#pragma unroll
    for (int i = 0; i < C::NITA; ++i) {
      const int r = ra0 + i * RSA, g = m0 * SA + offmin + r;
      rs.a[i] = make_uint4(0, 0, 0, 0);
      if (m0 < mend && r < C::RA && g >= 0 && g < a.Lin)
        rs.a[i] = *reinterpret_cast<const uint4*>(xb + (size_t)g * CIN + ca * VECA);
    }
#pragma unroll
    for (int i = 0; i < C::NITB; ++i) {
      const int r = rb0 + i * RSB, g = m0 * U + r;
      rs.b[i] = make_uint4(0, 0, 0, 0);
      if (m0 < mend && r < C::RB && g < dyend)
        rs.b[i] = *reinterpret_cast<const uint4*>(yb + (size_t)g * COUT + cb * VECB);
    }
  };
  auto put = [&](LT* base, int planes_stride, int pitch, int r, int c, const float* f) {
    LT* dst = base + (size_t)r * pitch + c * VEC;
    if constexpr (NPL == 2) {
      uint2 hi, lo;
      sa_split4(f, hi, lo);
      *reinterpret_cast<uint2*>(dst) = hi;
      *reinterpret_cast<uint2*>(dst + planes_stride) = lo;
    } else if constexpr (sizeof(S) == 4 && sizeof(LT) == 2) {
      *reinterpret_cast<uint2*>(dst) = sa_pack_bf16x4(f);
    } else {
      *reinterpret_cast<uint4*>(dst) = tr::pack(f);
    }
  };
  auto stage = [&](LT* At, const RegSet& rs, int m0) {
    LT* Bt = At + C::AEL;
    if constexpr (XPRE) {
#pragma unroll
      for (int i = 0; i < C::NITA; ++i) {
        const int r = ra0 + i * RSA;
        if (r < C::RA) *reinterpret_cast<uint4*>(At + (size_t)r * PA + ca * VECA) = rs.a[i];
      }
    } else {
      float s1r[VEC], t1r[VEC], s2r[VEC], t2r[VEC];
      if (has1 || has2) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          s1r[j] = coef[ca * VEC + j]; t1r[j] = coef[CIN + ca * VEC + j];
          s2r[j] = coef[2 * CIN + ca * VEC + j]; t2r[j] = coef[3 * CIN + ca * VEC + j];
        }
      }
#pragma unroll
      for (int i = 0; i < C::NITA; ++i) {
        const int r = ra0 + i * RSA;
        if (r < C::RA) {
          const int g = m0 * SA + offmin + r;
          float f[VEC];
          tr::unpack(rs.a[i], f);
          if ((has1 || has2 || sw) && g >= 0 && g < a.Lin) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
              float v = f[j];
              if (has1) v = fmaf(v, s1r[j], t1r[j]);
              if (sw) v = sa_swish(v);
              if (has2) v = fmaf(v, s2r[j], t2r[j]);
              f[j] = v;
            }
          }
          put(At, C::RA * PA, PA, r, ca, f);
        }
      }
    }
    if constexpr (YPRE) {
#pragma unroll
      for (int i = 0; i < C::NITB; ++i) {
        const int r = rb0 + i * RSB;
        if (r < C::RB) *reinterpret_cast<uint4*>(Bt + (size_t)r * PB + cb * VECB) = rs.b[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < C::NITB; ++i) {
        const int r = rb0 + i * RSB;
        if (r < C::RB) {
          float f[VEC];
          tr::unpack(rs.b[i], f);
          put(Bt, C::RB * PB, PB, r, cb, f);
        }
      }
    }
  };

  const int g4 = lane >> 4, i16 = lane & 15;
  const int trow = 8 * (g4 >> 1) + (i16 >> 2);             // tr-read: k within the 16-deep step
  const int tcol = 16 * (g4 & 1) + 4 * (i16 & 3);          //          column within the 32-wide tile

  auto mfma_tile = [&](const LT* At) {
    const LT* Bt = At + C::AEL;
    for (int ks = kw; ks < KT / KS; ks += C::KW) {
      if constexpr (sizeof(LT) == 2) {
        Frag bh[PPW][U], bl[PPW][U];
#pragma unroll
        for (int q = 0; q < PPW; ++q)
#pragma unroll
          for (int ph = 0; ph < U; ++ph) {
            const LT* p = Bt + (size_t)((ks * 16 + trow) * U + ph) * PB + (nt0 + q) * 32 + tcol;
            bh[q][ph] = sa_tr_frag(p, U * PB);
            if constexpr (NPL == 2) bl[q][ph] = sa_tr_frag(p + C::RB * PB, U * PB);
          }
#pragma unroll
        for (int t = 0; t < SA_MAX_TAPS; ++t) {
          if (t < a.ntaps) {
            const LT* p = At + (size_t)((ks * 16 + trow) * SA + a.off[t] - offmin) * PA + mt * 32 + tcol;
            const Frag ah = sa_tr_frag(p, SA * PA);
            const int ph = U == 1 ? 0 : a.ph[t];
            Frag al;
            if constexpr (NPL == 2) al = sa_tr_frag(p + C::RA * PA, SA * PA);
#pragma unroll
            for (int q = 0; q < PPW; ++q) {
              if constexpr (NPL == 2) {
                acc[q][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, ph ? bh[q][U - 1] : bh[q][0], acc[q][t], 0, 0, 0);
                acc[q][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ph ? bl[q][U - 1] : bl[q][0], acc[q][t], 0, 0, 0);
              }
              acc[q][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ph ? bh[q][U - 1] : bh[q][0], acc[q][t], 0, 0, 0);
            }
          }
        }
      } else {
        float bv[PPW][U];
#pragma unroll
        for (int q = 0; q < PPW; ++q)
#pragma unroll
          for (int ph = 0; ph < U; ++ph)
            bv[q][ph] = Bt[(size_t)((ks * 2 + (lane >> 5)) * U + ph) * PB + (nt0 + q) * 32 + (lane & 31)];
#pragma unroll
        for (int t = 0; t < SA_MAX_TAPS; ++t) {
          if (t < a.ntaps) {
            const float av = At[(size_t)((ks * 2 + (lane >> 5)) * SA + a.off[t] - offmin) * PA + mt * 32 + (lane & 31)];
            const int ph = U == 1 ? 0 : a.ph[t];
#pragma unroll
            for (int q = 0; q < PPW; ++q)
              acc[q][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, ph ? bv[q][U - 1] : bv[q][0], acc[q][t], 0, 0, 0);
          }
        }
      }
    }
  };

  issue(set0, mbeg);
  if constexpr (NSETS == 2) issue(set1, mbeg + KT);
  __syncthreads();                                   // coefficients visible
  stage(tiles, set0, mbeg);
  issue(set0, mbeg + NSETS * KT);
  __syncthreads();
  const bool stage_first = wave < C::NW / 2;
  // tile i is multiplied from buffer i&1 while tile i+1 (register set `rs`) is staged into the
  // other buffer and the set is re-issued NSETS tiles ahead
  auto step = [&](RegSet& rs, int m0, int cur) {
    LT* now = tiles + (size_t)cur * C::BUFEL;
    LT* nxt = tiles + (size_t)(cur ^ 1) * C::BUFEL;
    const bool more = m0 + KT < mend;
    if (stage_first) {
      if (more) { stage(nxt, rs, m0 + KT); issue(rs, m0 + KT + NSETS * KT); }
      mfma_tile(now);
    } else {
      mfma_tile(now);
      if (more) { stage(nxt, rs, m0 + KT); issue(rs, m0 + KT + NSETS * KT); }
    }
    __syncthreads();
  };
  for (int m0 = mbeg; m0 < mend; m0 += 2 * KT) {
    step(NSETS == 2 ? set1 : set0, m0, 0);
    if (m0 + KT < mend) step(set0, m0 + KT, 1);
  }

  // ---- fp32 partial slab [kw][tap][CIN][COUT] of this (utterance, chunk) ----
  float* slab = a.slabs + ((((size_t)b * a.nchunk + chunk) * C::KW + kw) * a.ntaps) * CIN * COUT;
#pragma unroll
  for (int q = 0; q < PPW; ++q)
#pragma unroll
    for (int t = 0; t < SA_MAX_TAPS; ++t) {
      if (t < a.ntaps) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          slab[((size_t)t * CIN + mt * 32 + sa_acc_row(r, lane)) * COUT + (nt0 + q) * 32 + (lane & 31)] = acc[q][t][r];
      }
    }
}

// slabs per (utterance, chunk): the k-step split factor of the configuration
extern "C" int sa_wgrad_kw(int cin, int cout) {
  const int np = (cin / 32) * (cout / 32);
  const int nw = np >= 8 ? 8 : 4, ppw = np > nw ? np / nw : 1;
  return nw / (np / ppw);
}

template <typename T, int CIN, int COUT, int SA, int U, bool XPRE = false, bool YPRE = false>
static int launch_wgrad(const SaWgradArgs& a, hipStream_t st) {
  typedef WgCfg<T, CIN, COUT, SA, U, XPRE, YPRE> C;
  int omin = a.off[0], omax = a.off[0];
  for (int t = 1; t < a.ntaps; ++t) {
    omin = a.off[t] < omin ? a.off[t] : omin;
    omax = a.off[t] > omax ? a.off[t] : omax;
  }
  if (omax - omin > C::HALO) return -22;
  for (int t = 0; t < a.ntaps; ++t) if (a.ph[t] < 0 || a.ph[t] >= U) return -22;
  if (C::LDS > 160 * 1024) return -12;
  auto kern = sa_wgrad_kernel<T, CIN, COUT, SA, U, XPRE, YPRE>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return -(int)e;
    attr_set = true;
  }
  dim3 grid(a.nchunk * a.B);
  hipLaunchKernelGGL(kern, grid, dim3(C::NTHR), C::LDS, st, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

#define SA_WG_CASE(CI, CO, S, UU)                                               \
  if (cin == CI && cout == CO && sa == S && u == UU)                            \
    return dtype == SA_BF16 ? (a->dy_pre ? launch_wgrad<bf16_t, CI, CO, S, UU, true, true>(*a, st)    \
                               : a->x_pre ? launch_wgrad<bf16_t, CI, CO, S, UU, true>(*a, st)       \
                                          : launch_wgrad<bf16_t, CI, CO, S, UU>(*a, st))            \
           : dtype == SA_BF16X3 ? launch_wgrad<bf16x3_t, CI, CO, S, UU>(*a, st) \
           : dtype == SA_BF16X1F ? (a->dy_pre ? launch_wgrad<bf16x1f_t, CI, CO, S, UU, true, true>(*a, st) \
                                    : a->x_pre ? launch_wgrad<bf16x1f_t, CI, CO, S, UU, true>(*a, st)     \
                                             : launch_wgrad<bf16x1f_t, CI, CO, S, UU>(*a, st)) \
                                : launch_wgrad<float, CI, CO, S, UU>(*a, st);

extern "C" int sa_wgrad(int dtype, int cin, int cout, int sa, int u, const SaWgradArgs* a,
                        void* stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!a || !a->x || !a->dy || !a->slabs || a->ntaps < 1 || a->ntaps > SA_MAX_TAPS ||
      a->chunk <= 0 || a->chunk % 64 || a->nchunk * a->chunk < a->Mrows ||
      (a->x_pre && dtype != SA_BF16X1F && dtype != SA_BF16) || (a->dy_pre && !a->x_pre))
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
// 256 threads = 4 waves; a wave covers 64 x VEC consecutive outputs (one 256 x VEC byte row of a
// slab per load instruction), the four waves take the slabs k = wave, wave + 4, ... with eight
// loads in flight per thread; the four partial sums are added in wave order (fixed summation
// order -> deterministic).  VEC is chosen so that the grid still covers the chip for the small
// layers.  (Before: 16 outputs x 16 slab lanes, 64-byte segments: 18.6 us for the 84 MB of a
// 128 -> 128 layer.)
template <int VEC>
__device__ __forceinline__ void wred_body(const float* __restrict__ slabs, float* __restrict__ dst, int nslab,
                                          int ntaps, int CIN, int COUT, int sk, int sn, int st, int accumulate,
                                          int bx, double (*part)[64 * 4]) {
  const int per = ntaps * CIN * COUT;
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int i0 = (bx * 64 + lane) * VEC;
  double s[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) s[j] = 0.0;
  if (i0 < per) {                                   // per % VEC == 0 (host check)
    const float* p = slabs + i0;
    int k = q;
    for (; k + 28 < nslab; k += 32) {               // eight independent loads, then the adds in k order
      float v[8][VEC];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float* r = p + (size_t)(k + 4 * u) * per;
        if constexpr (VEC == 4) { const float4 t = *reinterpret_cast<const float4*>(r); v[u][0] = t.x; v[u][1] = t.y; v[u][2] = t.z; v[u][3] = t.w; }
        else if constexpr (VEC == 2) { const float2 t = *reinterpret_cast<const float2*>(r); v[u][0] = t.x; v[u][1] = t.y; }
        else v[u][0] = *r;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int j = 0; j < VEC; ++j) s[j] += (double)v[u][j];
    }
    for (; k < nslab; k += 4) {
      const float* r = p + (size_t)k * per;
#pragma unroll
      for (int j = 0; j < VEC; ++j) s[j] += (double)r[j];
    }
  }
  if (q > 0) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) part[q - 1][lane * VEC + j] = s[j];
  }
  __syncthreads();
  if (q == 0 && i0 < per) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const double t = ((s[j] + part[0][lane * VEC + j]) + part[1][lane * VEC + j]) + part[2][lane * VEC + j];
      const int i = i0 + j;
      const int co = i % COUT, ci = (i / COUT) % CIN, tp = i / (COUT * CIN);
      const size_t d = (size_t)ci * sk + (size_t)co * sn + (size_t)tp * st;
      dst[d] = accumulate ? dst[d] + (float)t : (float)t;
    }
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void sa_wgrad_reduce_kernel(const float* __restrict__ slabs,
                                                              float* __restrict__ dst, int nslab,
                                                              int ntaps, int CIN, int COUT, int sk,
                                                              int sn, int st, int accumulate) {
  __shared__ double part[3][64 * 4];
  wred_body<VEC>(slabs, dst, nslab, ntaps, CIN, COUT, sk, sn, st, accumulate, blockIdx.x, part);
}

// the reducers of all weight gradients of a backward stage in ONE launch (they are read by nobody before the
// stage's bucket is reduced): record blockIdx.y, the same per-record kernel bodies, order and bits
__global__ __launch_bounds__(256) void sa_wgrad_reduce_multi_kernel(SaWredMulti m) {
  __shared__ double part[3][64 * 4];
  const SaWredDesc& d = m.d[blockIdx.y];
  const int per = d.ntaps * d.cin * d.cout;
  if ((int)blockIdx.x * 64 * d.vec >= per) return;               // (uniform per workgroup)
  if (d.vec == 4) wred_body<4>(d.slabs, d.dst, d.nslab, d.ntaps, d.cin, d.cout, d.sk, d.sn, d.st, d.accumulate, blockIdx.x, part);
  else if (d.vec == 2) wred_body<2>(d.slabs, d.dst, d.nslab, d.ntaps, d.cin, d.cout, d.sk, d.sn, d.st, d.accumulate, blockIdx.x, part);
  else wred_body<1>(d.slabs, d.dst, d.nslab, d.ntaps, d.cin, d.cout, d.sk, d.sn, d.st, d.accumulate, blockIdx.x, part);
}

static int wred_vec(int per) {                                   // (the choice of sa_wgrad_reduce)
  if (per % 4 == 0 && per >= 4 * 64 * 256) return 4;
  if (per % 2 == 0 && per >= 2 * 64 * 256) return 2;
  return 1;
}

extern "C" int sa_wgrad_reduce_multi(const SaWredMulti* m, void* stream) {
  if (!m || m->n <= 0 || m->n > SA_WRED_MAX) return -22;
  SaWredMulti mm = *m;
  int gx = 0;
  for (int j = 0; j < mm.n; ++j) {
    SaWredDesc& d = mm.d[j];
    if (!d.slabs || !d.dst || d.nslab <= 0 || d.ntaps <= 0 || d.cin <= 0 || d.cout <= 0) return -22;
    const int per = d.ntaps * d.cin * d.cout;
    d.vec = wred_vec(per);
    const int nb = sa_div_up(per, 64 * d.vec);
    gx = nb > gx ? nb : gx;
  }
  hipLaunchKernelGGL(sa_wgrad_reduce_multi_kernel, dim3(gx, mm.n), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), mm);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

extern "C" int sa_wgrad_reduce(const float* slabs, float* dst, int nslab, int ntaps, int cin,
                               int cout, int sk, int sn, int st, int accumulate, void* stream) {
  if (!slabs || !dst || nslab <= 0) return -22;
  const int per = ntaps * cin * cout;
  hipStream_t s_ = reinterpret_cast<hipStream_t>(stream);
  // widest rows that still give the chip >= 256 workgroups (and divide `per`)
  if (per % 4 == 0 && per >= 4 * 64 * 256)
    hipLaunchKernelGGL(sa_wgrad_reduce_kernel<4>, dim3(sa_div_up(per, 256)), dim3(256), 0, s_, slabs, dst,
                       nslab, ntaps, cin, cout, sk, sn, st, accumulate);
  else if (per % 2 == 0 && per >= 2 * 64 * 256)
    hipLaunchKernelGGL(sa_wgrad_reduce_kernel<2>, dim3(sa_div_up(per, 128)), dim3(256), 0, s_, slabs, dst,
                       nslab, ntaps, cin, cout, sk, sn, st, accumulate);
  else
    hipLaunchKernelGGL(sa_wgrad_reduce_kernel<1>, dim3(sa_div_up(per, 64)), dim3(256), 0, s_, slabs, dst,
                       nslab, ntaps, cin, cout, sk, sn, st, accumulate);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}
