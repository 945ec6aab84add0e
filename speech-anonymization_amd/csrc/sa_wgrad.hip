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
  if (dtype == SA_BF16 || dtype == SA_BF16X1F)
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

// 8 consecutive k (tile rows k, k+1, .. at row stride `rs` elements) of one channel column per
// lane, from an untransposed [row][channel] bf16 LDS tile: two ds_read_b64_tr_b16.
// `p` = this lane's block address (its row q = (lane&15)>>2, its 4-column group).
__device__ static inline bf16x8 sa_tr_frag(const bf16_t* p, int rs) {
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p));
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
      (__attribute__((address_space(3))) bf16x4*)(p + 4 * rs));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// One workgroup = (utterance b, row chunk, MS x NS channel sub-block) and ALL taps: the
// (transformed) A rows and the dY rows are staged once per K-tile and reused by every tap (a
// tap is a row offset into the A tile).  Wave w owns the 32x32 output pair (mt, nt) =
// (w % NP / NT, w % NP % NT) for all taps (5 accumulators) and, when the sub-block has fewer
// than 4 pairs, the k-steps ks = w / NP (mod KW).  HBM loads of K-tile i+1 are in flight (in
// registers) while tile i is in the MFMAs.
template <typename T, int CIN, int COUT, int SA, int U>
struct WgCfg {
  typedef Pol<T> P;
  typedef typename P::lds_t LT;
  static constexpr int MS = CIN < 64 ? CIN : 64, NS = COUT < 64 ? COUT : 64;
  static constexpr int MT = MS / 32, NT = NS / 32, NP = MT * NT, KW = 4 / NP;
  static constexpr int KT = (P::NPL == 2 || sizeof(typename P::store_t) == 4) ? 32 : 64;
  static constexpr int HALO = 8;                                   // max tap offset spread
  static constexpr int RA = KT * SA + HALO, RB = KT * U;           // staged rows
  static constexpr int PA = WgPitch<LT, MS>::value, PB = WgPitch<LT, NS>::value;
  static constexpr int CHA = MS / P::VEC, CHB = NS / P::VEC;       // 16-byte chunks per row
  static constexpr int NITA = (RA * CHA + 255) / 256, NITB = (RB * CHB + 255) / 256;
};

template <typename T, int CIN, int COUT, int SA, int U>
__global__ __launch_bounds__(256) void sa_wgrad_kernel(SaWgradArgs a) {
  typedef WgCfg<T, CIN, COUT, SA, U> C;
  typedef Pol<T> P;
  typedef typename P::store_t S;
  typedef typename P::lds_t LT;
  typedef typename P::Frag Frag;
  typedef Tr<S> tr;
  constexpr int VEC = P::VEC, KS = P::KS, NPL = P::NPL, KT = C::KT;
  constexpr int PA = C::PA, PB = C::PB;
  __shared__ __attribute__((aligned(16))) LT At[NPL * C::RA * PA];
  __shared__ __attribute__((aligned(16))) LT Bt[NPL * C::RB * PB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // The NSUB channel sub-blocks of one (utterance, row chunk) read the same rows.  Workgroups are
  // dealt round-robin over the 8 XCDs, so siblings are given linear ids that differ by 8 (same
  // XCD L2, dispatched close together): id = group*8*NSUB + sub*8 + lane, tile = group*8 + lane.
  // Pure speed: any placement computes the same slabs.
  constexpr int NSUB = (CIN / C::MS) * (COUT / C::NS);
  const int lin = blockIdx.x;
  const int grp = lin / (8 * NSUB), rem = lin % (8 * NSUB);
  const int tile = grp * 8 + rem % 8, sub = rem / 8;
  const int ntile = a.nchunk * a.B;
  if (tile >= ntile) return;
  const int chunk = tile % a.nchunk, b = tile / a.nchunk;
  const int cm0 = (sub / (COUT / C::NS)) * C::MS, cn0 = (sub % (COUT / C::NS)) * C::NS;
  const int mbeg = chunk * a.chunk;
  int mend = mbeg + a.chunk; if (mend > a.Mrows) mend = a.Mrows;
  int offmin = a.off[0];
  for (int t = 1; t < a.ntaps; ++t) offmin = a.off[t] < offmin ? a.off[t] : offmin;

  const int pair = wave % C::NP, kw = wave / C::NP;
  const int mt = pair / C::NT, nt = pair % C::NT;
  f32x16 acc[SA_MAX_TAPS];
#pragma unroll
  for (int t = 0; t < SA_MAX_TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

  // prologue coefficients: 256 % CHA == 0, so a thread stages the same 16-byte channel chunk
  // in every iteration slot and its coefficients live in registers
  static_assert(256 % C::CHA == 0 && 256 % C::CHB == 0, "chunk column must be fixed per thread");
  const bool has1 = a.s1 != nullptr, has2 = a.s2 != nullptr, sw = a.swish != 0;
  float s1r[VEC], t1r[VEC], s2r[VEC], t2r[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const int ch = cm0 + (tid % C::CHA) * VEC + j;
    s1r[j] = has1 ? a.s1[(size_t)b * CIN + ch] : 1.0f;
    t1r[j] = has1 ? a.t1[(size_t)b * CIN + ch] : 0.0f;
    s2r[j] = has2 ? a.s2[ch] : 1.0f;
    t2r[j] = has2 ? a.t2[ch] : 0.0f;
  }
  const S* xb = reinterpret_cast<const S*>(a.x) + (size_t)b * a.Lin * CIN + cm0;
  const S* yb = reinterpret_cast<const S*>(a.dy) + (size_t)b * a.Ldy * COUT + cn0;
  const int dyend = mend * U < a.Ldy ? mend * U : a.Ldy;

  // two register sets: the rows of K-tiles i+1 and i+2 are in flight while tile i is in the MFMAs
  uint4 rawA0[C::NITA], rawB0[C::NITB], rawA1[C::NITA], rawB1[C::NITB];
  auto issue = [&](uint4 (&rawA)[C::NITA], uint4 (&rawB)[C::NITB], int m0) {
#pragma unroll
    for (int i = 0; i < C::NITA; ++i) {
      const int e = tid + i * 256, r = e / C::CHA, c = e % C::CHA;
      const int g = m0 * SA + offmin + r;
      rawA[i] = make_uint4(0, 0, 0, 0);
      if (m0 < mend && r < C::RA && g >= 0 && g < a.Lin)
        rawA[i] = *reinterpret_cast<const uint4*>(xb + (size_t)g * CIN + c * VEC);
    }
#pragma unroll
    for (int i = 0; i < C::NITB; ++i) {
      const int e = tid + i * 256, r = e / C::CHB, c = e % C::CHB;
      const int g = m0 * U + r;
      rawB[i] = make_uint4(0, 0, 0, 0);
      if (m0 < mend && r < C::RB && g < dyend) rawB[i] = *reinterpret_cast<const uint4*>(yb + (size_t)g * COUT + c * VEC);
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
  auto stage = [&](const uint4 (&rawA)[C::NITA], const uint4 (&rawB)[C::NITB], int m0) {
#pragma unroll
    for (int i = 0; i < C::NITA; ++i) {
      const int e = tid + i * 256, r = e / C::CHA, c = e % C::CHA;
      if (r < C::RA) {
        const int g = m0 * SA + offmin + r;
        float f[VEC];
        tr::unpack(rawA[i], f);
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
        put(At, C::RA * PA, PA, r, c, f);
      }
    }
#pragma unroll
    for (int i = 0; i < C::NITB; ++i) {
      const int e = tid + i * 256, r = e / C::CHB, c = e % C::CHB;
      if (r < C::RB) {
        float f[VEC];
        tr::unpack(rawB[i], f);
        put(Bt, C::RB * PB, PB, r, c, f);
      }
    }
  };

  const int g4 = lane >> 4, i16 = lane & 15;
  const int trow = 8 * (g4 >> 1) + (i16 >> 2);             // tr-read: k within the 16-deep step
  const int tcol = 16 * (g4 & 1) + 4 * (i16 & 3);          //          column within the 32-wide tile

  auto mfma_tile = [&]() {
    for (int ks = kw; ks < KT / KS; ks += C::KW) {
      if constexpr (sizeof(LT) == 2) {
        Frag bh[U], bl[U];
#pragma unroll
        for (int ph = 0; ph < U; ++ph) {
          const LT* p = Bt + (size_t)((ks * 16 + trow) * U + ph) * PB + nt * 32 + tcol;
          bh[ph] = sa_tr_frag(p, U * PB);
          if constexpr (NPL == 2) bl[ph] = sa_tr_frag(p + C::RB * PB, U * PB);
        }
#pragma unroll
        for (int t = 0; t < SA_MAX_TAPS; ++t) {
          if (t < a.ntaps) {
            const LT* p = At + (size_t)((ks * 16 + trow) * SA + a.off[t] - offmin) * PA + mt * 32 + tcol;
            const Frag ah = sa_tr_frag(p, SA * PA);
            const int ph = U == 1 ? 0 : a.ph[t];
            if constexpr (NPL == 2) {
              const Frag al = sa_tr_frag(p + C::RA * PA, SA * PA);
              acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, ph ? bh[U - 1] : bh[0], acc[t], 0, 0, 0);
              acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ph ? bl[U - 1] : bl[0], acc[t], 0, 0, 0);
              acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ph ? bh[U - 1] : bh[0], acc[t], 0, 0, 0);
            } else {
              acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ph ? bh[U - 1] : bh[0], acc[t], 0, 0, 0);
            }
          }
        }
      } else {
        float bv[U];
#pragma unroll
        for (int ph = 0; ph < U; ++ph)
          bv[ph] = Bt[(size_t)((ks * 2 + (lane >> 5)) * U + ph) * PB + nt * 32 + (lane & 31)];
#pragma unroll
        for (int t = 0; t < SA_MAX_TAPS; ++t) {
          if (t < a.ntaps) {
            const float av = At[(size_t)((ks * 2 + (lane >> 5)) * SA + a.off[t] - offmin) * PA + mt * 32 + (lane & 31)];
            const int ph = U == 1 ? 0 : a.ph[t];
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, ph ? bv[U - 1] : bv[0], acc[t], 0, 0, 0);
          }
        }
      }
    }
  };
  issue(rawA0, rawB0, mbeg);
  issue(rawA1, rawB1, mbeg + KT);
  for (int m0 = mbeg; m0 < mend; m0 += 2 * KT) {
    stage(rawA0, rawB0, m0);
    __syncthreads();
    issue(rawA0, rawB0, m0 + 2 * KT);
    mfma_tile();
    __syncthreads();
    if (m0 + KT < mend) {
      stage(rawA1, rawB1, m0 + KT);
      __syncthreads();
      issue(rawA1, rawB1, m0 + 3 * KT);
      mfma_tile();
      __syncthreads();
    }
  }

  // ---- fp32 partial slab [kw][tap][CIN][COUT]; this workgroup writes its MS x NS sub-block ----
  float* slab = a.slabs + ((((size_t)b * a.nchunk + chunk) * C::KW + kw) * a.ntaps) * CIN * COUT;
#pragma unroll
  for (int t = 0; t < SA_MAX_TAPS; ++t) {
    if (t < a.ntaps) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        slab[((size_t)t * CIN + cm0 + mt * 32 + sa_acc_row(r, lane)) * COUT + cn0 + nt * 32 + (lane & 31)] = acc[t][r];
    }
  }
}

// slabs per (utterance, chunk): the k-step split factor of the configuration
extern "C" int sa_wgrad_kw(int cin, int cout) {
  const int ms = cin < 64 ? cin : 64, ns = cout < 64 ? cout : 64;
  return 4 / ((ms / 32) * (ns / 32));
}

template <typename T, int CIN, int COUT, int SA, int U>
static int launch_wgrad(const SaWgradArgs& a, hipStream_t st) {
  typedef WgCfg<T, CIN, COUT, SA, U> C;
  int omin = a.off[0], omax = a.off[0];
  for (int t = 1; t < a.ntaps; ++t) {
    omin = a.off[t] < omin ? a.off[t] : omin;
    omax = a.off[t] > omax ? a.off[t] : omax;
  }
  if (omax - omin > C::HALO) return -22;
  for (int t = 0; t < a.ntaps; ++t) if (a.ph[t] < 0 || a.ph[t] >= U) return -22;
  const int nsub = (CIN / C::MS) * (COUT / C::NS);
  dim3 grid(sa_div_up(a.nchunk * a.B, 8) * 8 * nsub);
  hipLaunchKernelGGL((sa_wgrad_kernel<T, CIN, COUT, SA, U>), grid, dim3(256), 0, st, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

#define SA_WG_CASE(CI, CO, S, UU)                                               \
  if (cin == CI && cout == CO && sa == S && u == UU)                            \
    return dtype == SA_BF16 ? launch_wgrad<bf16_t, CI, CO, S, UU>(*a, st)       \
           : dtype == SA_BF16X3 ? launch_wgrad<bf16x3_t, CI, CO, S, UU>(*a, st) \
           : dtype == SA_BF16X1F ? launch_wgrad<bf16x1f_t, CI, CO, S, UU>(*a, st) \
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
// 256 threads = 16 outputs x 16 slab lanes (fixed summation order -> deterministic)
__global__ __launch_bounds__(256) void sa_wgrad_reduce_kernel(const float* __restrict__ slabs,
                                                              float* __restrict__ dst, int nslab,
                                                              int ntaps, int CIN, int COUT, int sk,
                                                              int sn, int st, int accumulate) {
  __shared__ double part[16][17];
  const int per = ntaps * CIN * COUT;
  const int o = threadIdx.x & 15, q = threadIdx.x >> 4, i = blockIdx.x * 16 + o;
  double s = 0.0;
  if (i < per) {
#pragma unroll 4
    for (int k = q; k < nslab; k += 16) s += (double)slabs[(size_t)k * per + i];
  }
  part[q][o] = s;
  __syncthreads();
  if (q == 0 && i < per) {
    double t = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += part[r][o];
    const int co = i % COUT, ci = (i / COUT) % CIN, tp = i / (COUT * CIN);
    const size_t d = (size_t)ci * sk + (size_t)co * sn + (size_t)tp * st;
    dst[d] = accumulate ? dst[d] + (float)t : (float)t;
  }
}

extern "C" int sa_wgrad_reduce(const float* slabs, float* dst, int nslab, int ntaps, int cin,
                               int cout, int sk, int sn, int st, int accumulate, void* stream) {
  if (!slabs || !dst || nslab <= 0) return -22;
  const int per = ntaps * cin * cout;
  hipLaunchKernelGGL(sa_wgrad_reduce_kernel, dim3(sa_div_up(per, 16)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), slabs, dst, nslab, ntaps, cin, cout,
                     sk, sn, st, accumulate);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}
