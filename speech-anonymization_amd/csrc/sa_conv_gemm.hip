// Implicit-GEMM 1-D convolution on MFMA (gfx950), channels-last.
//
// Replaces the cuDNN/ATen conv1d / conv_transpose1d forward and their data gradients that
// the reference reaches from models/ConvAutoEncoder.py:141-172 (encoder / decoder stacks)
// and :33-43 (TDNN convs of the sex classifier).
//
// One workgroup (4 waves) produces 64 or 128 output rows (per-shape policy, tile_rows()) x COUT
// channels of one utterance:
//   prologue : the input rows it needs (rows*SA/U + halo) are read ONCE from HBM with
//              16-byte coalesced loads, transformed on the fly and staged in LDS (split into
//              hi / lo bf16 planes in the bf16x3 mode).  The transform is either the producer's
//              normalisation (InstanceNorm/BatchNorm affine + x*sigmoid(x): forward launches,
//              optionally also emitting the bf16 operand cache `a_out` and its statistics
//              `pro_stats`) or the normalisation BACKWARD apply d y = c1*dz + c2*y + c3 over two
//              input tensors (data-gradient launches, `nb_*`; also emits bf16 d y and column
//              sums) -- neither is a pass of its own over HBM;
//   main loop: per tap and 16-deep (bf16) / 2-deep (f32) k-step, A fragments come from
//              LDS (ds_read_b128, padded pitch => conflict-free), B fragments (weights,
//              pre-packed fragment-major, L2-resident) straight from global memory;
//              v_mfma_f32_32x32x16_bf16 / v_mfma_f32_32x32x2_f32, fp32 accumulate;
//   epilogue : bias (+ReLU), transpose through LDS, 16-byte coalesced stores, and the
//              per-(utterance, channel) sum / sum-of-squares of the stored values written
//              as a per-tile partial slab (deterministic two-level reduction; no atomics);
//              data-gradient launches instead apply the activation backward and emit the
//              normalisation-backward statistics of the layer above (`ep_*`).
#include <type_traits>
#include "sa_common.h"

// -DSA_ABL=<mask>: timing-only ablation builds (tools/conv_ablate.py; WRONG numerics, never shipped):
//   1 no MFMA   2 weight fragments loaded once   4 no epilogue   8 no prologue transform
//   16 A fragments loaded once   32 no prologue row loads
#ifndef SA_ABL
#define SA_ABL 0
#endif
// -DSA_CONV_STAMPS: diagnostic build (tools/conv_stamps.py) that stamps s_memtime at the phase
// boundaries of a few workgroups; no stamp exists in the normal build.
#ifdef SA_CONV_STAMPS
__device__ unsigned long long sa_conv_dbg[8 * 64];
#define SA_STAMP_(i, op) do { if (tid == 0 && (blockIdx.x % 97) == 5 && blockIdx.y == 0) { \
  unsigned long long t_; asm volatile(op " %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
  sa_conv_dbg[(blockIdx.x / 97 % 64) * 8 + (i)] = t_; } } while (0)
#define SA_STAMP(i) SA_STAMP_(i, "s_memtime")
#define SA_STAMP_RT(i) SA_STAMP_(i, "s_memrealtime")
extern "C" int sa_conv_dbg_read(unsigned long long* out) {
  return -(int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sa_conv_dbg), sizeof(sa_conv_dbg));
}
#else
#define SA_STAMP(i)
#define SA_STAMP_RT(i)
#endif      // max (largest - smallest) tap row offset the prologue is sized for

#include "sa_conv_cfg.h"

template <typename T, int CIN, int COUT, int SA, int U, int TM, bool PRO2>
// (256, 2): with an explicit minimum of resident workgroups hipcc allocates one unified register
// file (104 registers for 128->128) instead of parking weight fragments in AGPRs behind
// v_accvgpr_write copies (32 extra instructions per 12 MFMAs in the main loop).
__global__ __launch_bounds__(256, 2) void sa_conv_gemm_kernel(SaConvArgs a, int red_off, int col_off) {
  typedef ConvCfg<T, CIN, COUT, SA, U, TM> C;
  typedef Pol<T> P;
  typedef typename P::store_t S;
  typedef typename P::lds_t LT;
  typedef typename P::Frag Frag;
  typedef Tr<S> tr;
  constexpr int VEC = C::VEC, OVEC = C::OVEC;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  LT* As = reinterpret_cast<LT*>(smem);
  S* Os = reinterpret_cast<S*>(smem);                       // overlays As after the main loop
  float* red = reinterpret_cast<float*>(smem + red_off);
  const int plane = a.nrows * C::APITCH;                    // lo plane offset (split mode)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile = blockIdx.x, b = blockIdx.y;
  const int m0 = tile * C::BMB;

  SA_STAMP(0);
  // ---------------- prologue: stage + transform the input rows --------------------
  {
    const int c = tid % C::CHI, r0 = tid / C::CHI;
    float s1[VEC], t1[VEC], s2[VEC], t2[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      s1[j] = a.s1 ? a.s1[(size_t)b * CIN + c * VEC + j] : 1.0f;
      t1[j] = a.t1 ? a.t1[(size_t)b * CIN + c * VEC + j] : 0.0f;
      s2[j] = a.s2 ? a.s2[c * VEC + j] : 1.0f;
      t2[j] = a.t2 ? a.t2[c * VEC + j] : 0.0f;
    }
    const bool has1 = a.s1 != nullptr, has2 = a.s2 != nullptr, sw = a.swish != 0;
    const S* xb = reinterpret_cast<const S*>(a.x) + (size_t)b * a.Lin * CIN + c * VEC;
    const int gbase = m0 * SA + a.rowmin;
    // rows this tile owns (a_out, nb_colsum): its base rows; the last tile also the trailing halo
    const int own_lo = m0 * SA;
    int own_hi = tile == a.ntiles - 1 ? a.Lin : (m0 + C::BMB) * SA;
    if (own_hi > a.Lin) own_hi = a.Lin;
    // All row loads of this thread are issued before the first one is consumed (a
    // load -> transform -> LDS-write loop would pay one HBM round trip per iteration).
    constexpr int NIT = ((C::BMB - 1) * SA + 1 + SA_MAX_HALO + C::RPPI - 1) / C::RPPI;
    uint4 raw[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int r = r0 + i * C::RPPI, g = gbase + r;
      raw[i] = make_uint4(0, 0, 0, 0);
      if (!(SA_ABL & 32) || a.B < 0)
      if (r < a.nrows && g >= 0 && g < a.Lin) raw[i] = *reinterpret_cast<const uint4*>(xb + (size_t)g * CIN);
    }
    // PRO2 (normalisation-backward prologue, SaConvArgs.nb_*): the rows are d z of the layer above;
    // d y = c1*dz + c2*y + c3 [* (y > 0)] is formed here from the stored forward tensor y instead
    // of in a separate sa_ew_apply pass over HBM
    uint4 raw2[PRO2 ? NIT : 1];
    float k1[VEC], k2[VEC], k3[VEC], csum[VEC];
    if constexpr (PRO2) {
      const S* x2 = reinterpret_cast<const S*>(a.nb_x) + (size_t)b * a.Lin * CIN + c * VEC;
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        const int r = r0 + i * C::RPPI, g = gbase + r;
        raw2[i] = make_uint4(0, 0, 0, 0);
        if (!(SA_ABL & 32) || a.B < 0)
        if (r < a.nrows && g >= 0 && g < a.Lin) raw2[i] = *reinterpret_cast<const uint4*>(x2 + (size_t)g * CIN);
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const size_t q = (size_t)b * a.nb_bstride + c * VEC + j;
        k1[j] = a.nb_c1[q]; k2[j] = a.nb_c2[q]; k3[j] = a.nb_c3[q]; csum[j] = 0.0f;
      }
    }
#ifdef SA_CONV_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SA_STAMP(1);                                      // all row loads have landed
#endif
    // The transform is selected by kernel arguments, i.e. uniformly: one specialised copy of the
    // staging loop per case (none: data gradients and ConvT forwards, 13 of 22 launches per step;
    // affine + x*sigmoid(x): the encoder / decoder forwards; generic) instead of per-element
    // selects on the flags.
    auto stage_rows = [&](auto xform) {
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        const int r = r0 + i * C::RPPI, g = gbase + r;
        if (r < a.nrows) {
          float f[VEC];
          tr::unpack(raw[i], f);
          if (g >= 0 && g < a.Lin) xform(f, i, g);
          LT* dst = As + (size_t)r * C::APITCH + c * VEC;
          if constexpr (P::NPL == 2) {
            uint2 hi, lo;
            sa_split4(f, hi, lo);
            *reinterpret_cast<uint2*>(dst) = hi;
            *reinterpret_cast<uint2*>(dst + plane) = lo;
          } else if constexpr (sizeof(LT) == 1) {
            *reinterpret_cast<uint2*>(dst) = sa_pack_fp8x8(f);
          } else if constexpr (sizeof(LT) == 2 && sizeof(S) == 4) {
            *reinterpret_cast<uint2*>(dst) = sa_pack_bf16x4(f);
          } else if constexpr (sizeof(LT) == 2) {
            *reinterpret_cast<uint4*>(dst) = tr::pack(f);
          } else {
            float* d = reinterpret_cast<float*>(dst);
            d[0] = f[0]; d[1] = f[1]; d[2] = f[2]; d[3] = f[3];
          }
        }
      }
    };
    if ((SA_ABL & 8) && a.B > 0) {
      stage_rows([](float*, int, int) {});
    } else if constexpr (PRO2) {
      stage_rows([&](float* f, int i, int g) {
        float y[VEC];
        tr::unpack(raw2[i], y);
        const bool own = g >= own_lo && g < own_hi;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          float v = fmaf(k1[j], f[j], fmaf(k2[j], y[j], k3[j]));
          if (a.nb_relu_mask && !(y[j] > 0.0f)) v = 0.0f;
          f[j] = v;
          if (own) csum[j] += v;
        }
      });
      if (a.nb_colsum) {
        // column sums of d y over the owned rows (bias gradient of the layer below): lanes
        // holding the same channel chunk are folded by shuffles, then one LDS slot per wave
        float* colred = reinterpret_cast<float*>(smem + col_off);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          float v = csum[j];
          for (int off = C::CHI; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
          if (lane < C::CHI) colred[wave * CIN + (lane % C::CHI) * VEC + j] = v;
        }
      }
    } else if (!has1 && !has2 && !sw) {
      stage_rows([](float*, int, int) {});
    } else if (has1 && sw && !has2 && a.pro_stats) {
      // forward launch that also leaves (sum, sum of squares) of its transformed input rows, per
      // tile over the owned rows: the statistics of the activation for a BatchNorm that reads
      // the same tensor (the classifier's input BatchNorm; saves a pass over the encoder output)
      float ps[VEC], pq[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) { ps[j] = 0.0f; pq[j] = 0.0f; }
      stage_rows([&](float* f, int, int g) {
        const bool own = g >= own_lo && g < own_hi;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float v = sa_swish(fmaf(f[j], s1[j], t1[j]));
          f[j] = v;
          if (own) { ps[j] += v; pq[j] = fmaf(v, v, pq[j]); }
        }
      });
      float* colred = reinterpret_cast<float*>(smem + col_off);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float v = ps[j], w = pq[j];
        for (int off = C::CHI; off < 64; off <<= 1) { v += __shfl_xor(v, off, 64); w += __shfl_xor(w, off, 64); }
        if (lane < C::CHI) {
          colred[(wave * CIN + (lane % C::CHI) * VEC + j) * 2 + 0] = v;
          colred[(wave * CIN + (lane % C::CHI) * VEC + j) * 2 + 1] = w;
        }
      }
    } else if (has1 && sw && !has2) {
      stage_rows([&](float* f, int, int) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) f[j] = sa_swish(fmaf(f[j], s1[j], t1[j]));
      });
    } else {
      stage_rows([&](float* f, int, int) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          float v = f[j];
          if (has1) v = fmaf(v, s1[j], t1[j]);
          if (sw) v = sa_swish(v);
          if (has2) v = fmaf(v, s2[j], t2[j]);
          f[j] = v;
        }
      });
    }
  }
  __syncthreads();
  SA_STAMP(2);
  // activation cache for sa_wgrad (x_pre): the bf16 (hi) plane of the rows this tile owns -- its
  // base rows, the last tile also the trailing halo -- goes out in 16-byte pieces; the stores
  // drain while the MFMA loop runs
  if (!PRO2 && a.pro_stats && tid < CIN) {
    const float* colred = reinterpret_cast<const float*>(smem + col_off);
    float* d = a.pro_stats + (((size_t)b * a.ntiles + tile) * CIN + tid) * 2;
#pragma unroll
    for (int q = 0; q < 2; ++q)
      d[q] = (colred[tid * 2 + q] + colred[(CIN + tid) * 2 + q]) +
             (colred[(2 * CIN + tid) * 2 + q] + colred[(3 * CIN + tid) * 2 + q]);
  }
  if constexpr (PRO2) {
    if (a.nb_colsum && tid < CIN) {
      const float* colred = reinterpret_cast<const float*>(smem + col_off);
      a.nb_colsum[((size_t)b * a.ntiles + tile) * CIN + tid] =
          (colred[tid] + colred[CIN + tid]) + (colred[2 * CIN + tid] + colred[3 * CIN + tid]);
    }
  }
  if constexpr (sizeof(LT) == 2) {
    if (a.a_out) {
      constexpr int CH16 = CIN / 8;
      const int own_lo = m0 * SA;
      int own_hi = tile == a.ntiles - 1 ? a.Lin : (m0 + C::BMB) * SA;
      if (own_hi > a.Lin) own_hi = a.Lin;
      const int rlo = own_lo - (m0 * SA + a.rowmin);
      bf16_t* ao = reinterpret_cast<bf16_t*>(a.a_out) + ((size_t)b * a.Lin + own_lo) * CIN;
      for (int e = tid; e < (own_hi - own_lo) * CH16; e += 256) {
        const int r = e / CH16, c = e % CH16;
        *reinterpret_cast<uint4*>(ao + (size_t)r * CIN + c * 8) =
            *reinterpret_cast<const uint4*>(As + (size_t)(rlo + r) * C::APITCH + c * 8);
      }
    }
  }

  // ---------------- main loop: MFMA over taps x channels -------------------------
  const int wn = wave % C::WN, wm = wave / C::WN;
  f32x16 acc[C::VPW][C::MT];
#pragma unroll
  for (int v = 0; v < C::VPW; ++v)
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[v][mt][i] = 0.0f;

  // Weight fragments (B operand) come straight from global memory (L2-resident image); they are
  // software-pipelined one group of KU k-steps ahead in registers so their latency hides
  // behind the previous group's MFMAs (two statically indexed register buffers, ping-pong).
  constexpr int KUP = P::NPL == 2 ? 2 : 8;
  constexpr int KU = C::KSTEPS < KUP ? C::KSTEPS : KUP;
  constexpr int GPT = C::KSTEPS / KU;                // groups per tap
  static_assert(C::KSTEPS % KU == 0, "k-steps per tap must be a multiple of the prefetch group");
  const Frag* wp = reinterpret_cast<const Frag*>(a.wp);
#pragma unroll
  for (int v = 0; v < C::VPW; ++v) {
    const int vt = wn + v * C::WN;
    const int ph = vt / C::NT, nt = vt % C::NT;
    const int G = a.taps.ntaps[ph] * GPT;
    const LT* abase = As + (size_t)((wm * C::MT * 32 + (lane & 31)) * SA - a.rowmin) * C::APITCH
                      + (lane >> 5) * (C::KS / 2);
    auto load_group = [&](Frag (&dst)[P::NPL][KU], int g) {
      const int ti = g / GPT, kg = g % GPT;
      const Frag* wt = wp + (((size_t)a.taps.widx[ph][ti] * C::KSTEPS + kg * KU) * C::NT + nt) * 64 + lane;
#pragma unroll
      for (int ku = 0; ku < KU; ++ku) {
        dst[0][ku] = wt[(size_t)ku * C::NT * 64];
        if constexpr (P::NPL == 2) dst[1][ku] = wt[(size_t)a.wlo_off + (size_t)ku * C::NT * 64];
      }
    };
    auto compute_group = [&](const Frag (&bq)[P::NPL][KU], int g) {
      const int ti = g / GPT, kg = g % GPT;
      const LT* arow = abase + (size_t)a.taps.off[ph][ti] * C::APITCH + kg * KU * C::KS;
      // A fragments of k-step ku+1 are requested from LDS before the MFMAs of k-step ku issue
      Frag ah[2][C::MT], al[2][C::MT];
      auto load_a = [&](int slot, int ku) {
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt) {
          const LT* ap = arow + (size_t)mt * 32 * SA * C::APITCH + ku * C::KS;
          ah[slot][mt] = *reinterpret_cast<const Frag*>(ap);
          if constexpr (P::NPL == 2) al[slot][mt] = *reinterpret_cast<const Frag*>(ap + plane);
        }
      };
      load_a(0, 0);
#pragma unroll
      for (int ku = 0; ku < KU; ++ku) {
        if (ku + 1 < KU && (!(SA_ABL & 16) || a.B < 0)) load_a((ku + 1) & 1, ku + 1);
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt) {
          if constexpr ((SA_ABL & 1) != 0) {
            asm volatile("" :: "v"(ah[ku & 1][mt]), "v"(bq[0][ku]));
            if constexpr (P::NPL == 2) asm volatile("" :: "v"(al[ku & 1][mt]), "v"(bq[1][ku]));
          } else if constexpr (P::NPL == 2) {
            acc[v][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ku & 1][mt], bq[0][ku], acc[v][mt], 0, 0, 0);
            acc[v][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ku & 1][mt], bq[1][ku], acc[v][mt], 0, 0, 0);
            acc[v][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ku & 1][mt], bq[0][ku], acc[v][mt], 0, 0, 0);
          } else {
            acc[v][mt] = Tr<LT>::mfma(ah[ku & 1][mt], bq[0][ku], acc[v][mt]);
          }
        }
      }
    };
    Frag b0[P::NPL][KU], b1[P::NPL][KU];
    load_group(b0, 0);
    if ((SA_ABL & 2) && a.B > 0) load_group(b1, 1);
    for (int g = 0; g < G; g += 2) {
      if (g + 1 < G && (!(SA_ABL & 2) || a.B < 0)) load_group(b1, g + 1);
      compute_group(b0, g);
      if (g + 1 < G) {
        if (g + 2 < G && (!(SA_ABL & 2) || a.B < 0)) load_group(b0, g + 2);
        compute_group(b1, g + 1);
      }
    }
  }
  SA_STAMP(3);
  if ((SA_ABL & 4) && a.B > 0) {
    float t = 0.0f;
#pragma unroll
    for (int v = 0; v < C::VPW; ++v)
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) t += acc[v][mt][i];
    if (t == 1.2345e-33f) reinterpret_cast<float*>(a.y)[tid] = t;       // keeps the accumulators live
    return;
  }
  constexpr int NOT = TM / C::RPPO;
  const int ec = tid % C::CHO, er0 = tid / C::CHO;
  __syncthreads();                                   // every wave is done reading As
  SA_STAMP(4);

  // ---------------- epilogue: bias/ReLU -> LDS transpose -> coalesced store ------
#pragma unroll
  for (int v = 0; v < C::VPW; ++v) {
    const int vt = wn + v * C::WN;
    const int ph = vt / C::NT, nt = vt % C::NT;
    const int col = nt * 32 + (lane & 31);
    const float bv = a.bias ? a.bias[col] : 0.0f;
    float winv = 1.0f;                                  // SA_FP8: undo the weight image's scale
    if constexpr (sizeof(LT) == 1) winv = a.wscale ? 1.0f / a.wscale[0] : 1.0f;
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = wm * C::MT * 32 + mt * 32 + sa_acc_row(i, lane);
        float val = sizeof(LT) == 1 ? fmaf(acc[v][mt][i], winv, bv) : acc[v][mt][i] + bv;
        if (a.relu) val = fmaxf(val, 0.0f);
        Os[(size_t)(m * U + ph) * C::OPITCH + col] = tr::from_f(val);
      }
    }
  }
  __syncthreads();
  SA_STAMP(5);
  {
    const int c = ec, r0 = er0;
    float ssum[OVEC], ssq[OVEC], es1[OVEC], et1[OVEC], emu[OVEC], ers[OVEC];
#pragma unroll
    for (int j = 0; j < OVEC; ++j) {
      ssum[j] = 0.0f; ssq[j] = 0.0f;
      es1[j] = (a.ep_mode && a.ep_s1) ? a.ep_s1[(size_t)b * COUT + c * OVEC + j] : 1.0f;
      et1[j] = (a.ep_mode && a.ep_t1) ? a.ep_t1[(size_t)b * COUT + c * OVEC + j] : 0.0f;
      emu[j] = (a.ep_mode && a.ep_mean) ? a.ep_mean[(size_t)b * a.ep_bstride + c * OVEC + j] : 0.0f;
      ers[j] = (a.ep_mode && a.ep_rstd) ? a.ep_rstd[(size_t)b * a.ep_bstride + c * OVEC + j] : 1.0f;
    }
    // ep_g2k*: the second gradient is itself a pending BatchNorm-backward apply over the activation
    // (the classifier's input BatchNorm behind GradReverse): g2 <- k1*g2 + k2*swish(z) + k3
    const bool g2k = a.ep_mode && a.ep_g2 && a.ep_g2k1;
    float gk1[OVEC], gk2[OVEC], gk3[OVEC];
#pragma unroll
    for (int j = 0; j < OVEC; ++j) {
      gk1[j] = g2k ? a.ep_g2k1[c * OVEC + j] : 1.0f;
      gk2[j] = g2k ? a.ep_g2k2[c * OVEC + j] : 0.0f;
      gk3[j] = g2k ? a.ep_g2k3[c * OVEC + j] : 0.0f;
    }
    S* yb = reinterpret_cast<S*>(a.y) + (size_t)b * a.Lout * COUT + c * OVEC;
    const S* xe = a.ep_mode ? reinterpret_cast<const S*>(a.ep_x) + (size_t)b * a.Lout * COUT + c * OVEC : nullptr;
    const S* ge = (a.ep_mode && a.ep_g2) ? reinterpret_cast<const S*>(a.ep_g2) + (size_t)b * a.Lout * COUT + c * OVEC : nullptr;
    const int o0 = m0 * U;
    // one row group of the fused backward epilogue (shared by the two loops below)
    auto ep_rows = [&](uint4 u, const uint4& epx, const uint4& epg, int o) {
      // mode 1: g' = (g + g2) * swish'(z), xhat from x (InstanceNorm + x*sigmoid(x) block)
      // mode 2: g' = g + g2, xhat from x, or from swish(z) when ep_xp_is_act (BatchNorm blocks)
      float g[OVEC], x[OVEC], g2[OVEC], xn[OVEC];
      tr::unpack(u, g); tr::unpack(epx, x); tr::unpack(epg, g2);
#pragma unroll
      for (int j = 0; j < OVEC; ++j) {
        const float z = fmaf(x[j], es1[j], et1[j]);
        if (g2k) g2[j] = fmaf(gk1[j], g2[j], fmaf(gk2[j], sa_swish(z), gk3[j]));
        float gg = g[j] + g2[j];
        if (a.ep_mode == 1) gg *= sa_swish_grad(z);
        const float xv = a.ep_xp_is_act ? sa_swish(z) : x[j];
        xn[j] = (xv - emu[j]) * ers[j];
        g[j] = gg;
      }
      u = tr::pack(g);
      *reinterpret_cast<uint4*>(yb + (size_t)o * COUT) = u;
      if (a.stats) {
        tr::unpack(u, g);
#pragma unroll
        for (int j = 0; j < OVEC; ++j) { ssum[j] += g[j]; ssq[j] = fmaf(g[j], xn[j], ssq[j]); }
      }
    };
    if (a.ep_mode && o0 + TM <= a.Lout) {
      // Full tile: the stored forward tensor (and the optional second gradient) of row group i+2 are
      // requested before group i is processed.  In the general loop below every load sits behind
      // the previous group's store (y may alias them as far as hipcc knows) and behind a per-thread
      // bounds branch: one 16-byte load in flight per thread, and the 3-4 resident workgroups wait
      // out an HBM round trip per row group together (ablation: the epilogue cost 104 of 453 us).
      // Two groups ahead is 16 registers; all of them ahead would be 64 and a resident workgroup.
      auto run = [&](auto has_g2_c) {
        constexpr bool HG = decltype(has_g2_c)::value;
        constexpr int D = NOT < 2 ? NOT : 2;
        uint4 px[D], pg[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
          px[d] = *reinterpret_cast<const uint4*>(xe + (size_t)(o0 + r0 + d * C::RPPO) * COUT);
          pg[d] = make_uint4(0, 0, 0, 0);
          if constexpr (HG) pg[d] = *reinterpret_cast<const uint4*>(ge + (size_t)(o0 + r0 + d * C::RPPO) * COUT);
        }
#pragma unroll
        for (int i = 0; i < NOT; ++i) {
          const int r = r0 + i * C::RPPO, o = o0 + r;
          const uint4 epx = px[i % D], epg = pg[i % D];
          if (i + D < NOT) {
            px[i % D] = *reinterpret_cast<const uint4*>(xe + (size_t)(o + D * C::RPPO) * COUT);
            if constexpr (HG) pg[i % D] = *reinterpret_cast<const uint4*>(ge + (size_t)(o + D * C::RPPO) * COUT);
          }
          const uint4 u = *reinterpret_cast<const uint4*>(Os + (size_t)r * C::OPITCH + c * OVEC);
          ep_rows(u, epx, epg, o);
        }
      };
      if (ge) run(std::true_type{}); else run(std::false_type{});
    } else {
#pragma unroll
    for (int i = 0; i < NOT; ++i) {
      const int r = r0 + i * C::RPPO, o = o0 + r;
      if (o < a.Lout) {
        uint4 u = *reinterpret_cast<const uint4*>(Os + (size_t)r * C::OPITCH + c * OVEC);
        if (a.ep_mode) {
          const uint4 epx = *reinterpret_cast<const uint4*>(xe + (size_t)o * COUT);
          const uint4 epg = ge ? *reinterpret_cast<const uint4*>(ge + (size_t)o * COUT) : make_uint4(0, 0, 0, 0);
          ep_rows(u, epx, epg, o);
        } else {
          *reinterpret_cast<uint4*>(yb + (size_t)o * COUT) = u;
          if (a.stats) {
            float f[OVEC];
            tr::unpack(u, f);
#pragma unroll
            for (int j = 0; j < OVEC; ++j) { ssum[j] += f[j]; ssq[j] = fmaf(f[j], f[j], ssq[j]); }
          }
        }
      }
    }
    }
    if (a.stats) {
      __syncthreads();                               // Os fully consumed: red overlays it
#pragma unroll
      for (int j = 0; j < OVEC; ++j) {
        red[((size_t)r0 * COUT + c * OVEC + j) * 2 + 0] = ssum[j];
        red[((size_t)r0 * COUT + c * OVEC + j) * 2 + 1] = ssq[j];
      }
    }
  }
  SA_STAMP(6);
  SA_STAMP_RT(7);
  if (a.stats) {
    __syncthreads();
    if (tid < COUT) {
      float s = 0.0f, q = 0.0f;
      for (int r = 0; r < C::RPPO; ++r) {
        s += red[((size_t)r * COUT + tid) * 2 + 0];
        q += red[((size_t)r * COUT + tid) * 2 + 1];
      }
      float* dst = a.stats + (((size_t)b * a.ntiles + tile) * COUT + tid) * 2;
      dst[0] = s; dst[1] = q;
    }
  }
}

template <typename T, int CIN, int COUT, int SA, int U, int TM, bool PRO2 = false>
static int launch_cfg(const SaConvArgs& a, hipStream_t st) {
  typedef ConvCfg<T, CIN, COUT, SA, U, TM> C;
  SaConvArgs args = a;
  args.ntiles = sa_div_up(sa_div_up(a.Lout, U), C::BMB);
  // host-side shape check: every LDS row the main loop touches must be staged
  int omin = 1 << 30, omax = -(1 << 30);
  for (int ph = 0; ph < U; ++ph)
    for (int t = 0; t < a.taps.ntaps[ph]; ++t) {
      omin = a.taps.off[ph][t] < omin ? a.taps.off[ph][t] : omin;
      omax = a.taps.off[ph][t] > omax ? a.taps.off[ph][t] : omax;
    }
  if (omin > omax || omax - omin > SA_MAX_HALO) return -22;
  int wmax = 0;
  for (int ph = 0; ph < U; ++ph)
    for (int t = 0; t < a.taps.ntaps[ph]; ++t) wmax = a.taps.widx[ph][t] > wmax ? a.taps.widx[ph][t] : wmax;
  args.rowmin = omin;
  args.nrows = (C::BMB - 1) * SA + (omax - omin) + 1;
  args.wlo_off = (wmax + 1) * C::KSTEPS * C::NT * 64;      // Frag units: hi image size
  // a_out: every input row must be staged by the tile that owns it
  if ((a.a_out || a.nb_colsum || a.pro_stats) &&
      (omin > 0 || (C::BMB - 1) * SA + omax < C::BMB * SA - 1 ||
       (args.ntiles - 1) * C::BMB * SA + omin + args.nrows < a.Lin))
    return -22;
  if (a.a_out && sizeof(typename C::LT) != 2) return -22;
  if (PRO2 && (!a.nb_c1 || !a.nb_c2 || !a.nb_c3)) return -22;
  if (a.pro_stats && (PRO2 || !a.s1 || !a.swish || a.s2)) return -22;   // affine + x*sigmoid(x) launches only
  const size_t tile_lds = C::lds_bytes(args.nrows);
  const size_t lds = tile_lds + (PRO2 ? 4 * CIN * sizeof(float) : a.pro_stats ? 8 * CIN * sizeof(float) : 0);
  if (lds > 160 * 1024) return -12;
  auto kern = sa_conv_gemm_kernel<T, CIN, COUT, SA, U, TM, PRO2>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return -(int)e;
    attr_set = true;
  }
  dim3 grid(args.ntiles, a.B);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, args, 0, (int)tile_lds);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// Output rows per workgroup, chosen per shape from measurements at the training sizes (B=32,
// 161k-sample utterances): 64-row tiles keep 4-5 workgroups resident per CU so that one
// workgroup's HBM staging / epilogue overlaps another's MFMA phase; 128-row tiles halve the
// weight-fragment traffic per output row and win where the 128-row kernel still fits three
// workgroups per CU (64->64 k5: 301 -> 255 us, 128->64 U=2: 316 -> 301 us), lose where registers
// and LDS leave two (128->128: 469 -> 520 us, 64->128 s2: 264 -> 368 us).  The 64->32 U=2 layer
// needs 128 rows to give every wave a full 32-row m-tile.  sa_conv_gemm_set_tile_rows() is a
// tuning knob (0 = this policy).
static int g_tile_rows = 0;
extern "C" int sa_conv_gemm_set_tile_rows(int rows) {
  if (rows != 0 && rows != 64 && rows != 128) return -22;
  g_tile_rows = rows;
  return 0;
}
static int tile_rows(int cin, int cout, int u) {
  if (cin == 64 && cout == 32 && u == 2) return 128;
  if (g_tile_rows) return g_tile_rows;
  if ((cin == 64 && cout == 64) || (cin == 128 && cout == 64 && u == 2)) return 128;
  return 64;
}

// number of (sum, sumsq) partial tiles per utterance the epilogue writes (one-tile-per-workgroup kernel)
extern "C" int sa_conv_gemm_ntiles(int cin, int cout, int u, int Lout) {
  const int tm = tile_rows(cin, cout, u);
  return sa_div_up(sa_div_up(Lout, u), tm / u);
}

// ---- implementation choice: sa_conv_gemm_set_impl(1) routes the f32 and bf16x3 policies to the
// two-groups-in-anti-phase kernel (sa_conv_pp.hip).  Default 0: on the round-2 measurements
// (profiles/r02_conv_structure_experiments.md) it ties the one-tile kernel on the forward launches
// (370 vs 370 us, 128->128, B = 32) and loses on the fused data gradients (614 vs 450 us).
int sa_conv_pp_dispatch(int dtype, int cin, int cout, int sa, int u, const SaConvArgs* a, hipStream_t st);
int sa_pp_tile_rows(int cin, int cout, int u);
int sa_pp_share(int cout);
// Kernel choice.  Default (2): the 128->128 and 64->64 bf16x3 launches the weight-stationary kernel covers
// (sa_conv_ws.hip: persistent, one wave per SIMD, weights in registers, rows by LDS-DMA, epilogue and
// transform in the MFMA loop's issue gaps) go there when the launch has at least six tiles per CU (below that the un-overlapped first tiles of
// every workgroup cost more than the rest gains: B = 4 at the training length loses, B = 6 wins) --
// 266 / 289 us against 330 / 367 us (plain / forward with cache + statistics), 190 against 265 us for
// the 3-tap layers, 190 against 255 us for 64->64 (B = 32, profiles/r02_conv_structure_experiments.md);
// same slab geometry as this file's 64-row tiles, so nothing else changes for the caller.
// sa_conv_gemm_set_impl(0): this file's kernel only; (1): the ping-pong kernel for f32 / bf16x3.
bool sa_conv_ws_covers(int dtype, int cin, int cout, int sa, int u, const SaConvArgs* a);
int sa_conv_ws_dispatch(int cin, int cout, const SaConvArgs* a, hipStream_t st);
int sa_conv_ws_tile_rows(int cout);
int sa_conv_ws_rows_per_tile(int cout);
// (3): the fused data gradients 128 -> 128 that sa_conv_wsd.hip covers (normalisation-backward prologue
// and / or fused backward epilogue) go to that kernel under the same conditions.
bool sa_conv_wsd_covers(int dtype, int cin, int cout, int sa, int u, const SaConvArgs* a);
int sa_conv_wsd_dispatch(int cin, int cout, const SaConvArgs* a, hipStream_t st);
static int g_use_pp = 0, g_use_ws = 1;
extern "C" int sa_conv_gemm_set_impl(int impl) {
  if (impl < 0 || impl > 2) return -22;
  g_use_pp = impl == 1;
  g_use_ws = impl == 2;
  return 0;
}
static bool uses_pp(int dtype) { return g_use_pp && (dtype == SA_F32 || dtype == SA_BF16X3); }

// slabs per utterance of nb_colsum / pro_stats ([B][*ntiles][cin](x2)) and of `stats`
// ([B][*nslabs][cout][2]) of a launch with this dtype code and shape
extern "C" int sa_conv_gemm_geometry(int dtype, int cin, int cout, int u, int Lout, int* ntiles, int* nslabs) {
  if (!ntiles || !nslabs || cin <= 0 || cout <= 0 || u <= 0 || Lout <= 0) return -22;
  if (uses_pp(dtype)) {
    const int tm = sa_pp_tile_rows(cin, cout, u);
    const int nt = sa_div_up(sa_div_up(Lout, u), tm / u);
    *ntiles = nt * 4;                      // nb_colsum / pro_stats: one slab per wave of the tile's group
    *nslabs = nt * sa_pp_share(cout);
  } else {
    *ntiles = sa_conv_gemm_ntiles(cin, cout, u, Lout);
    *nslabs = *ntiles;
  }
  return 0;
}

extern "C" int sa_conv_gemm_ntiles_tm(int tile_rows, int u, int Lout) {
  if ((tile_rows != 64 && tile_rows != 128) || u < 1 || Lout < 1) return -22;
  return sa_div_up(sa_div_up(Lout, u), tile_rows / u);
}

template <typename T, int CI, int CO, int S, int UU, bool PRO2 = false>
static int launch_tm(const SaConvArgs& a, hipStream_t st) {
  if constexpr (CI == 64 && CO == 32 && UU == 2) {
    return launch_cfg<T, CI, CO, S, UU, 128, PRO2>(a, st);
  } else {
    const int tm = (a.tile_rows == 64 || a.tile_rows == 128) ? a.tile_rows : tile_rows(CI, CO, UU);
    return tm == 64 ? launch_cfg<T, CI, CO, S, UU, 64, PRO2>(a, st)
                    : launch_cfg<T, CI, CO, S, UU, 128, PRO2>(a, st);
  }
}

// the normalisation-backward prologue (nb_x) is built for the bf16x3 and bf16 policies
#define SA_CONV_CASE(CI, CO, S, UU)                                              \
  if (cin == CI && cout == CO && sa == S && u == UU) {                           \
    if (a->nb_x)                                                                 \
      return dtype == SA_BF16X3 ? launch_tm<bf16x3_t, CI, CO, S, UU, true>(*a, st) \
             : dtype == SA_BF16 ? launch_tm<bf16_t, CI, CO, S, UU, true>(*a, st) : -22; \
    if (dtype == SA_FP8) return (a->wscale && !a->ep_mode) ? launch_tm<fp8_t, CI, CO, S, UU>(*a, st) : -22; \
    return dtype == SA_BF16 ? launch_tm<bf16_t, CI, CO, S, UU>(*a, st)           \
           : dtype == SA_BF16X3 ? launch_tm<bf16x3_t, CI, CO, S, UU>(*a, st)     \
           : dtype == SA_BF16X1F ? launch_tm<bf16x1f_t, CI, CO, S, UU>(*a, st)   \
                                : launch_tm<float, CI, CO, S, UU>(*a, st);       \
  }

// sizeof of the argument records, for bindings to check their mirror of include/sa_hip.h
extern "C" int sa_abi_sizeof(int which) {
  switch (which) {
    case 0: return (int)sizeof(SaConvArgs);
    case 1: return (int)sizeof(SaWgradArgs);
    case 2: return (int)sizeof(SaEwArgs);
    case 3: return (int)sizeof(SaPackDesc);
    case 4: return (int)sizeof(SaTaps);
    case 5: return (int)sizeof(SaFinArgs);
    case 6: return (int)sizeof(SaBiasMulti);
    case 7: return (int)sizeof(SaWredMulti);
    default: return -22;
  }
}

// which kernel sa_conv_gemm routes this launch to: 0 one-tile, 1 ping-pong, 2 weight-stationary, 3 weight-stationary fused data gradient
static int conv_route(int dtype, int cin, int cout, int sa, int u, const SaConvArgs* a) {
  if (uses_pp(dtype)) return 1;
  if (g_use_ws && sa_conv_ws_covers(dtype, cin, cout, sa, u, a) &&
      tile_rows(cin, cout, u) == sa_conv_ws_tile_rows(cout) &&
      (long)a->B * sa_div_up(a->Lout, sa_conv_ws_rows_per_tile(cout)) >= 1536)
    return 2;
  if (g_use_ws && sa_conv_wsd_covers(dtype, cin, cout, sa, u, a) && tile_rows(cin, cout, u) == 64 &&
      (long)a->B * sa_div_up(a->Lout, 64) >= 1536)
    return 3;
  return 0;
}
extern "C" int sa_conv_gemm_route(int dtype, int cin, int cout, int sa, int u, const SaConvArgs* a) {
  if (!a || a->B <= 0 || a->Lout <= 0) return -22;
  return conv_route(dtype, cin, cout, sa, u, a);
}

// C-ABI entry (see include/sa_hip.h).  Returns 0, or a negative hipError_t / errno.
extern "C" int sa_conv_gemm(int dtype, int cin, int cout, int sa, int u, const SaConvArgs* a,
                            void* stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!a || !a->x || !a->wp || !a->y || a->B <= 0 || a->Lin <= 0 || a->Lout <= 0) return -22;
  if (a->ep_mode < 0 || a->ep_mode > 2 || (a->ep_mode && !a->ep_x)) return -22;
  const int route = conv_route(dtype, cin, cout, sa, u, a);
  if (route == 1) return sa_conv_pp_dispatch(dtype, cin, cout, sa, u, a, st);
  if (route == 2) return sa_conv_ws_dispatch(cin, cout, a, st);
  if (route == 3) return sa_conv_wsd_dispatch(cin, cout, a, st);
  SA_CONV_CASE(32, 64, 2, 1)
  SA_CONV_CASE(64, 64, 1, 1)
  SA_CONV_CASE(64, 128, 2, 1)
  SA_CONV_CASE(128, 128, 1, 1)
  SA_CONV_CASE(128, 64, 1, 2)
  SA_CONV_CASE(64, 32, 1, 2)
  return -38;                   // ENOSYS: shape not instantiated
}
