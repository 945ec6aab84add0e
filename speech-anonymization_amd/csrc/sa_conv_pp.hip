// Implicit-GEMM 1-D convolution on MFMA (gfx950), channels-last -- the "ping-pong" form.
//
// Same operation, arguments and numerics policy as sa_conv_gemm.hip (it replaces the cuDNN/ATen
// conv1d / conv_transpose1d forward and data gradients the reference reaches from
// models/ConvAutoEncoder.py:141-172 and :33-43); different execution structure:
//
// Measured on the one-tile-per-workgroup kernel (tools/conv_ablate.py, 128->128, B = 32): removing
// the MFMAs saves 122 us, the row loads 76 us, the weight-fragment loads 61 us, the epilogue 37 us
// of 332 us -- the parts add up, i.e. they do not overlap.  The 3-4 workgroups resident on a CU
// fall into lock-step (they contend for the same unit in every phase, so a lagging one catches up
// as soon as the leader leaves that phase) and the CU alternates between "all loading", "all
// transforming", "all on the matrix pipe", "all storing".
//
// Here a workgroup has TWO groups of 4 waves that work on two different tiles in ANTI-PHASE, held
// there by one workgroup barrier per half-step:
//     half-step h     group A: MFMA loop of its tile k         group B: epilogue of its tile k-1
//                                                                       + row loads, transform and
//                                                                       LDS staging of its tile k
//     half-step h+1   group A: epilogue(k), load/stage(k+1)    group B: MFMA loop(k)
// so on every SIMD one wave feeds the matrix pipe while its partner streams HBM and does the VALU
// work.  Each group owns its own LDS tile buffer; nothing is exchanged between the groups.  A
// workgroup walks a contiguous range of tiles (grid = resident workgroups of the chip).
//
// The epilogue works from the accumulator registers (no LDS transpose, no barrier): one register
// of a 32x32 accumulator is two 128-byte row segments across the wave, stored as it stands; the
// per-(utterance, channel) statistics are per-lane sums (a lane owns one column) folded once across
// the two lane halves.  Waves that share a column block write separate statistics slabs
// (SHARE per tile), summed in fixed order by the same reducers as before.
//
// Policies f32 and bf16x3 (fp32 storage); bf16 storage and bf16x1f stay on sa_conv_gemm.hip.
#include <type_traits>
#include "sa_conv_cfg.h"
// -DSA_ABL=<mask>: timing-only ablation builds (WRONG numerics, never shipped): 1 no MFMA, 2 weight
// fragments loaded once, 4 no epilogue stores, 16 A fragments loaded once, 32 no row loads, 64 no a_out
#ifndef SA_ABL
#define SA_ABL 0
#endif

// -DSA_PP_STAMPS: diagnostic build (tools/pp_stamps.py): s_memtime at the phase boundaries of one
// workgroup (wave 0 of each group); no stamp exists in the normal build.
#ifdef SA_PP_STAMPS
__device__ unsigned long long sa_pp_dbg[2 * 64 * 8];
#define PP_STAMP(hs, i) do { if (lane == 0 && wave == 0 && sub == 0 && blockIdx.x == 7 && (hs) < 64) { \
  unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
  sa_pp_dbg[(grp * 64 + (hs)) * 8 + (i)] = t_; } } while (0)   /* wave 0 of sub-group 0 of each phase group */
extern "C" int sa_pp_dbg_read(unsigned long long* out) {
  return -(int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sa_pp_dbg), sizeof(sa_pp_dbg));
}
#else
#define PP_STAMP(hs, i)
#endif

template <typename T, int CIN, int COUT, int SA, int U, int TM>
struct PPCfg : ConvCfg<T, CIN, COUT, SA, U, TM> {
  typedef ConvCfg<T, CIN, COUT, SA, U, TM> C;
  static_assert(C::VPW == 1, "one virtual n-tile per wave");
  static_assert(sizeof(typename C::S) == 4, "fp32 storage");
  static constexpr int SHARE = 4 / C::NT;           // waves of a group that share a column block
  static size_t buf_bytes(int nrows) {               // one group's operand tile
    size_t a = (size_t)C::P::NPL * nrows * C::APITCH * sizeof(typename C::LT);
    return (a + 15) & ~(size_t)15;
  }
};

// NG = 4-wave sub-groups per phase group: the workgroup has 2*NG sub-groups (512*NG threads), each
// with its own tile and LDS buffer; NG of them are in the MFMA half while the other NG are in the
// memory half.  NG = 2 with 64-row tiles puts TWO waves per SIMD on the matrix pipe (they cover
// each other's LDS / weight-fragment latencies) at the 128-register budget of 4 waves per SIMD.
template <typename T, int CIN, int COUT, int SA, int U, int TM, bool PRO2, int NG>
__global__ __launch_bounds__(512 * NG, TM == 64 ? 4 : 2 * NG) void sa_conv_pp_kernel(SaConvArgs a, int tiles_per_wg,
                                                                     int total_tiles, int buf_bytes,
                                                                     int col_off) {
  typedef PPCfg<T, CIN, COUT, SA, U, TM> C;
  typedef Pol<T> P;
  typedef float S;
  typedef typename P::lds_t LT;
  typedef typename P::Frag Frag;
  typedef Tr<S> tr;
  constexpr int VEC = C::VEC;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, sg = tid >> 8, gt = tid & 255, lane = tid & 63, wave = gt >> 6;
  const int grp = sg / NG, sub = sg % NG;                   // phase group, sub-group within it
  LT* As = reinterpret_cast<LT*>(smem + (size_t)sg * buf_bytes);
  const int plane = a.nrows * C::APITCH;                    // lo plane offset (split mode)
  const int first = blockIdx.x * tiles_per_wg;
  int last = first + tiles_per_wg;
  if (last > total_tiles) last = total_tiles;
  const int wn = wave % C::WN, wm = wave / C::WN;
  const int ph = wn / C::NT, nt = wn % C::NT;               // this wave's output phase / column block

  f32x16 acc[C::MT];
  int cur_h = 0;                                            // (half-step index for the stamps build)
  (void)cur_h;

  // ================= load + transform + stage the input rows of tile t =================
  auto stage_tile = [&](int t) {
    const int b = t / a.ntiles, tile = t % a.ntiles, m0 = tile * C::BMB;
    // opaque per call: otherwise hipcc hoists the per-row global / LDS address chains of all NIT
    // rows (and of every staging variant) out of the half-step loop and keeps them in registers
    // for the whole kernel (100+ spilled VGPRs)
    int gtv = gt;
    asm volatile("" : "+v"(gtv));
    const int c = gtv % C::CHI, r0 = gtv / C::CHI;
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
    const S* x2 = PRO2 ? reinterpret_cast<const S*>(a.nb_x) + (size_t)b * a.Lin * CIN + c * VEC : nullptr;
    const int gbase = m0 * SA + a.rowmin;
    const int own_lo = m0 * SA;
    int own_hi = tile == a.ntiles - 1 ? a.Lin : (m0 + C::BMB) * SA;
    if (own_hi > a.Lin) own_hi = a.Lin;
    constexpr int NIT = ((C::BMB - 1) * SA + 1 + SA_MAX_HALO + C::RPPI - 1) / C::RPPI;
    // the rows go through registers in PASSES portions (all loads of a portion in flight at once;
    // the other group's MFMA phase covers the latency)
    constexpr int PASSES = (PRO2 && NIT > 6) ? 2 : 1;
    constexpr int NPP = (NIT + PASSES - 1) / PASSES;
    float k1[VEC], k2[VEC], k3[VEC], csum[VEC], ps[VEC], pq[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { csum[j] = 0.0f; ps[j] = 0.0f; pq[j] = 0.0f; k1[j] = 0.0f; k2[j] = 0.0f; k3[j] = 0.0f; }
    if constexpr (PRO2) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const size_t q = (size_t)b * a.nb_bstride + c * VEC + j;
        k1[j] = a.nb_c1[q]; k2[j] = a.nb_c2[q]; k3[j] = a.nb_c3[q];
      }
    }
    const int mode = PRO2 ? 0 : (!has1 && !has2 && !sw) ? 1 : (has1 && sw && !has2 && a.pro_stats) ? 2
                     : (has1 && sw && !has2) ? 3 : 4;
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
      uint4 raw[NPP];
      uint4 raw2[PRO2 ? NPP : 1];
#pragma unroll
      for (int ii = 0; ii < NPP; ++ii) {
        const int i = pass * NPP + ii;
        const int r = r0 + i * C::RPPI, g = gbase + r;
        raw[ii] = make_uint4(0, 0, 0, 0);
        if (!(SA_ABL & 32) || a.B < 0)
        if (i < NIT && r < a.nrows && g >= 0 && g < a.Lin) raw[ii] = *reinterpret_cast<const uint4*>(xb + (size_t)g * CIN);
      }
      if constexpr (PRO2) {
#pragma unroll
        for (int ii = 0; ii < NPP; ++ii) {
          const int i = pass * NPP + ii;
          const int r = r0 + i * C::RPPI, g = gbase + r;
          raw2[ii] = make_uint4(0, 0, 0, 0);
          if (!(SA_ABL & 32) || a.B < 0)
          if (i < NIT && r < a.nrows && g >= 0 && g < a.Lin) raw2[ii] = *reinterpret_cast<const uint4*>(x2 + (size_t)g * CIN);
        }
      }
#ifdef SA_PP_STAMPS
      if (pass == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); PP_STAMP(cur_h, 5); }
#endif
      auto stage_rows = [&](auto xform) {
#pragma unroll
        for (int ii = 0; ii < NPP; ++ii) {
          const int i = pass * NPP + ii;
          const int r = r0 + i * C::RPPI, g = gbase + r;
          if (i < NIT && r < a.nrows) {
            float f[VEC];
            tr::unpack(raw[ii], f);
            if (g >= 0 && g < a.Lin) xform(f, ii, g);
            LT* dst = As + (size_t)r * C::APITCH + c * VEC;
            if constexpr (P::NPL == 2) {
              uint2 hi, lo;
              sa_split4(f, hi, lo);
              *reinterpret_cast<uint2*>(dst) = hi;
              *reinterpret_cast<uint2*>(dst + plane) = lo;
            } else if constexpr (sizeof(LT) == 2) {
              *reinterpret_cast<uint2*>(dst) = sa_pack_bf16x4(f);
            } else {
              float* d = reinterpret_cast<float*>(dst);
              d[0] = f[0]; d[1] = f[1]; d[2] = f[2]; d[3] = f[3];
            }
          }
        }
      };
      // the transform is selected by kernel arguments, i.e. uniformly: one specialised staging loop per case
      if constexpr (PRO2) {
        // normalisation-backward prologue: d y = c1*dz + c2*y + c3 [* (y > 0)]
        stage_rows([&](float* f, int ii, int g) {
          float y[VEC];
          tr::unpack(raw2[ii], y);
          const bool own = g >= own_lo && g < own_hi;
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            float v = fmaf(k1[j], f[j], fmaf(k2[j], y[j], k3[j]));
            if (a.nb_relu_mask && !(y[j] > 0.0f)) v = 0.0f;
            f[j] = v;
            if (own) csum[j] += v;
          }
        });
      } else if (mode == 1) {
        stage_rows([](float*, int, int) {});
      } else if (mode == 2) {
        stage_rows([&](float* f, int, int g) {
          const bool own = g >= own_lo && g < own_hi;
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            const float v = sa_swish(fmaf(f[j], s1[j], t1[j]));
            f[j] = v;
            if (own) { ps[j] += v; pq[j] = fmaf(v, v, pq[j]); }
          }
        });
      } else if (mode == 3) {
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
    // by-products of the staging pass: per-WAVE partial sums, written as 4 slabs per tile (the
    // reducers downstream sum slabs in index order; no LDS, no barrier)
    if constexpr (PRO2) {
      if (a.nb_colsum) {
        float* dst = a.nb_colsum + (((size_t)b * a.ntiles + tile) * 4 + wave) * CIN;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          float v = csum[j];
          for (int off = C::CHI; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
          if (lane < C::CHI) dst[(lane % C::CHI) * VEC + j] = v;
        }
      }
    } else if (mode == 2) {
      float* dst = a.pro_stats + (((size_t)b * a.ntiles + tile) * 4 + wave) * CIN * 2;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float v = ps[j], w = pq[j];
        for (int off = C::CHI; off < 64; off <<= 1) { v += __shfl_xor(v, off, 64); w += __shfl_xor(w, off, 64); }
        if (lane < C::CHI) {
          dst[((lane % C::CHI) * VEC + j) * 2 + 0] = v;
          dst[((lane % C::CHI) * VEC + j) * 2 + 1] = w;
        }
      }
    }
  };

  // ================= MFMA loop over taps x channels of tile t (operands staged) =================
  auto mfma_tile = [&](int t) {
    const int b = t / a.ntiles, tile = t % a.ntiles, m0 = tile * C::BMB;
    if constexpr (sizeof(LT) == 2) {
      if (a.a_out && (!(SA_ABL & 64) || a.B < 0)) {               // bf16 operand cache for sa_wgrad: the (hi) plane of the owned rows
        constexpr int CH16 = CIN / 8;
        const int own_lo = m0 * SA;
        int own_hi = tile == a.ntiles - 1 ? a.Lin : (m0 + C::BMB) * SA;
        if (own_hi > a.Lin) own_hi = a.Lin;
        const int rlo = own_lo - (m0 * SA + a.rowmin);
        bf16_t* ao = reinterpret_cast<bf16_t*>(a.a_out) + ((size_t)b * a.Lin + own_lo) * CIN;
        for (int e = gt; e < (own_hi - own_lo) * CH16; e += 256) {
          const int r = e / CH16, cc = e % CH16;
          *reinterpret_cast<uint4*>(ao + (size_t)r * CIN + cc * 8) =
              *reinterpret_cast<const uint4*>(As + (size_t)(rlo + r) * C::APITCH + cc * 8);
        }
      }
    }
    constexpr int KUP = P::NPL == 2 ? 2 : 8;
    constexpr int KU = C::KSTEPS < KUP ? C::KSTEPS : KUP;
    constexpr int GPT = C::KSTEPS / KU;                // groups per tap
    static_assert(C::KSTEPS % KU == 0, "k-steps per tap must be a multiple of the prefetch group");
    const Frag* wp = reinterpret_cast<const Frag*>(a.wp);
    const int G = a.taps.ntaps[ph] * GPT;
    int lanev = lane;
    asm volatile("" : "+v"(lanev));                    // (see stage_tile)
    const LT* abase = As + (size_t)((wm * C::MT * 32 + (lanev & 31)) * SA - a.rowmin) * C::APITCH
                      + (lanev >> 5) * (C::KS / 2);
    auto load_group = [&](Frag (&dst)[P::NPL][KU], int g) {
      const int ti = g / GPT, kg = g % GPT;
      const Frag* wt = wp + (((size_t)a.taps.widx[ph][ti] * C::KSTEPS + kg * KU) * C::NT + nt) * 64 + lanev;
#pragma unroll
      for (int ku = 0; ku < KU; ++ku) {
        dst[0][ku] = wt[(size_t)ku * C::NT * 64];
        if constexpr (P::NPL == 2) dst[1][ku] = wt[(size_t)a.wlo_off + (size_t)ku * C::NT * 64];
      }
    };
    auto compute_group = [&](const Frag (&bq)[P::NPL][KU], int g) {
      const int ti = g / GPT, kg = g % GPT;
      const LT* arow = abase + (size_t)a.taps.off[ph][ti] * C::APITCH + kg * KU * C::KS;
      // A fragments rotate through 3 slots, requested from LDS two (k-step, m-tile) steps before the
      // MFMAs that use them (a whole double-buffered k-step of them would cost 64 registers at MT = 4)
      constexpr int NS = KU * C::MT;
      Frag ah[3], al[3];
      auto load_a = [&](int slot, int s) {
        const int ku = s / C::MT, mt = s % C::MT;
        const LT* ap = arow + (size_t)mt * 32 * SA * C::APITCH + ku * C::KS;
        ah[slot] = *reinterpret_cast<const Frag*>(ap);
        if constexpr (P::NPL == 2) al[slot] = *reinterpret_cast<const Frag*>(ap + plane);
      };
      load_a(0, 0);
      if (NS > 1) load_a(1, 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        if (s + 2 < NS && (!(SA_ABL & 16) || a.B < 0)) load_a((s + 2) % 3, s + 2);
        const int ku = s / C::MT, mt = s % C::MT;
        if constexpr ((SA_ABL & 1) != 0) {
          asm volatile("" :: "v"(ah[s % 3]), "v"(bq[0][ku]));
          if constexpr (P::NPL == 2) asm volatile("" :: "v"(al[s % 3]), "v"(bq[1][ku]));
        } else if constexpr (P::NPL == 2) {
          acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[s % 3], bq[0][ku], acc[mt], 0, 0, 0);
          acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s % 3], bq[1][ku], acc[mt], 0, 0, 0);
          acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s % 3], bq[0][ku], acc[mt], 0, 0, 0);
        } else {
          acc[mt] = Tr<LT>::mfma(ah[s % 3], bq[0][ku], acc[mt]);
        }
        // keep the source order (fragment reads two steps ahead of their MFMAs): left alone, hipcc
        // moves every read next to its use and waits lgkmcnt(0) in front of each MFMA
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // Weight fragments: two statically indexed register buffers, each refilled one group ahead of
    // its use.  The loop body is straight-line (no conditional loads: past the end the last group
    // is fetched again, harmlessly) so that hipcc emits counted s_waitcnt vmcnt(N) -- behind a
    // branch it falls back to vmcnt(0), which drains the prefetch it was meant to cover.
    Frag b0[P::NPL][KU], b1[P::NPL][KU];
    load_group(b0, 0);
    int g = 0;
    for (; g + 1 < G; g += 2) {
      // the scheduling barriers keep each refill AHEAD of the group it overlaps (hipcc otherwise
      // sinks the loads to just before their use, i.e. no prefetch at all)
      if (!(SA_ABL & 2) || a.B < 0 || g == 0) load_group(b1, g + 1);
      __builtin_amdgcn_sched_barrier(0);
      compute_group(b0, g);
      __builtin_amdgcn_sched_barrier(0);
      if (!(SA_ABL & 2) || a.B < 0) load_group(b0, g + 2 < G ? g + 2 : G - 1);
      __builtin_amdgcn_sched_barrier(0);
      compute_group(b1, g + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (g < G) compute_group(b0, g);                  // odd number of groups: the last one
  };

  // ================= epilogue of tile t, straight from the accumulator registers =================
  auto epilogue = [&](int t) {
    if ((SA_ABL & 4) && a.B > 0) {
      float tt = 0.0f;
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) tt += acc[mt][i];
      if (tt == 1.2345e-33f) reinterpret_cast<float*>(a.y)[tid] = tt;     // keeps the accumulators live
      return;
    }
    const int b = t / a.ntiles, tile = t % a.ntiles, m0 = tile * C::BMB;
    int lanev = lane;
    asm volatile("" : "+v"(lanev));                    // (see stage_tile)
    const int col = nt * 32 + (lanev & 31);
    const float bv = a.bias ? a.bias[col] : 0.0f;
    const bool epm = a.ep_mode != 0;
    const float es1 = (epm && a.ep_s1) ? a.ep_s1[(size_t)b * COUT + col] : 1.0f;
    const float et1 = (epm && a.ep_t1) ? a.ep_t1[(size_t)b * COUT + col] : 0.0f;
    const float emu = (epm && a.ep_mean) ? a.ep_mean[(size_t)b * a.ep_bstride + col] : 0.0f;
    const float ers = (epm && a.ep_rstd) ? a.ep_rstd[(size_t)b * a.ep_bstride + col] : 1.0f;
    const bool g2k = epm && a.ep_g2 && a.ep_g2k1;
    const float gk1 = g2k ? a.ep_g2k1[col] : 1.0f, gk2 = g2k ? a.ep_g2k2[col] : 0.0f, gk3 = g2k ? a.ep_g2k3[col] : 0.0f;
    float* yb = reinterpret_cast<float*>(a.y) + (size_t)b * a.Lout * COUT + col;
    const float* xe = epm ? reinterpret_cast<const float*>(a.ep_x) + (size_t)b * a.Lout * COUT + col : nullptr;
    const float* ge = (epm && a.ep_g2) ? reinterpret_cast<const float*>(a.ep_g2) + (size_t)b * a.Lout * COUT + col : nullptr;
    float ssum = 0.0f, ssq = 0.0f;
    // FULL: every row of the tile exists (all tiles of an utterance but, possibly, its last):
    // no per-element bounds checks -- a check per element puts every load / store behind its own
    // branch, and hipcc then waits vmcnt(0) in front of each
    auto body = [&](auto full_c, auto epm_c) {
      constexpr bool FULL = decltype(full_c)::value, EPM = decltype(epm_c)::value;
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt) {
        const int mb = m0 + wm * C::MT * 32 + mt * 32 + 4 * (lanev >> 5);
        if constexpr (EPM) {
          // the stored forward tensor (and the optional second gradient) at this lane's positions:
          // all 16 (32) loads of the m-tile are in flight before the first is used
#pragma unroll
          for (int h8 = 0; h8 < 2; ++h8) {
          float x[8], g2[8];
#pragma unroll
          for (int ii = 0; ii < 8; ++ii) {
            const int i = h8 * 8 + ii;
            const int o = (mb + (i & 3) + 8 * (i >> 2)) * U + ph;
            const bool ok = FULL || o < a.Lout;
            x[ii] = ok ? xe[(size_t)o * COUT] : 0.0f;
            g2[ii] = (ge && ok) ? ge[(size_t)o * COUT] : 0.0f;
          }
#pragma unroll
          for (int ii = 0; ii < 8; ++ii) {
            const int i = h8 * 8 + ii;
            const int o = (mb + (i & 3) + 8 * (i >> 2)) * U + ph;
            // mode 1: g' = (g + g2) * swish'(z), xhat from x (InstanceNorm + x*sigmoid(x) block)
            // mode 2: g' = g + g2, xhat from x, or from swish(z) when ep_xp_is_act (BatchNorm blocks)
            const float z = fmaf(x[ii], es1, et1);
            float gg2 = g2[ii];
            if (g2k) gg2 = fmaf(gk1, gg2, fmaf(gk2, sa_swish(z), gk3));
            float gg = acc[mt][i] + bv;
            if (a.relu) gg = fmaxf(gg, 0.0f);
            gg += gg2;
            if (a.ep_mode == 1) gg *= sa_swish_grad(z);
            const float xv = a.ep_xp_is_act ? sa_swish(z) : x[ii];
            const float xn = (xv - emu) * ers;
            if (FULL || o < a.Lout) {
              yb[(size_t)o * COUT] = gg;
              ssum += gg; ssq = fmaf(gg, xn, ssq);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          }
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int o = (mb + (i & 3) + 8 * (i >> 2)) * U + ph;
            float val = acc[mt][i] + bv;
            if (a.relu) val = fmaxf(val, 0.0f);
            if (FULL || o < a.Lout) {
              yb[(size_t)o * COUT] = val;
              ssum += val; ssq = fmaf(val, val, ssq);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);        // one m-tile at a time (bounds the live loads)
      }
    };
    const bool full = (m0 + C::BMB) * U <= a.Lout;      // uniform
    if (epm) {
      if (full) body(std::true_type{}, std::true_type{}); else body(std::false_type{}, std::true_type{});
    } else {
      if (full) body(std::true_type{}, std::false_type{}); else body(std::false_type{}, std::false_type{});
    }
    if (a.stats) {
      // a lane owns one column; the other half of the wave holds the rows +4: one fold, and the
      // waves that share this column block (SHARE of them) write their own slab
      ssum += __shfl_xor(ssum, 32, 64);
      ssq += __shfl_xor(ssq, 32, 64);
      if (lane < 32) {
        const int s = wm * (C::WN / C::NT) + wn / C::NT;
        float* dst = a.stats + ((((size_t)b * a.ntiles + tile) * C::SHARE + s) * COUT + col) * 2;
        dst[0] = ssum; dst[1] = ssq;
      }
    }
  };

  // ================= the two groups in anti-phase =================
  // group g: half-step hh = h - g even -> memory half (epilogue of its previous tile, staging of its
  // next), odd -> MFMA half.  Sub-group (g, sub) takes tile first + (2k + g) * NG + sub as its k-th.
  // group g runs its last epilogue at half-step 2*(its tile count) + g <= nset + 1
  const int nset = (last - first + NG - 1) / NG;
  for (int h = 0; h <= nset + 1; ++h) {
    const int hh = h - grp;
    cur_h = h;
    PP_STAMP(h, 0);
    if (hh >= 0) {
      if ((hh & 1) == 0) {
        const int kk = hh >> 1;
        const int tp = first + (2 * (kk - 1) + grp) * NG + sub, tn = first + (2 * kk + grp) * NG + sub;
        if (kk > 0 && tp < last) epilogue(tp);
        PP_STAMP(h, 1);
        if (tn < last) stage_tile(tn);
        PP_STAMP(h, 2);
      } else {
        const int tc = first + (2 * ((hh - 1) >> 1) + grp) * NG + sub;
        // zeroed on every path: the accumulators are then dead from the end of the epilogue to
        // here, and the staging pass has their registers
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[mt][i] = 0.0f;
        if (tc < last) mfma_tile(tc);
        PP_STAMP(h, 3);
      }
    }
    __syncthreads();
    PP_STAMP(h, 4);
  }
}

template <typename T, int CIN, int COUT, int SA, int U, int TM, bool PRO2 = false, int NG = 1>
static int launch_pp(const SaConvArgs& a, hipStream_t st) {
  typedef PPCfg<T, CIN, COUT, SA, U, TM> C;
  SaConvArgs args = a;
  args.ntiles = sa_div_up(sa_div_up(a.Lout, U), C::BMB);
  int omin = 1 << 30, omax = -(1 << 30), wmax = 0;
  for (int ph = 0; ph < U; ++ph)
    for (int t = 0; t < a.taps.ntaps[ph]; ++t) {
      omin = a.taps.off[ph][t] < omin ? a.taps.off[ph][t] : omin;
      omax = a.taps.off[ph][t] > omax ? a.taps.off[ph][t] : omax;
      wmax = a.taps.widx[ph][t] > wmax ? a.taps.widx[ph][t] : wmax;
    }
  if (omin > omax || omax - omin > SA_MAX_HALO) return -22;
  args.rowmin = omin;
  args.nrows = (C::BMB - 1) * SA + (omax - omin) + 1;
  args.wlo_off = (wmax + 1) * C::KSTEPS * C::NT * 64;      // Frag units: hi image size
  // a_out / colsum / pro_stats: every input row must be staged by the tile that owns it
  if ((a.a_out || a.nb_colsum || a.pro_stats) &&
      (omin > 0 || (C::BMB - 1) * SA + omax < C::BMB * SA - 1 ||
       (args.ntiles - 1) * C::BMB * SA + omin + args.nrows < a.Lin))
    return -22;
  if (a.a_out && sizeof(typename C::LT) != 2) return -22;
  if (PRO2 && (!a.nb_c1 || !a.nb_c2 || !a.nb_c3)) return -22;
  if (a.pro_stats && (PRO2 || !a.s1 || !a.swish || a.s2)) return -22;
  const size_t buf = C::buf_bytes(args.nrows);
  const size_t lds = 2 * NG * buf;
  if (lds > 160 * 1024) return -12;
  auto kern = sa_conv_pp_kernel<T, CIN, COUT, SA, U, TM, PRO2, NG>;
  static bool attr_set = false;
  static int n_cu = 0;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return -(int)e;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -19;
    n_cu = prop.multiProcessorCount;
    attr_set = true;
  }
  // grid: as many workgroups as the chip holds at once (1 or 2 per CU by LDS), each walking a
  // contiguous range of tiles; never fewer than 2 tiles per workgroup where the work allows it
  const int total = args.ntiles * a.B;
  const int per_cu = (int)((160 * 1024) / lds) >= 2 ? 2 : 1;
  int nwg = n_cu * per_cu;
  int per = sa_div_up(total, nwg);
  per = sa_div_up(per, 2 * NG) * 2 * NG;                 // whole rounds of the 2*NG sub-groups
  nwg = sa_div_up(total, per);
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(512 * NG), lds, st, args, per, total, (int)buf,
                     (int)(2 * NG * buf));
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// tile rows of the ping-pong kernel (0 = policy below; 64 / 128 = tuning knob)
static int g_pp_rows = 0;
extern "C" int sa_conv_pp_set_tile_rows(int rows) {
  if (rows != 0 && rows != 64 && rows != 128) return -22;
  g_pp_rows = rows;
  return 0;
}
int sa_pp_tile_rows(int cin, int cout, int u) {
  if (cin == 64 && cout == 32 && u == 2) return 128;      // every wave needs a full 32-row m-tile
  if (g_pp_rows) return g_pp_rows;
  return 128;
}

template <typename T, int CI, int CO, int S, int UU, bool PRO2 = false>
static int launch_pp_tm(const SaConvArgs& a, hipStream_t st) {
  if constexpr (CI == 64 && CO == 32 && UU == 2) {
    return launch_pp<T, CI, CO, S, UU, 128, PRO2, 1>(a, st);
  } else {
    // 64-row tiles: two 512-thread workgroups per CU (128-register budget), i.e. two waves per
    // SIMD in the MFMA half and two in the memory half; 128-row tiles: one workgroup per CU
    return sa_pp_tile_rows(CI, CO, UU) == 64 ? launch_pp<T, CI, CO, S, UU, 64, PRO2, 1>(a, st)
                                             : launch_pp<T, CI, CO, S, UU, 128, PRO2, 1>(a, st);
  }
}

#define SA_PP_CASE(CI, CO, S, UU)                                                  \
  if (cin == CI && cout == CO && sa == S && u == UU) {                             \
    if (a->nb_x)                                                                   \
      return dtype == SA_BF16X3 ? launch_pp_tm<bf16x3_t, CI, CO, S, UU, true>(*a, st) : -22; \
    return dtype == SA_BF16X3 ? launch_pp_tm<bf16x3_t, CI, CO, S, UU>(*a, st)      \
           : dtype == SA_F32 ? launch_pp_tm<float, CI, CO, S, UU>(*a, st) : -22;   \
  }

// called by sa_conv_gemm (sa_conv_gemm.hip) for the fp32-storage policies
int sa_conv_pp_dispatch(int dtype, int cin, int cout, int sa, int u, const SaConvArgs* a, hipStream_t st) {
  SA_PP_CASE(32, 64, 2, 1)
  SA_PP_CASE(64, 64, 1, 1)
  SA_PP_CASE(64, 128, 2, 1)
  SA_PP_CASE(128, 128, 1, 1)
  SA_PP_CASE(128, 64, 1, 2)
  SA_PP_CASE(64, 32, 1, 2)
  return -38;
}

// statistics slabs per tile of the ping-pong kernel
int sa_pp_share(int cout) { return 4 / (cout / 32); }
