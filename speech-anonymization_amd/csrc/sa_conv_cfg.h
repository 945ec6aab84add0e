// Tile configuration shared by the two implicit-GEMM convolution kernels (sa_conv_gemm.hip: one
// tile per 4-wave workgroup; sa_conv_pp.hip: two 4-wave groups per workgroup in anti-phase).
#pragma once
#include "sa_common.h"

#define SA_MAX_HALO 16      // max (largest - smallest) tap row offset the prologue is sized for

template <typename T, int CIN, int COUT, int SA, int U, int TM>
struct ConvCfg {
  typedef Pol<T> P;
  typedef typename P::store_t S;
  typedef typename P::lds_t LT;
  static constexpr int VEC = P::VEC;                // storage elements per 16-byte chunk
  static constexpr int KS = P::KS;
  static constexpr int KSTEPS = CIN / KS;
  static constexpr int NT = COUT / 32;
  static constexpr int VT = NT * U;                 // virtual n-tiles (phase, n-tile)
  static constexpr int BMB = TM / U;                // base rows per workgroup (TM output rows)
  static constexpr int WN = VT >= 4 ? 4 : VT;       // waves along N
  static constexpr int WM = 4 / WN;                 // waves along M
  static constexpr int VPW = VT / WN;               // virtual n-tiles per wave
  static constexpr int MT = BMB / (32 * WM);        // 32-row m-tiles per wave
  static constexpr int APITCH = CIN + P::PAD;       // LDS operand pitch (lds_t elements)
  static constexpr int OPITCH = COUT + (sizeof(S) == 2 ? 8 : 4);
  static constexpr int CHI = CIN / VEC;             // 16-byte chunks per input row
  static constexpr int RPPI = 256 / CHI;
  static constexpr int OVEC = 16 / sizeof(S);
  static constexpr int CHO = COUT / OVEC;
  static constexpr int RPPO = 256 / CHO;
  static_assert(MT >= 1 && VT % WN == 0, "tile shape");
  static size_t tile_bytes(int nrows) {
    size_t a = (size_t)P::NPL * nrows * APITCH * sizeof(LT);
    size_t o = (size_t)TM * OPITCH * sizeof(S);
    size_t m = a > o ? a : o;
    return (m + 15) & ~(size_t)15;
  }
  static size_t lds_bytes(int nrows) {                 // the statistics scratch overlays the tile
    size_t t = tile_bytes(nrows), r = (size_t)RPPO * COUT * 2 * sizeof(float);
    return t > r ? t : r;
  }
};

