// Shared pieces of the weight-stationary convolution kernels (sa_conv_ws.hip: forward launches;
// sa_conv_wsd.hip: fused data gradients): tile geometry, LDS-DMA / store / convert helpers.
#pragma once
#include <type_traits>
#include "sa_conv_cfg.h"

namespace {

// geometry of one instantiation: C_ channels in and out (128 or 64), NT taps spanning HALO rows (5 taps
// at unit spacing: 4; the dilated 3-tap TDNN layers: 4 and 6).  A wave always owns 64 rows x 32 columns
// (two 32x32 accumulators, six MFMAs per k-step): with 128 channels the four waves are four column
// blocks of a 64-row tile, with 64 channels two column blocks x two row halves of a 128-row tile.
// CO_ / SA_: output channels and input stride of the stride-2 encoder layer (64 -> 128: two input rows
// per output row; everything on the input side -- staging, transform, operand planes -- keeps C_).
// U_ = 2: a transposed layer (128 -> 64, stride 2) as two output phases of the same 64 base rows: the
// two waves that share a column block are the two PHASES (output rows 2m and 2m+1, each with its own
// taps; NT is the larger tap count, the other phase runs zero fragments for the missing tap).
template <int C_, int NT, int HALO_, int CO_ = C_, int SA_ = 1, int U_ = 1>
struct WsGeo {
  static constexpr int C = C_, CO = CO_, SA = SA_, U = U_, NTAPS = NT, HALO = HALO_;
  static constexpr int NWN = CO_ / 32, NWM = 4 / NWN;      // waves along the columns / the rows (or phases)
  static_assert(U_ == 1 || NWM == 2 || NWM == 4, "two phases = wave pairs of a column block");
  static constexpr int RHW = NWM / U_;                     // row halves (64 base rows each) of a tile
  static constexpr int BM = 64 * RHW;                      // base (input-grid) rows per tile
  // statistics slabs per tile (the one-tile kernel's tiles are 128 output rows for these shapes: a
  // transposed tile with two row halves spans two of them) and waves whose sums share a slab entry
  static constexpr int SLABS = U_ == 2 ? RHW : 1, COMBINE = NWM / SLABS;
  static constexpr int TM = BM * U_;                       // output rows per tile
  static constexpr int KSTEPS = C_ / 16;
  static constexpr int PITCH = C_ + 8;                     // bf16 elements per LDS operand row (conflict-free ds_read_b128)
  static constexpr int RPP = 256 / C_;                     // rows per 1-KiB DMA piece (2 or 4)
  static constexpr int LPR = 64 / RPP;                     // lanes per row of a piece (4 channels each)
  static constexpr int ROWS = ((BM - 1) * SA_ + 1 + HALO_ + RPP - 1) / RPP * RPP;   // staged input rows per tile, whole DMA pieces
  static constexpr int PLANE = ROWS * PITCH;               // bf16 elements per plane
  static constexpr int NDMA = ROWS / RPP;                  // 1-KiB DMA pieces per tile and tensor
  static constexpr int DPW = (NDMA + 3) / 4;               // pieces per wave (the last waves have one less)
  static constexpr int RAW_BYTES = ROWS * C_ * 4;          // one raw fp32 tile
  static constexpr int BUF_BYTES = 2 * PLANE * 2;          // one operand buffer (hi + lo planes)
  // (tap, k-step) pairs whose hi + lo weight fragments live in AGPRs (at most all 256 of them)
  static constexpr int NAGPR_FRAGS = NT * KSTEPS < 32 ? NT * KSTEPS : 32;
  static constexpr int NSLOT = NT * KSTEPS * 6;            // MFMAs = filler slots per tile
  // slots per piece of the transform: 18 where the tile has 240 slots (up to seven arithmetic
  // levels), 12 (one level) for the 3-tap layers' 144.  Multiples of a step's six slots: the LDS
  // instructions of a piece (raw read, operand-plane writes) then always sit in slots 0 / 1 of a step,
  // four MFMAs ahead of the lgkmcnt(0) that hipcc puts in front of the next step's first MFMA.  The
  // epilogue takes slots 0..33 (one accumulator register per slot, then the statistics).
  static constexpr int SUBS = NSLOT == 72 ? 4 : NSLOT == 120 ? 6 : NT == 5 ? 18 : 12;   // (64 input channels: 120 / 72 slots per tile, levels packed)
  static constexpr int FT = NSLOT - SUBS * DPW;            // first transform slot
  static_assert(FT > 34 && FT % 6 == 0, "epilogue slots / step alignment");
};

typedef __attribute__((address_space(3))) unsigned char lds_byte;

// one 1-KiB piece: lane l's 16 bytes land at lds_dst + 16*l (cdna_hip_programming.md 5.7: M0 is
// written in the statement that reads it; the s_nop is the M0 -> LDS-DMA wait state)
__device__ static inline void ws_dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// the same with a uniform base and a 32-bit lane offset (no per-piece vector address arithmetic)
__device__ static inline void ws_dma16s(const void* gbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(gbase), "s"(lds_dst) : "memory");
}

// stores of the filler slots: uniform base (SGPR pair) + 32-bit lane offset, exactly one instruction
// (hipcc rebuilds a 64-bit vector address per store otherwise); not counted by hipcc's vmcnt
// bookkeeping -- the kernel waits with its own s_waitcnt vmcnt(0)
__device__ static inline void ws_store_b32(void* base, unsigned voff, float v) {
  asm volatile("global_store_dword %0, %1, %2" :: "v"(voff), "v"(v), "s"(base) : "memory");
}
__device__ static inline void ws_store_b64(void* base, unsigned voff, uint2 v) {
  asm volatile("global_store_dwordx2 %0, %1, %2" :: "v"(voff), "v"(v), "s"(base) : "memory");
}

// (bf16(a), bf16(b)) in one instruction, RNE like the cast (hipcc converts one value per
// instruction and assembles the pair with shifts and ors)
__device__ static inline unsigned ws_cvt_pk_bf16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

template <int I, int N, class F>
__device__ __forceinline__ void ws_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    ws_static_for<I + 1, N>(f);
  }
}

}  // namespace
