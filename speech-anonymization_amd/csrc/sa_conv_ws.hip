// Implicit-GEMM 1-D convolution on MFMA (gfx950), channels-last -- the WEIGHT-STATIONARY form of
// the stride-1 FORWARD launches with equal channel counts in the bf16x3 policy (fp32 storage, split-
// bf16 operands): 128 -> 128, models/ConvAutoEncoder.py:150-158 (encoder.11), :161-166 (decoder.0) and
// :33-43 (the three TDNN layers of the sex classifier: 5 taps, 3 taps dilation 2, 3 taps dilation 3);
// 64 -> 64, :146-148 (encoder.5) and :167-169 (decoder.4); the stride-2 layer 64 -> 128, :149 (encoder.8);
// the transposed layers 128 -> 64, :161-163 (decoder.1), and 64 -> 32, :170-172 (decoder.5), as two output phases.
//
// Same operation, arguments, statistics-slab geometry and output BITS as the 64-row one-tile kernel
// (sa_conv_gemm.hip); different execution structure, chosen from the measurements in
// profiles/r02_conv_structure_experiments.md (the one-tile kernel's phases do not overlap: row loads
// 76 us + weight-fragment stream 61 us + MFMA 122 us + epilogue 37 us + launch floor 45 us):
//
//   * ONE persistent 4-wave workgroup per CU (one wave per SIMD, the whole 512-register file per
//     lane); a workgroup walks a contiguous range of 64-row tiles (64 channels: 128-row tiles, the
//     waves being two column blocks x two row halves).
//   * The weights never move: wave w owns output columns [32w, 32w+32) and keeps its B fragments
//     for all taps x 128 input channels x (hi, lo) -- 80 fragments = 320 registers with 5 taps -- for
//     the whole launch (the one-tile kernel re-streams 327 KB of fragments from L2 per tile); 256 of
//     them are AGPRs, read directly by MFMAs issued from inline asm.
//   * Input rows arrive by LDS-DMA (global_load_lds_dwordx4, no register destination), two tiles
//     ahead of the MFMA loop that consumes them.
//   * A wave issues one instruction per 4 cycles and a 32x32x16 MFMA holds the matrix pipe for 32:
//     a tile is 240 (144) single-MFMA statements with a FILLER SLOT of up to ~7 instructions behind
//     each.  The slots of tile t's loop hold the epilogue of tile t-1 (from an asm copy of its
//     accumulators; a register of a 32x32 accumulator = two 128-byte row segments per store), the
//     transform of tile t+1 (raw LDS -> registers -> hi / lo operand planes, cut by arithmetic stage
//     so that a slot holds independent instructions) and the refill DMA of tile t+2.  One barrier
//     and one s_waitcnt vmcnt(0) per tile, the latter where everything outstanding is a microsecond
//     old.
//   * Tiles at the ends of an utterance (clamped DMA addresses, zeroed rows, ownership tests),
//     partial output tiles and a workgroup's first tile take a plain path without overlap.
//
// tools/ws_audit.py (ISA audit, also a CPU test), tools/ws_slots.py (instructions per slot),
// tools/ws_stamps.py (-DSA_WS_STAMPS: cycles per phase), tools/ws_check.py (bits against the
// one-tile kernel).
//
// Launches it does not cover stay on sa_conv_gemm.hip: the data gradients.  A fused data gradient
// reads two more fp32 tiles per output tile (the stored forward tensor for the normalisation-backward
// prologue, and it again + a second gradient in the backward epilogue): 70 KB more LDS or 64 more
// registers per lane than this structure has left (512 registers: 320 weights, 64 accumulators +
// their epilogue copy, 24 A fragments, the rest transform state), and with one wave per SIMD its
// heavier transform and epilogue would be instruction-issue-bound.  The prologue alone (MODE 2) is
// kept, bit-equal, behind -DSA_WS_PRO2.
#include <type_traits>
#include "sa_conv_cfg.h"

#ifndef SA_ABL
#define SA_ABL 0
#endif

// -DSA_WS_STAMPS: diagnostic build (tools/ws_stamps.py): s_memtime at the phase boundaries of one
// workgroup's wave 0; no stamp exists in the normal build.
#ifdef SA_WS_STAMPS
__device__ unsigned long long sa_ws_dbg[64 * 8];
#define WS_STAMP(it, i) do { if (lane_ == 0 && wave_ == 0 && blockIdx.x == 7 && (it) < 64) { \
  unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
  sa_ws_dbg[(it) * 8 + (i)] = t_; } } while (0)
extern "C" int sa_ws_dbg_read(unsigned long long* out) {
  return -(int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sa_ws_dbg), sizeof(sa_ws_dbg));
}
#else
#define WS_STAMP(it, i)
#endif

#include "sa_conv_ws_common.h"

extern unsigned long long g_ws_xcd_weights;                  // (defined below, next to its setter)

namespace {

// Filler slots.  With one wave per SIMD a wave issues one instruction (of any kind) per 4 cycles, a
// 32x32x16 MFMA holds the matrix pipe for 32: the MFMA loop of a tile is 240 (3 taps: 144) single-MFMA
// asm statements, and the gap behind statement f ("slot f") takes about seven instructions of OTHER
// tiles' work without delaying statement f+1.  So the fillers are written to the instruction: lane
// offsets, LDS addresses and store offsets are set up once per kernel / tile, every slot is base +
// immediate, and consecutive dependence levels of a computation sit in consecutive slots.
//   slots 0..31   epilogue of the previous tile, one accumulator register (two 128-byte row
//                 segments) per slot, the value formed one slot ahead; slots 32, 33: its statistics
//   slot  FT-1    counted s_waitcnt: everything up to the refill DMA of the previous loop has landed
//                 (in order), the 32 (33) stores above may stay in flight; first raw read
//   slots FT..    transform of the next tile, 18 (12) slots = 3 (2) MFMA steps per piece: arithmetic
//                 levels, split levels, operand cache, pro_stats, refill DMA of the piece for the tile
//                 after next.  LDS instructions (raw read of the next piece, operand-plane writes of the
//                 previous one) sit in slots 0 / 1 of a step only: hipcc waits lgkmcnt(0) in front of
//                 every step's first MFMA.
// Tiles at the ends of an utterance (rows outside it: clamped DMA addresses, zeroed operand rows,
// ownership checks), partial output tiles and a workgroup's first tile take a plain path.
// MODE: 0 no transform, 1 affine (per utterance, channel) + x*sigmoid(x), 2 normalisation-backward
// prologue (nb_*: d y = c1*dz + c2*y + c3 [* (y > 0)] over two input tensors)
// 3: mode 1 + pro_stats (per-tile sum / sum of squares of the transformed rows the tile owns),
// 4: mode 1 + a second, per-channel affine (the classifier's input BatchNorm behind the activation)
// 5: one per-channel affine only (the dilated TDNN layers: BatchNorm of the layer below in front)
template <int MODE, int NT, int HALO, int CC = 128, int CO = CC, int SA = 1, int UU = 1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void sa_conv_ws_kernel(SaConvArgs a, int bcost, int total_tiles, unsigned long long xw) {
  typedef WsGeo<CC, NT, HALO, CO, SA, UU> G;
  constexpr int WS_CO = G::CO, WS_SA = G::SA, WS_BM = G::BM;
  static_assert(UU == 1 || MODE == 0, "transposed layers: plain rows");
  constexpr int WS_C = G::C, WS_TM = G::TM, WS_KSTEPS = G::KSTEPS, WS_PITCH = G::PITCH, RPP = G::RPP, LPR = G::LPR,
                NWN = G::NWN, NWM = G::NWM;
  static_assert(CC == 128 || MODE == 0 || MODE == 1, "64 channels: plain and affine + x*sigmoid(x) prologues");
  constexpr int WS_NTAPS = G::NTAPS, WS_ROWS = G::ROWS, WS_PLANE = G::PLANE, WS_NDMA = G::NDMA, WS_DPW = G::DPW,
                WS_RAW_BYTES = G::RAW_BYTES, WS_BUF_BYTES = G::BUF_BYTES, WS_NAGPR_FRAGS = G::NAGPR_FRAGS,
                WS_FT = G::FT, WS_SUBS = G::SUBS;
  (void)WS_ROWS; (void)NWM;
  static_assert(MODE == 0 || MODE == 5 || NT == 5, "the staged transforms need the 240-slot tile");
  constexpr bool PRO2 = MODE == 2;
  constexpr bool SWISH = MODE == 1 || MODE == 3 || MODE == 4;   // affine (per utterance, channel) + x*sigmoid(x)
  constexpr bool PSTAT = MODE == 3, AFF2 = MODE == 4 || MODE == 5;   // (mode 5: the second affine alone)
  constexpr bool COLRED = PRO2 || PSTAT;                    // per-tile column reductions through the LDS scratch
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // LDS: [operand buffer 0][operand buffer 1][raw tile x][raw tile nb_x (PRO2)][column-reduction scratch (PRO2, pro_stats)]
  bf16_t* const planes = reinterpret_cast<bf16_t*>(smem);
  unsigned char* const raw = smem + 2 * WS_BUF_BYTES;
  const unsigned raw_lds = (unsigned)(uintptr_t)(lds_byte*)raw;
  const int tid = threadIdx.x, lane_ = tid & 63;
  const int wave_ = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn_ = wave_ % NWN, wm_ = wave_ / NWN;           // this wave's column block / row half (or phase) of the tile
  const int ph_ = UU == 2 ? wm_ % 2 : 0, rh_ = UU == 2 ? wm_ / 2 : wm_;   // output phase, row half
  const int ntap_ = a.taps.ntaps[ph_];                      // (transposed layers: 3 and 2)
  // This workgroup's contiguous tile range, cut at equal COST in quarter tiles: an iteration whose NEXT or
  // next-but-one tile is an edge tile (the last tiles of an utterance; the wrap into the next one) runs
  // un-overlapped and costs `pq` (about 2.25 tiles, measured with tools/wsd_stamps.py on the data-gradient
  // kernel, which shares this structure), an overlapped one 4; every tile carries its own cost, and a launch
  // ends with its slowest workgroup.
#ifdef SA_WS_OLD_RANGES
  int first, last;
  {                                                         // (A/B build only: round 2's ranges, one surcharge behind each utterance)
    const unsigned long long U = (unsigned long long)a.ntiles + 4u;
    const unsigned long long ctot = (unsigned long long)(total_tiles / a.ntiles) * U;
    auto inv = [&](unsigned long long c) {
      const unsigned long long k = c / U, r = c - k * U;
      const unsigned long long t = k * (unsigned)a.ntiles + (r < (unsigned)a.ntiles ? r : (unsigned)a.ntiles);
      return (int)(t < (unsigned)total_tiles ? t : (unsigned)total_tiles);
    };
    first = inv(ctot * blockIdx.x / gridDim.x);
    last = blockIdx.x + 1 == gridDim.x ? total_tiles : inv(ctot * (blockIdx.x + 1) / gridDim.x);
    (void)xw;
  }
#else
  int first, last;
  {
    const int e_num = a.Lin - WS_ROWS - a.rowmin, e_den = WS_BM * WS_SA;
    int e_hi = e_num >= 0 ? e_num / e_den : -((-e_num + e_den - 1) / e_den);     // (floor)
    if (e_hi > a.ntiles - 2) e_hi = a.ntiles - 2;
    int nt = a.ntiles + 1 - e_hi;                           // tiles t with t + 2 > e_hi
    if (nt > a.ntiles) nt = a.ntiles;
    if (nt < 0) nt = 0;
    const unsigned long long pq = (unsigned)(bcost & 0xffff), ni = (unsigned)(a.ntiles - nt);
    const unsigned long long U = pq * (unsigned)nt + 4 * ni;
    const unsigned long long ctot = (unsigned long long)(total_tiles / a.ntiles) * U;
    auto inv = [&](unsigned long long c) {                  // tiles wholly in front of cost position c
      const unsigned long long k = c / U, r = c - k * U;
      unsigned long long t = r < 4 * ni ? r / 4 : ni + (r - 4 * ni) / pq;
      t += k * (unsigned)a.ntiles;
      return (int)(t < (unsigned)total_tiles ? t : (unsigned)total_tiles);
    };
    // (workgroup i runs on XCD i % 8; its share of the cost follows the weight of its XCD: see sa_conv_wsd.hip)
    unsigned S8 = 0;
#pragma unroll
    for (int x = 0; x < 8; ++x) S8 += (unsigned)(xw >> (8 * x)) & 255u;
    auto prefix = [&](unsigned i) {
      unsigned pfx = (i >> 3) * S8;
#pragma unroll
      for (int x = 0; x < 8; ++x) pfx += x < (int)(i & 7) ? (unsigned)(xw >> (8 * x)) & 255u : 0u;
      return (unsigned long long)pfx;
    };
    const unsigned long long wtot = prefix(gridDim.x);
    first = inv(ctot * prefix(blockIdx.x) / wtot);
    last = blockIdx.x + 1 == gridDim.x ? total_tiles : inv(ctot * prefix(blockIdx.x + 1) / wtot);
  }
#endif
  if (first >= last) return;

  // ---- the weights: this wave's 32 output columns, all taps / channels, hi and lo images ----
  bf16x8 Bh[WS_NTAPS][WS_KSTEPS], Bl[WS_NTAPS][WS_KSTEPS];
  {
    const bf16x8* wp = reinterpret_cast<const bf16x8*>(a.wp);
#pragma unroll
    for (int t = 0; t < WS_NTAPS; ++t) {
      const bool has_t = UU == 1 || t < ntap_;
      const bf16x8* wt = wp + ((size_t)a.taps.widx[ph_][has_t ? t : 0] * WS_KSTEPS * NWN + wn_) * 64 + lane_;
#pragma unroll
      for (int k = 0; k < WS_KSTEPS; ++k) {
        Bh[t][k] = wt[(size_t)k * NWN * 64];
        Bl[t][k] = wt[(size_t)a.wlo_off + (size_t)k * NWN * 64];
        if (UU == 2 && !has_t) {                           // the phase with fewer taps: zero fragments
          Bh[t][k] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
          Bl[t][k] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
      }
      // one tap at a time, moved to its AGPR home before the next tap is fetched (all 80 loads at
      // once would need 320 VGPRs).  The empty asm also makes hipcc wait for the loads HERE: a value
      // still "pending" at the loop header gets its s_waitcnt vmcnt(0) inside the loop, in front of
      // every use, and that wait would drain the LDS-DMA in flight there.
#pragma unroll
      for (int k = 0; k < WS_KSTEPS; ++k) {
        if (t * WS_KSTEPS + k < WS_NAGPR_FRAGS) asm volatile("" : "+a"(Bh[t][k]), "+a"(Bl[t][k]));
        else asm volatile("" : "+v"(Bh[t][k]), "+v"(Bl[t][k]));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float bv = a.bias ? a.bias[wn_ * 32 + (lane_ & 31)] : 0.0f;
  float relu_floor = a.relu ? 0.0f : -__builtin_inff();     // max(v, -inf) = v: no branch in the slot
  const bool has_stats = a.stats != nullptr, has_ao = a.a_out != nullptr;

  // ---- lane constants of the filler slots (made opaque so that hipcc keeps THESE and derives the
  // per-slot addresses from them by immediates, instead of hoisting one address chain per slot) ----
  const int half_ = lane_ >> 5, l31_ = lane_ & 31;             // accumulator layout: rows +4, column
  const int rowp_ = lane_ / LPR, cq_ = lane_ % LPR;            // transform layout: row of the piece, channel quad
  // this wave's piece j is piece i = wave + 4j of the tile: rows RPP*i + rowp, i.e. row0 + 4*RPP*j
  unsigned row0 = RPP * wave_ + rowp_;
  unsigned dma_off = wave_ * 1024 + lane_ * 16;            // byte offset in the row block of a tile (rows are contiguous): + j*4096
  unsigned raw_off = wave_ * 1024 + lane_ * 16;            // byte offset in the raw tile: + j*4096
  unsigned pl_off = (row0 * WS_PITCH + cq_ * 4) * 2;       // byte offset in an operand plane: + j*4*RPP*PITCH*2
  unsigned ao_off = (row0 * WS_C + cq_ * 4) * 2;           // byte offset in the a_out row block: + j*4*RPP*C*2
  unsigned y_off = (UU == 2 ? (8 * half_ + ph_ + 128 * rh_) * WS_CO + wn_ * 32 + l31_   // phase rows 2(64 rh + ro + 4 half) + ph: + 2*ro*CO*4
                            : (4 * half_ + 64 * rh_) * WS_CO + wn_ * 32 + l31_) * 4;  // byte offset in the y row block: + ro*CO*4
  unsigned st_off = ((G::SLABS == 2 ? rh_ * WS_CO : 0) + wn_ * 32 + l31_) * 8;   // byte offset of this lane's column in the tile's statistics slab(s)
  unsigned swap_off = (lane_ ^ 32) * 4;                    // ds_bpermute address of the lane in the other half
  asm volatile("" : "+v"(bv), "+v"(relu_floor), "+v"(row0), "+v"(dma_off), "+v"(raw_off), "+v"(pl_off), "+v"(ao_off), "+v"(y_off),
               "+v"(st_off), "+v"(swap_off));

#define WS_IDS int lane = lane_, wave = wave_; asm volatile("" : "+v"(lane), "+s"(wave)); (void)wave; (void)lane

  // tile index -> (utterance, tile of the utterance) and what the slots need of it
  // a tile = (utterance, tile of the utterance) + its first staged input row, first output row and first
  // statistics slab in the [B*Lin] / [B*Lout] / [B*slabs] index spaces, advanced incrementally (the
  // per-tile bookkeeping is scalar code between two tile bodies: nothing overlaps it)
  struct Tile { int b, tile, irow, orow, srow; };
  // (two slabs per tile: the caller's slab count per utterance is that of 64-base-row tiles, possibly odd)
  const int slab_short = G::SLABS == 2 ? 2 * a.ntiles - (((a.Lout + UU - 1) / UU + 63) / 64) : 0;
  const int nslab_b = a.ntiles * G::SLABS - slab_short;
  auto tile_of = [&](int t) {
    Tile r; r.b = t / a.ntiles; r.tile = t - r.b * a.ntiles;
    r.irow = r.b * a.Lin + r.tile * (WS_BM * WS_SA) + a.rowmin; r.orow = r.b * a.Lout + r.tile * WS_TM;
    r.srow = r.b * nslab_b + r.tile * G::SLABS;
    return r;
  };
  const int iwrap = a.Lin - (a.ntiles - 1) * (WS_BM * WS_SA), owrap = a.Lout - (a.ntiles - 1) * WS_TM,
            swrap = nslab_b - (a.ntiles - 1) * G::SLABS;
  auto next_tile = [&](Tile T) {
    Tile r; const bool wrap = T.tile + 1 == a.ntiles;
    r.b = wrap ? T.b + 1 : T.b; r.tile = wrap ? 0 : T.tile + 1;
    r.irow = T.irow + (wrap ? iwrap : WS_BM * WS_SA); r.orow = T.orow + (wrap ? owrap : WS_TM);
    r.srow = T.srow + (wrap ? swrap : G::SLABS);
    return r;
  };
  // rows outside the utterance among the 68 staged ones, or the trailing rows it owns beyond its 64
  // tile*BM*SA + rowmin < 0, tile*BM*SA + rowmin + ROWS > Lin, or the last tile: two thresholds on the tile index
  auto floor_div = [](int n, int d) { return n >= 0 ? n / d : -((-n + d - 1) / d); };
  const int edge_lo = a.rowmin < 0 ? (-a.rowmin + WS_BM * WS_SA - 1) / (WS_BM * WS_SA) : 0;
  int edge_hi = floor_div(a.Lin - WS_ROWS - a.rowmin, WS_BM * WS_SA);
  if (edge_hi > a.ntiles - 2) edge_hi = a.ntiles - 2;
  const int part_hi = floor_div(a.Lout - WS_TM, WS_TM);     // tiles beyond it are partial output tiles
  auto is_edge = [&](Tile T) { return T.tile < edge_lo || T.tile > edge_hi; };

  // per-utterance transform constants (reloaded when the tile range crosses an utterance)
  float s1[4], t1[4], k1[4], k2[4], k3[4], s2[4], t2[4];
  int cur_b = -1;
  if constexpr (AFF2) {                                     // per channel: loaded once
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s2[j] = a.s2[cq_ * 4 + j];
      t2[j] = a.t2 ? a.t2[cq_ * 4 + j] : 0.0f;
      asm volatile("" : "+v"(s2[j]), "+v"(t2[j]));
    }
  }
  auto load_consts = [&](int b) {
    WS_IDS;
    const int ch = (lane % LPR) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (SWISH) {
        s1[j] = a.s1[(size_t)b * WS_C + ch + j];
        t1[j] = a.t1[(size_t)b * WS_C + ch + j];
      }
      if constexpr (PRO2) {
        const size_t q = (size_t)b * a.nb_bstride + ch + j;
        k1[j] = a.nb_c1[q]; k2[j] = a.nb_c2[q]; k3[j] = a.nb_c3[q];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {                          // waited for here, not in the filler slots
      if constexpr (SWISH) asm volatile("" : "+v"(s1[j]), "+v"(t1[j]));
      if constexpr (PRO2) asm volatile("" : "+v"(k1[j]), "+v"(k2[j]), "+v"(k3[j]));
    }
    cur_b = b;
  };

  // ---- LDS-DMA of piece j of this wave (rows 2i, 2i+1; i = wave + 4j) of tile T ----
  // uniform per-tile bases of the slots (set once per iteration, opaque: hipcc otherwise recomputes
  // them -- two scalar multiplies and a 64-bit add chain -- in every slot)
  const char* xbase_d = nullptr; const char* x2base_d = nullptr;    // rows of the DMA tile (x, nb_x)
  char* aobase_t = nullptr;                                          // a_out rows of the transform tile
  char* ybase_e = nullptr;                                           // y rows of the epilogue tile
  auto row_ptr = [&](const void* p, int row, int row_bytes) {
    const char* r = reinterpret_cast<const char*>(p) + (long)row * row_bytes;   // B*L rows < 2^31 (checked at launch)
    asm volatile("" : "+s"(r));
    return r;
  };
  auto dma_piece = [&](Tile T, bool edge, int j, int part = 2) {
    const int i = wave_ + 4 * j;
    if (j == WS_DPW - 1 && i >= WS_NDMA) return;
    const int g0 = T.tile * (WS_BM * WS_SA) + a.rowmin;
    if (!edge) {
      if (part != 1) ws_dma16s(xbase_d + j * 4096, dma_off, raw_lds + i * 1024);
      if constexpr (PRO2) {
        if (part != 0) ws_dma16s(x2base_d + j * 4096, dma_off, raw_lds + WS_RAW_BYTES + i * 1024);
      }
    } else {
      WS_IDS;
      int g = g0 + RPP * i + lane / LPR;
      g = g < 0 ? 0 : (g >= a.Lin ? a.Lin - 1 : g);        // rows outside the utterance: any valid address (zeroed in the transform)
      const size_t off = ((size_t)T.b * a.Lin + g) * (WS_C * 4) + (lane % LPR) * 16;
      ws_dma16(reinterpret_cast<const char*>(a.x) + off, raw_lds + i * 1024);
      if constexpr (PRO2) ws_dma16(reinterpret_cast<const char*>(a.nb_x) + off, raw_lds + WS_RAW_BYTES + i * 1024);
    }
  };

  // ---- transform of a piece, in sub-steps (each small enough for one filler slot) ----
  f32x4 vx[1], vy[1];                                      // the raw piece being transformed (the next one is read as soon as stage 0 has consumed it)
  float f[4], csum[4], csq[4];
  uint2 phi, plo;                                          // hi / lo bf16 quadruples of the piece
  unsigned pl_cur = 0;                                     // LDS byte offset of this lane's first row in the transform tile's buffer
  auto piece_read = [&](int j, int part = 2) {              // 0: x, 1: nb_x (PRO2), 2: both
    if (j == WS_DPW - 1 && wave_ + 4 * j >= WS_NDMA) return;
    if (part != 1) vx[0] = *reinterpret_cast<const f32x4*>(raw + (raw_off + j * 4096));
    if constexpr (PRO2) {
      if (part != 0) vy[0] = *reinterpret_cast<const f32x4*>(raw + (raw_off + WS_RAW_BYTES + j * 4096));
    }
  };
  auto own_range = [&](Tile T, int& lo, int& hi) {
    lo = T.tile * (WS_BM * WS_SA);
    hi = T.tile == a.ntiles - 1 ? a.Lin : lo + WS_BM * WS_SA;
    if (hi > a.Lin) hi = a.Lin;
  };
  // is this lane's row of piece j one the tile owns (operand cache, column sums)?  interior tiles:
  // every piece but the two halo pieces (i = 0: rows -2, -1; i = 33: rows 64, 65), a wave-uniform test
  auto owns = [&](Tile T, bool edge, int j) -> bool {
    if (!edge) {                                           // own rows = staged rows [-rowmin, -rowmin + TM)
      if (j > 0 && j < WS_DPW - 2) return true;            // the middle pieces are owned in every geometry
      const int r = (int)row0 + 4 * RPP * j;               // (per lane where a piece straddles the boundary)
      return r >= -a.rowmin && r < -a.rowmin + WS_BM * WS_SA;
    }
    int lo, hi;
    own_range(T, lo, hi);
    const int g = T.tile * (WS_BM * WS_SA) + a.rowmin + (int)row0 + 4 * RPP * j;
    return g >= lo && g < hi;
  };
  // hi = bf16(v), lo = bf16(v - hi) of a channel pair (same roundings as sa_split4)
  auto split_pair = [&](float a0, float a1, unsigned& hi, unsigned& lo) {
    // (opaque: with the product z * r that formed a0 in the same block, hipcc contracts a0 - hi into
    // fma(z, r, -hi), i.e. splits the UNROUNDED product -- one bf16 ulp of lo off the one-tile kernel
    // in 1.4 % of the elements, and dependent on which path staged the tile)
    asm volatile("" : "+v"(a0), "+v"(a1));
    hi = ws_cvt_pk_bf16(a0, a1);
    lo = ws_cvt_pk_bf16(a0 - __uint_as_float(hi << 16), a1 - __uint_as_float(hi & 0xffff0000u));
  };
  auto piece_elem = [&](Tile T, bool edge, int j, int q) {            // channel q of the lane's four
    if (j == WS_DPW - 1 && wave_ + 4 * j >= WS_NDMA) return;
    float v = vx[0][q];
    if constexpr (SWISH) {                                 // the very operations of piece_level: a row's operand
      const float zz = fmaf(v, s1[q], t1[q]);              // must not depend on which path staged it
      v = zz * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(zz * -1.4426950408889634f));
      if constexpr (AFF2) v = fmaf(v, s2[q], t2[q]);
      if constexpr (PSTAT) {
        if (owns(T, edge, j)) { csum[q] += v; csq[q] = fmaf(v, v, csq[q]); }
      }
    }
    if constexpr (MODE == 5) v = fmaf(v, s2[q], t2[q]);
    if constexpr (PRO2) {
      const float y = vy[0][q];
      v = fmaf(k1[q], v, fmaf(k2[q], y, k3[q]));
      if (a.nb_relu_mask) v = y > 0.0f ? v : 0.0f;
      if (a.nb_colsum && owns(T, edge, j)) csum[q] += v;
    }
    if (edge) {                                            // rows outside the utterance are zero operands
      const int g = T.tile * (WS_BM * WS_SA) + a.rowmin + (int)row0 + 4 * RPP * j;
      if (!(g >= 0 && g < a.Lin)) v = 0.0f;
    }
    f[q] = v;
    if (q == 1) split_pair(f[0], f[1], phi.x, plo.x);
    if (q == 3) split_pair(f[2], f[3], phi.y, plo.y);
  };
  // The same transform as piece_elem for an interior tile, cut by DEPENDENCE LEVEL instead of by
  // channel: a slot holds the four independent instructions of one level (one per channel), the
  // level that consumes them sits in the next slot, 32 cycles later.  With one wave per SIMD nothing
  // else covers the latency of a dependent VALU or transcendental op: cut by channel pair (two
  // two-deep chains per slot) the same instructions ran at 42 cycles per MFMA instead of 34.
  //   arithmetic levels (x*sigmoid(x) modes): 0 z = x*s1 + t1 | 1 w = z * -log2(e) | 2 w = exp2(w) |
  //   3 w = 1 + w | 4 w = rcp(w) | 5 f = z * w | 6 (second affine) f = f*s2 + t2
  //   split levels: 0 hi = bf16 pairs | 1 hi as floats | 2 f - hi | 3 lo = bf16 pairs
  float z[4], w[4];
  auto piece_level = [&](Tile T, int j, int lv) {
    if (j == WS_DPW - 1 && wave_ + 4 * j >= WS_NDMA) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if constexpr (MODE == 0) {
        if (lv == 0) f[q] = vx[0][q];
      } else if constexpr (MODE == 5) {
        if (lv == 0) f[q] = fmaf(vx[0][q], s2[q], t2[q]);
      } else if constexpr (SWISH) {                        // the operations of sa_swish, in its order
        if (lv == 0) z[q] = fmaf(vx[0][q], s1[q], t1[q]);
        if (lv == 1) w[q] = z[q] * -1.4426950408889634f;
        if (lv == 2) w[q] = (SA_ABL & 128) ? w[q] * 0.5f : __builtin_amdgcn_exp2f(w[q]);     // (128: timing-only, no transcendentals)
        if (lv == 3) w[q] = 1.0f + w[q];
        if (lv == 4) w[q] = (SA_ABL & 128) ? w[q] * 0.25f : __builtin_amdgcn_rcpf(w[q]);
        if (lv == 5) f[q] = z[q] * w[q];
        if constexpr (AFF2) { if (lv == 6) f[q] = fmaf(f[q], s2[q], t2[q]); }
      } else {                                             // PRO2 (experiment build)
        if (lv == 0) z[q] = fmaf(k2[q], vy[0][q], k3[q]);
        if (lv == 1) f[q] = fmaf(k1[q], vx[0][q], z[q]);
        if (lv == 2) { if (a.nb_relu_mask) f[q] = vy[0][q] > 0.0f ? f[q] : 0.0f; }
        if (lv == 3) { if (a.nb_colsum && owns(T, false, j)) csum[q] += f[q]; }
      }
    }
  };
  auto piece_split = [&](int j, int lv) {
    if (j == WS_DPW - 1 && wave_ + 4 * j >= WS_NDMA) return;
    if (lv == 0) {
      // (opaque: see split_pair -- the split must see the ROUNDED product)
      asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
      phi.x = ws_cvt_pk_bf16(f[0], f[1]); phi.y = ws_cvt_pk_bf16(f[2], f[3]);
    }
    if (lv == 1) {
      z[0] = __uint_as_float(phi.x << 16); z[1] = __uint_as_float(phi.x & 0xffff0000u);
      z[2] = __uint_as_float(phi.y << 16); z[3] = __uint_as_float(phi.y & 0xffff0000u);
    }
    if (lv == 2) {
#pragma unroll
      for (int q = 0; q < 4; ++q) w[q] = f[q] - z[q];
    }
    if (lv == 3) { plo.x = ws_cvt_pk_bf16(w[0], w[1]); plo.y = ws_cvt_pk_bf16(w[2], w[3]); }
  };
  auto piece_pstat = [&](Tile T, int j) {                   // pro_stats of an interior tile's piece (slot form)
    if constexpr (PSTAT) {
      if (j == WS_DPW - 1 && wave_ + 4 * j >= WS_NDMA) return;
      if (owns(T, false, j)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { csum[q] += f[q]; csq[q] = fmaf(f[q], f[q], csq[q]); }
      }
    }
  };
  auto piece_write = [&](int j, int part = 2) {             // operand planes of the transform tile (0 hi, 1 lo, 2 both)
    if (j == WS_DPW - 1 && wave_ + 4 * j >= WS_NDMA) return;
    unsigned char* dst = smem + (pl_cur + j * (4 * RPP * WS_PITCH * 2));
    if (part != 1) *reinterpret_cast<uint2*>(dst) = phi;
    if (part != 0) *reinterpret_cast<uint2*>(dst + WS_PLANE * 2) = plo;
  };
  auto piece_cache = [&](Tile T, bool edge, int j) {          // bf16 operand cache for sa_wgrad: hi values of the owned rows
    if (j == WS_DPW - 1 && wave_ + 4 * j >= WS_NDMA) return;
    if (has_ao && owns(T, edge, j)) ws_store_b64(aobase_t + j * (4 * RPP * WS_C * 2), ao_off, phi);
  };
  // per-tile column reductions of the transform (PRO2: column sums of d y; pro_stats: sum and sum of
  // squares of the transformed rows): fold the two row halves of the wave, one LDS slot per wave,
  // summed in wave order by 128 threads after the tile barrier
  constexpr int NRED = PSTAT ? 2 : 1;
  auto colsum_put = [&](int it) {
    if constexpr (COLRED) {
      if (PSTAT || a.nb_colsum) {
        float* colred = reinterpret_cast<float*>(raw + (PRO2 ? 2 : 1) * WS_RAW_BYTES) + (size_t)(it & 1) * 4 * WS_C * NRED;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float v = csum[q] + __int_as_float(__builtin_amdgcn_ds_bpermute((int)swap_off, __float_as_int(csum[q])));
          if (lane_ < 32) colred[(wave_ * WS_C + (lane_ & 31) * 4 + q) * NRED] = v;
          csum[q] = 0.0f;
          if constexpr (PSTAT) {
            const float u = csq[q] + __int_as_float(__builtin_amdgcn_ds_bpermute((int)swap_off, __float_as_int(csq[q])));
            if (lane_ < 32) colred[(wave_ * WS_C + (lane_ & 31) * 4 + q) * NRED + 1] = u;
            csq[q] = 0.0f;
          }
        }
      }
    }
  };
  auto colsum_out = [&](int t, int it) {                    // after the barrier that follows colsum_put(it)
    if constexpr (COLRED) {
      if ((PSTAT || a.nb_colsum) && tid < WS_C) {
        const float* colred = reinterpret_cast<const float*>(raw + (PRO2 ? 2 : 1) * WS_RAW_BYTES) + (size_t)(it & 1) * 4 * WS_C * NRED;
        float* dst = PSTAT ? a.pro_stats : a.nb_colsum;
#pragma unroll
        for (int r = 0; r < NRED; ++r)
          dst[((size_t)t * WS_C + tid) * NRED + r] =
              (colred[tid * NRED + r] + colred[(WS_C + tid) * NRED + r]) +
              (colred[(2 * WS_C + tid) * NRED + r] + colred[(3 * WS_C + tid) * NRED + r]);
      }
    }
  };

  // ---- epilogue of a tile from the (copied) accumulator registers, one register per call ----
  f32x16 acc[2], accp[2];
  float ssum = 0.0f, ssq = 0.0f;
  auto epi_form = [&](int n) {                              // n = 16 m + i: bias (+ReLU) of one accumulator register
    float val = accp[n >> 4][n & 15] + bv;
    asm("v_max_f32 %0, %0, %1" : "+v"(val) : "v"(relu_floor));      // (fmaxf canonicalises both operands first)
    return val;
  };
  float eval = 0.0f;                                        // the value the NEXT slot stores (formed one slot ahead)
  auto epi_store = [&](int n) {                             // slot form, full tiles: the store of value n
    const int m = n >> 4, i = n & 15;
    const int ro = m * 32 + (i & 3) + 8 * (i >> 2);         // row in the tile, before the lane half's +4
    if constexpr ((SA_ABL & 64) == 0) ws_store_b32(ybase_e + ro * (UU * WS_CO * 4), y_off, eval);   // (64: timing-only build without the stores)
  };
  auto epi_accum = [&](int n) {                             // ... its statistics, and value n+1 formed
    // (asm: hipcc otherwise sinks all 32 accumulations into the statistics slot, behind its branch)
    asm volatile("v_add_f32 %0, %0, %2\n\tv_fmac_f32 %1, %2, %2" : "+v"(ssum), "+v"(ssq) : "v"(eval));
    if (n + 1 < 32) eval = epi_form(n + 1);
  };
  auto epi_value = [&](Tile T, bool partial, int n) {       // whole value n, optionally bounds-checked
    if (!partial) {
      epi_store(n);
      epi_accum(n);
    } else {
      WS_IDS;
      const int m = n >> 4, i = n & 15;
      const int ro = m * 32 + (i & 3) + 8 * (i >> 2);
      const float val = epi_form(n);
      const int orow = UU == 2 ? 2 * (64 * rh_ + ro + 4 * (lane >> 5)) + ph_ : 64 * rh_ + ro + 4 * (lane >> 5);
      if (T.tile * WS_TM + orow < a.Lout) {
        *reinterpret_cast<float*>(ybase_e + ro * (UU * WS_CO * 4) + y_off) = val;
        ssum += val; ssq = fmaf(val, val, ssq);
      }
    }
  };
  // statistics of a tile: fold the two lane halves (a lane owns one column, the other half holds the
  // rows +4) and store; both halves then hold the same sums and write the same slab entry
  float st_s = 0.0f, st_q = 0.0f;
  bool st_slab_ok = true;                                   // (two slabs per tile: does this wave's slab exist?)
  bool st_pending = false;                                  // (64 channels) partial sums wait in LDS for the tile barrier
  int st_it = 0;
  char* stbase_e = nullptr;                                 // statistics slab of the epilogue tile
  auto epi_stats = [&](Tile T, int part = 2) {              // 0: fold the lane halves, 1: store, 2: both
    if (has_stats) {
      if (part != 1) {
        st_s = ssum + __int_as_float(__builtin_amdgcn_ds_bpermute((int)swap_off, __float_as_int(ssum)));
        st_q = ssq + __int_as_float(__builtin_amdgcn_ds_bpermute((int)swap_off, __float_as_int(ssq)));
      }
      if (part != 0) {
        if constexpr (G::COMBINE == 1) {
          ws_store_b64(stbase_e, st_off, make_uint2(__float_as_uint(st_s), __float_as_uint(st_q)));
        } else {                                           // two row halves share a column: summed after the tile barrier
          float2* stred = reinterpret_cast<float2*>(raw + WS_RAW_BYTES) + (size_t)(st_it & 1) * 4 * 32;
          if (lane_ < 32) stred[wave_ * 32 + lane_] = make_float2(st_s, st_q);
          st_pending = true;
        }
      }
    }
    if (part != 0) { ssum = 0.0f; ssq = 0.0f; }
  };
  // (64 channels) after the barrier that follows epi_stats: the waves of row half 0 add the two halves
  auto stats_out = [&]() {
    if constexpr (G::COMBINE == 2) {
      // (the leader of a pair: row half 0 / phase 0; its partner is NWN waves further; a transposed
      // tile's second row half may lie wholly beyond the utterance: no slab there)
      const bool leader = UU == 2 ? ph_ == 0 : wm_ == 0;
      if (st_pending && leader && st_slab_ok) {
        const float2* stred = reinterpret_cast<const float2*>(raw + WS_RAW_BYTES) + (size_t)(st_it & 1) * 4 * 32;
        const float2 p0 = stred[wave_ * 32 + (lane_ & 31)], p1 = stred[(wave_ + NWN) * 32 + (lane_ & 31)];
        ws_store_b64(stbase_e, st_off, make_uint2(__float_as_uint(p0.x + p1.x), __float_as_uint(p0.y + p1.y)));
      }
      st_pending = false;
      ++st_it;
    }
  };

  int toff[WS_NTAPS];
#pragma unroll
  for (int t = 0; t < WS_NTAPS; ++t) toff[t] = (a.taps.off[ph_][UU == 1 || t < ntap_ ? t : 0] - a.rowmin) * WS_PITCH;

  // ================= prologue: first tile staged without overlap =================
  // Past the end of the range the "next" tiles are clamped to its last one: the transform / DMA
  // slots then repeat that tile's work (same values to the same places) instead of branching.
  Tile Tc = tile_of(first), Tn = tile_of(first + 1 < last ? first + 1 : last - 1), Tp = Tc;
#pragma unroll
  for (int q = 0; q < 4; ++q) { csum[q] = 0.0f; csq[q] = 0.0f; }
  {
    const bool ec = is_edge(Tc), en = is_edge(Tn);
    xbase_d = row_ptr(a.x, Tc.irow, WS_C * 4);
    if constexpr (PRO2) x2base_d = row_ptr(a.nb_x, Tc.irow, WS_C * 4);
    aobase_t = const_cast<char*>(row_ptr(a.a_out, Tc.irow, WS_C * 2));
    pl_cur = pl_off;
#pragma unroll
    for (int j = 0; j < WS_DPW; ++j) dma_piece(Tc, ec, j);
    xbase_d = row_ptr(a.x, Tn.irow, WS_C * 4);
    if constexpr (PRO2) x2base_d = row_ptr(a.nb_x, Tn.irow, WS_C * 4);
    if (SWISH || PRO2) load_consts(Tc.b);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < WS_DPW; ++j) {
      piece_read(j);
#pragma unroll
      for (int q = 0; q < 4; ++q) piece_elem(Tc, ec, j, q);
      piece_write(j);
      piece_cache(Tc, ec, j);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the raw piece is in registers before its refill
      dma_piece(Tn, en, j);
    }
  }
  colsum_put(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  colsum_out(first, 0);

  // ================= the tile walk =================
  // iteration (t, it): MFMA loop of tile t from planes[it & 1]; in its filler slots the epilogue of
  // tile t-1 (from the copied accumulators), the transform of tile t+1 into planes[(it+1) & 1] and
  // the refill DMA of tile t+2.  One barrier per tile.
  asm volatile("" : "=v"(accp[0]), "=v"(accp[1]));        // (read by the dummy epilogue of a range's first tile)
  for (int t = first, it = 0; t < last; ++t, ++it) {
    int lanem = lane_;
    asm volatile("" : "+v"(lanem));
    const bf16_t* ab = planes + (size_t)(it & 1) * 2 * WS_PLANE + ((lanem & 31) + 64 * rh_) * (WS_SA * WS_PITCH) + (lanem >> 5) * 8;
    // A fragments of the next step, read behind MFMAs 0 and 1 of this one, i.e. four MFMAs or more
    // ahead of their use (hipcc waits for all of them once, in front of the next step's MFMA 0): the
    // lo halves (MFMAs 0, 1) have one slot, refilled behind their last use, the hi halves (2..5) two
    bf16x8 ah[2][2], al[2];
    auto a_ptr = [&](int s, int m) {
      const int tp = s / WS_KSTEPS, k = s % WS_KSTEPS;
      return ab + toff[tp] + m * 32 * (WS_SA * WS_PITCH) + k * 16;
    };
    auto load_ah = [&](int s, int m) { ah[s & 1][m] = *reinterpret_cast<const bf16x8*>(a_ptr(s, m)); };
    auto load_al = [&](int s, int m) { al[m] = *reinterpret_cast<const bf16x8*>(a_ptr(s, m) + WS_PLANE); };
    // the first step's fragments are requested here, behind the barrier: their LDS latency passes
    // under the scalar set-up of the iteration below
    load_al(0, 0); load_al(0, 1); load_ah(0, 0); load_ah(0, 1);
    __builtin_amdgcn_sched_barrier(0);
    const bool doE = t > first, doT = t + 1 < last;
    const Tile Tnn = t + 2 < last ? next_tile(Tn) : Tn;       // (clamped: Tn is already the last tile then)
    const bool edgeT = is_edge(Tn), edgeD = is_edge(Tnn);
    const bool partialE = Tp.tile > part_hi;
    // the slots hold the interior forms only; a tile at the end of an utterance (2 of 315 at the
    // training length) gets its masked transform / clamped DMA after the loop, a partial output tile
    // its bounds-checked epilogue in front of it, not overlapped
    // (slotE does not ask for doE: in the first iteration of a range the epilogue slots run on the undefined
    // accumulator copy with Tp = Tc and store into the rows and the statistics slab of THIS tile, which the next
    // iteration's real epilogue overwrites -- same waves, same addresses, stores retire in order -- so the first
    // tile is overlapped like the others)
    // (not with pro_stats: that instance's register allocation tips into scratch with the first iteration in
    // the overlapped body -- tools/ws_audit.py --, so its first tile stays plain)
    const bool slotE = (PSTAT || (bcost & 0x10000) ? doE : true) && !partialE, slotT = !edgeT, slotD = !edgeT && !edgeD;
    if ((SWISH || PRO2) && Tn.b != cur_b) load_consts(Tn.b);
    xbase_d = row_ptr(a.x, Tnn.irow, WS_C * 4);
    if constexpr (PRO2) x2base_d = row_ptr(a.nb_x, Tnn.irow, WS_C * 4);
    aobase_t = const_cast<char*>(row_ptr(a.a_out, Tn.irow, WS_C * 2));
    ybase_e = const_cast<char*>(row_ptr(a.y, Tp.orow, WS_CO * 4));
    stbase_e = const_cast<char*>(row_ptr(a.stats, Tp.srow, WS_CO * 8));
    st_slab_ok = G::SLABS == 1 || Tp.tile * 2 + rh_ < a.ntiles * 2 - slab_short;
    pl_cur = (it + 1) & 1 ? pl_off + WS_BUF_BYTES : pl_off;
    asm volatile("" : "+v"(pl_cur));
    if (doE && partialE) {
#pragma unroll
      for (int n = 0; n < 32; ++n) epi_value(Tp, true, n);
      epi_stats(Tp);
    }
    WS_STAMP(it, 0);
    // One tile: 240 single-MFMA asm statements with filler slot f behind statement f.  FAST: the
    // steady state (a full previous tile to store, interior tiles to transform and to fetch) with
    // every slot filled, unconditionally; otherwise the bare MFMA loop, the other work around it.
    auto tile_body = [&](auto fast_c) {
      constexpr bool FAST = decltype(fast_c)::value;
      auto filler = [&](auto f_c) {
        constexpr int fs = decltype(f_c)::value;
        if constexpr ((SA_ABL & 8) != 0 || !FAST) return;
        if constexpr (fs < 32) {
          epi_store(fs); epi_accum(fs);
        } else if constexpr (fs == 32 || fs == 33) {
          epi_stats(Tp, fs - 32);
        } else if constexpr (fs == 34) {
          WS_STAMP(it, 3);
        } else if constexpr (fs == WS_FT - 1) {
          WS_STAMP(it, 4);
          // Everything up to the refill DMA of the previous tile loop has landed: vector-memory
          // operations complete in order, and behind that DMA this wave has issued exactly the 32
          // stores (+ 1 statistics store) of the slots above, which may stay in flight.
          if (G::COMBINE == 1 && has_stats) asm volatile("s_waitcnt vmcnt(33)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");      // (64 channels: the statistics are stored behind the barrier)
          WS_STAMP(it, 5);
          piece_read(0);
        } else if constexpr (fs >= WS_FT && WS_SUBS == 18) {
          // 18 slots per piece = three steps: arithmetic levels in slots 0..6, split levels in 8..11,
          // operand cache, pro_stats, refill DMA behind them; the raw piece is consumed by level 0 (the
          // next one is read in slot 1) and the operand planes of piece j are written in slots 6 / 7
          // of piece j+1, i.e. slots 0 / 1 of a step (the last piece's after the loop)
          constexpr int j = (fs - WS_FT) / 18, k = (fs - WS_FT) % 18;
          constexpr int NA = AFF2 ? 7 : 6;                  // arithmetic levels of this mode (PRO2: 4, padded)
          if constexpr (j > 0 && (k == 6 || k == 7)) piece_write(j - 1, k - 6);
          if constexpr (k < NA) piece_level(Tn, j, k);
          if constexpr (j + 1 < WS_DPW && k == 1) piece_read(j + 1, 0);
          if constexpr (j + 1 < WS_DPW && k == 13 && PRO2) piece_read(j + 1, 1);
          if constexpr (k >= 8 && k < 12) piece_split(j, k - 8);
          if constexpr (k == 14) piece_cache(Tn, false, j);
          if constexpr (k == 15) piece_pstat(Tn, j);
          if constexpr (k == 16) dma_piece(Tnn, false, j, 0);
          if constexpr (k == 17 && PRO2) dma_piece(Tnn, false, j, 1);
        } else if constexpr (fs >= WS_FT && WS_SUBS == 4) {
          // 72 slots per tile (64 -> 32 transposed): 4 slots per piece, plain rows
          constexpr int j = (fs - WS_FT) / 4, k = (fs - WS_FT) % 4;
          if constexpr (j > 0 && k < 2) piece_write(j - 1, k);
          if constexpr (k == 0) { piece_level(Tn, j, 0); piece_split(j, 0); }
          if constexpr (j + 1 < WS_DPW && k == 1) piece_read(j + 1, 0);
          if constexpr (k == 1) piece_split(j, 1);
          if constexpr (k == 2) { piece_split(j, 2); piece_split(j, 3); }
          if constexpr (k == 3) { piece_cache(Tn, false, j); dma_piece(Tnn, false, j); }
        } else if constexpr (fs >= WS_FT && WS_SUBS == 6) {
          // 64 channels: a tile has 120 slots for the same transform work: 6 slots per piece = one
          // step, two dependence levels per slot (over-full on purpose: these launches are HBM-bound)
          constexpr int j = (fs - WS_FT) / 6, k = (fs - WS_FT) % 6;
          if constexpr (j > 0 && k < 2) piece_write(j - 1, k);
          if constexpr (k == 0) { piece_level(Tn, j, 0); piece_level(Tn, j, 1); }
          if constexpr (j + 1 < WS_DPW && k == 1) piece_read(j + 1, 0);
          if constexpr (k == 1) { piece_level(Tn, j, 2); piece_level(Tn, j, 3); }
          if constexpr (k == 2) { piece_level(Tn, j, 4); piece_level(Tn, j, 5); }
          if constexpr (k == 3) { piece_split(j, 0); piece_split(j, 1); }
          if constexpr (k == 4) { piece_split(j, 2); piece_split(j, 3); }
          if constexpr (k == 5) { piece_cache(Tn, false, j); dma_piece(Tnn, false, j); }
        } else if constexpr (fs >= WS_FT) {                 // 12 slots per piece = two steps: one arithmetic level
          constexpr int j = (fs - WS_FT) / 12, k = (fs - WS_FT) % 12;
          if constexpr (j > 0 && k < 2) piece_write(j - 1, k);                  // (before this piece's split overwrites hi / lo)
          if constexpr (k == 0) piece_level(Tn, j, 0);
          if constexpr (j + 1 < WS_DPW && k == 1) piece_read(j + 1, 0);
          if constexpr (k >= 2 && k < 6) piece_split(j, k - 2);
          if constexpr (k == 9) piece_cache(Tn, false, j);
          if constexpr (k == 10) dma_piece(Tnn, false, j);
        }
      };
      // The MFMAs are inline asm so that the weight fragments are AGPR operands where they live
      // (left to itself hipcc parks them in AGPRs and copies each one to VGPRs in front of every
      // use).  The first WS_NAGPR_FRAGS (tap, k-step) pairs take 240 of the 256 AGPRs, the rest stay
      // in VGPRs.  Hazards (cdna_hip_programming.md 5.7 item 2): an accumulate chain needs no wait
      // states; the A fragments come from ds_read (waited for by hipcc, which sees the operand); the
      // accumulators are read by VALU code only after the s_nop block below (tools/ws_audit.py
      // checks the ISA).
      ws_static_for<0, WS_NTAPS * WS_KSTEPS>([&](auto s_c) {
        constexpr int s = decltype(s_c)::value;
        constexpr int tp = s / WS_KSTEPS, k = s % WS_KSTEPS, sl = s & 1;
        constexpr bool more = s + 1 < WS_NTAPS * WS_KSTEPS;
        __builtin_amdgcn_sched_barrier(0);
#define WS_MFMA1(M, A, BC, B) \
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[M]) : "v"(A), BC(B))
#define WS_AL(M) if constexpr (more) { __builtin_amdgcn_sched_barrier(0); load_al(s + 1, M); load_ah(s + 1, M); }
#define WS_AH(M)
#define WS_SLOT(I) __builtin_amdgcn_sched_barrier(0); filler(std::integral_constant<int, 6 * s + (I)>{}); __builtin_amdgcn_sched_barrier(0)
        if constexpr ((SA_ABL & 1) != 0) {
          asm volatile("" :: "v"(ah[sl][0]), "v"(al[0]), "v"(ah[sl][1]), "v"(al[1]));
          WS_AL(0); WS_AL(1); WS_AH(0); WS_AH(1);
          WS_SLOT(0); WS_SLOT(1); WS_SLOT(2); WS_SLOT(3); WS_SLOT(4); WS_SLOT(5);
        } else {
          if constexpr (s == 0) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc[0]) : "v"(al[0]), "a"(Bh[0][0]));
            WS_AL(0); WS_SLOT(0);
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc[1]) : "v"(al[1]), "a"(Bh[0][0]));
            WS_AL(1); WS_SLOT(1);
          } else if constexpr (s < WS_NAGPR_FRAGS) {
            WS_MFMA1(0, al[0], "a", Bh[tp][k]); WS_AL(0); WS_SLOT(0);
            WS_MFMA1(1, al[1], "a", Bh[tp][k]); WS_AL(1); WS_SLOT(1);
          } else {
            WS_MFMA1(0, al[0], "v", Bh[tp][k]); WS_AL(0); WS_SLOT(0);
            WS_MFMA1(1, al[1], "v", Bh[tp][k]); WS_AL(1); WS_SLOT(1);
          }
          if constexpr (s < WS_NAGPR_FRAGS) {
            WS_MFMA1(0, ah[sl][0], "a", Bl[tp][k]); WS_AH(0); WS_SLOT(2);
            WS_MFMA1(1, ah[sl][1], "a", Bl[tp][k]); WS_AH(1); WS_SLOT(3);
            WS_MFMA1(0, ah[sl][0], "a", Bh[tp][k]); WS_SLOT(4);
            WS_MFMA1(1, ah[sl][1], "a", Bh[tp][k]); WS_SLOT(5);
          } else {
            WS_MFMA1(0, ah[sl][0], "v", Bl[tp][k]); WS_AH(0); WS_SLOT(2);
            WS_MFMA1(1, ah[sl][1], "v", Bl[tp][k]); WS_AH(1); WS_SLOT(3);
            WS_MFMA1(0, ah[sl][0], "v", Bh[tp][k]); WS_SLOT(4);
            WS_MFMA1(1, ah[sl][1], "v", Bh[tp][k]); WS_SLOT(5);
          }
        }
#undef WS_MFMA1
#undef WS_SLOT
#undef WS_AL
#undef WS_AH
      });
    };
    const bool fast = slotE && slotT && slotD;
    if (fast) {
      tile_body(std::true_type{});
      piece_write(WS_DPW - 1);                             // (its slots 0 / 1 of a next piece do not exist)
    } else {
      if (doE && !partialE) {                              // (a partial tile had its epilogue above)
#pragma unroll
        for (int n = 0; n < 32; ++n) epi_value(Tp, false, n);
        epi_stats(Tp);
      }
      tile_body(std::false_type{});
    }
    WS_STAMP(it, 1);
    if (!fast) {                                           // transform of the next tile (masked if it is at the end of an
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // utterance) and the DMA of the one after, not overlapped
#pragma unroll
      for (int j = 0; j < WS_DPW; ++j) {
        piece_read(j);
#pragma unroll
        for (int q = 0; q < 4; ++q) piece_elem(Tn, edgeT, j, q);
        piece_write(j);
        piece_cache(Tn, edgeT, j);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the raw pieces are in registers before their refill
#pragma unroll
      for (int j = 0; j < WS_DPW; ++j) dma_piece(Tnn, edgeD, j);
    }
    colsum_put(it + 1);
    // MFMA result -> VALU reader wait states (two back-to-back 8-pass MFMAs have just been issued:
    // 2 x 32 cycles + write-back), then the accumulators move to their epilogue copy.  The copy is
    // asm on purpose: given a plain assignment hipcc coalesces it away by MIGRATING the live
    // accumulators to the copy's registers in the middle of the MFMA loop (v_mov_b64 right behind an
    // asm MFMA whose latency it does not know: stale values in the first registers).
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(acc[0]), "+v"(acc[1]));
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float v;
        asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "v"(acc[m][i]));
        accp[m][i] = v;
      }
    eval = epi_form(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // planes[(it+1) & 1] complete; planes[it & 1] free
    WS_STAMP(it, 2);
    stats_out();
    if (doT) colsum_out(t + 1, it + 1);
    Tp = Tc; Tc = Tn; Tn = Tnn;
  }
  // ================= tail: epilogue of the last tile =================
  {
    const bool partialE = Tp.tile > part_hi;
    ybase_e = const_cast<char*>(row_ptr(a.y, Tp.orow, WS_CO * 4));
    stbase_e = const_cast<char*>(row_ptr(a.stats, Tp.srow, WS_CO * 8));
    st_slab_ok = G::SLABS == 1 || Tp.tile * 2 + rh_ < a.ntiles * 2 - slab_short;
    if (!partialE) eval = epi_form(0);
#pragma unroll
    for (int n = 0; n < 32; ++n) epi_value(Tp, partialE, n);
    epi_stats(Tp);
    if constexpr (G::COMBINE == 2) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      stats_out();
    }
  }
#undef WS_IDS
}

// extra cost of an utterance end, in tiles (see the kernel's range computation)
int g_ws_bcost = 9;

template <int MODE, int NT = 5, int HALO = 4, int CC = 128, int CO = CC, int SA = 1, int UU = 1>
int launch_ws(const SaConvArgs& a, hipStream_t st) {
  typedef WsGeo<CC, NT, HALO, CO, SA, UU> G;
  constexpr int WS_NTAPS = G::NTAPS, WS_HALO = G::HALO, WS_ROWS = G::ROWS, WS_RAW_BYTES = G::RAW_BYTES, WS_BUF_BYTES = G::BUF_BYTES,
                WS_C = G::C, WS_KSTEPS = G::KSTEPS;
  SaConvArgs args = a;
  args.ntiles = sa_div_up(sa_div_up(a.Lout, UU), G::BM);
  int omin = 1 << 30, omax = -(1 << 30), wmax = 0;
  for (int ph = 0; ph < UU; ++ph)
    for (int t = 0; t < (UU == 1 ? WS_NTAPS : a.taps.ntaps[ph]); ++t) {
      omin = a.taps.off[ph][t] < omin ? a.taps.off[ph][t] : omin;
      omax = a.taps.off[ph][t] > omax ? a.taps.off[ph][t] : omax;
      wmax = a.taps.widx[ph][t] > wmax ? a.taps.widx[ph][t] : wmax;
    }
  if (omax - omin != WS_HALO) return -22;
  args.rowmin = omin;
  args.nrows = WS_ROWS;
  args.wlo_off = (wmax + 1) * WS_KSTEPS * G::NWN * 64;   // fragment units: size of the hi image
  if ((a.a_out || a.nb_colsum || a.pro_stats) &&
      (omin > 0 || omax < 0 || (args.ntiles - 1) * G::BM * SA + omin + WS_ROWS < a.Lin))
    return -22;                                           // every input row must be staged by the tile that owns it
  if ((long)a.B * a.Lin >= (1L << 31) - 64 || (long)a.B * a.Lout >= (1L << 31) - 64) return -22;   // 32-bit row indices in the kernel
  const size_t lds = 2 * WS_BUF_BYTES + (MODE == 2 ? 2 : 1) * WS_RAW_BYTES +
                     (MODE == 2 ? 2 * 4 * WS_C * 4 : MODE == 3 ? 2 * 4 * WS_C * 8 : G::COMBINE == 2 ? 2 * 4 * 32 * 8 : 0);
  auto kern = sa_conv_ws_kernel<MODE, NT, HALO, CC, CO, SA, UU>;
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
  const int total = args.ntiles * a.B;
  const int nwg = total < n_cu ? total : n_cu;
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, st, args, g_ws_bcost, total, g_ws_xcd_weights);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

}  // namespace

// relative speed of the eight XCDs under the persistent kernels, one byte each (64 = nominal)
unsigned long long g_ws_xcd_weights = 0x4040404040404040ull;

extern "C" int sa_conv_ws_set_xcd_weights(const unsigned char* w8) {
  if (!w8) return -22;
  unsigned long long v = 0;
  for (int x = 0; x < 8; ++x) {
    if (w8[x] < 16) return -22;                              // (a zero weight would starve an XCD's workgroups)
    v |= (unsigned long long)w8[x] << (8 * x);
  }
  g_ws_xcd_weights = v;
  return 0;
}

extern "C" int sa_conv_ws_set_bcost(int tiles) {
  // (bit 16, timing A/B only: the first tile of every range takes the plain path, as before round 3's overlap of it)
  if ((tiles & 0xffff) < 4 || (tiles & 0xffff) > 64 || (tiles & ~0x1ffff)) return -22;
  g_ws_bcost = tiles;
  return 0;
}

// Does the weight-stationary kernel serve this launch?  (sa_conv_gemm.hip asks before routing.)
bool sa_conv_ws_covers(int dtype, int cin, int cout, int sa, int u, const SaConvArgs* a) {
  if (dtype == SA_BF16X3 && ((cin == 128 && cout == 64) || (cin == 64 && cout == 32)) && sa == 1 && u == 2) {   // decoder.1 / decoder.5: transposed
    if (a->taps.ntaps[0] > 3 || a->taps.ntaps[1] > 3 || a->taps.ntaps[0] < 1 || a->taps.ntaps[1] < 1) return false;
    int lo = 1 << 30, hi = -(1 << 30);
    for (int ph = 0; ph < 2; ++ph)
      for (int t = 0; t < a->taps.ntaps[ph]; ++t) {
        lo = a->taps.off[ph][t] < lo ? a->taps.off[ph][t] : lo;
        hi = a->taps.off[ph][t] > hi ? a->taps.off[ph][t] : hi;
      }
    if (hi - lo != 2 || lo > 0 || hi < 0) return false;
    if (a->tile_rows && a->tile_rows != 128) return false;
    return !a->s1 && !a->swish && !a->s2 && !a->t2 && !a->pro_stats && !a->nb_x && !a->ep_mode && !a->wscale && !a->relu;
  }
  const bool s2layer = cin == 64 && cout == 128 && sa == 2;                // encoder.8: 64 -> 128, stride 2
  if (dtype != SA_BF16X3 || u != 1 || !(s2layer || (cin == cout && sa == 1 && (cin == 128 || cin == 64)))) return false;
  const int nt = a->taps.ntaps[0];
  if (cin == 64 && (nt != 5 || a->s2 || a->t2 || a->pro_stats || a->nb_x)) return false;   // 64 input channels: the forward layers' two forms
  if ((nt != 5 && nt != 3) || a->wscale) return false;
  if (a->ep_mode) return false;
  int omin = 1 << 30, omax = -(1 << 30);
  for (int t = 0; t < nt; ++t) {
    omin = a->taps.off[0][t] < omin ? a->taps.off[0][t] : omin;
    omax = a->taps.off[0][t] > omax ? a->taps.off[0][t] : omax;
  }
  // 5 taps over 4 rows (unit spacing); 3 taps over 4 or 6 rows (dilation 2, 3)
  if (nt == 5 ? omax - omin != 4 : (omax - omin != 4 && omax - omin != 6)) return false;
  // the slots' ownership test (operand cache, column sums, pro_stats) works on whole DMA pieces (row
  // pairs): the tile's own rows must start at an even staged row
  if ((a->a_out || a->nb_colsum || a->pro_stats) && (omin > 0 || (omin & 1))) return false;
  if (a->tile_rows && a->tile_rows != (cout == 64 ? 128 : 64)) return false;
#ifndef SA_WS_PRO2
  if (a->nb_x) return false;                              // data gradients stay on the one-tile kernel (header)
#else
  if (a->nb_x) return nt == 5 && !a->s1 && !a->swish && a->nb_c1 && a->nb_c2 && a->nb_c3;
#endif
  if (nt == 3)                                            // the dilated TDNN layers: per-channel affine in front, or nothing
    return !a->s1 && !a->swish && !a->pro_stats && (a->s2 || !a->t2);
  if ((a->s2 || a->t2) && (!a->s2 || !a->s1 || a->pro_stats)) return false;    // second affine: behind affine + activation only
  if (a->pro_stats && !a->s1) return false;
  if (a->s1) return a->t1 && a->swish;                    // affine + x*sigmoid(x)
  return !a->swish;
}

// output rows per tile (the one-tile kernel must be on the same tile height: same slab geometry)
// tile height of the one-tile kernel this kernel's slab geometry matches, and its own output rows per tile
int sa_conv_ws_tile_rows(int cout) { return cout == 128 ? 64 : 128; }
int sa_conv_ws_rows_per_tile(int cout) { return cout == 128 ? 64 : cout == 64 ? 128 : 256; }

int sa_conv_ws_dispatch(int cin, int cout, const SaConvArgs* a, hipStream_t st) {
  if (cin == 128 && cout == 64) return launch_ws<0, 3, 2, 128, 64, 1, 2>(*a, st);
  if (cin == 64 && cout == 32) return launch_ws<0, 3, 2, 64, 32, 1, 2>(*a, st);
  if (cin == 64 && cout == 128) return a->s1 ? launch_ws<1, 5, 4, 64, 128, 2>(*a, st) : launch_ws<0, 5, 4, 64, 128, 2>(*a, st);
  if (cin == 64) return a->s1 ? launch_ws<1, 5, 4, 64>(*a, st) : launch_ws<0, 5, 4, 64>(*a, st);
  if (a->taps.ntaps[0] == 3) {
    int omin = 1 << 30, omax = -(1 << 30);
    for (int t = 0; t < 3; ++t) {
      omin = a->taps.off[0][t] < omin ? a->taps.off[0][t] : omin;
      omax = a->taps.off[0][t] > omax ? a->taps.off[0][t] : omax;
    }
    if (omax - omin == 4) return a->s2 ? launch_ws<5, 3, 4>(*a, st) : launch_ws<0, 3, 4>(*a, st);
    return a->s2 ? launch_ws<5, 3, 6>(*a, st) : launch_ws<0, 3, 6>(*a, st);
  }
#ifdef SA_WS_PRO2                                         // experiment build: MODE 2 without a fused epilogue
  if (a->nb_x) return launch_ws<2>(*a, st);
#endif
  if (a->s1 && a->pro_stats) return launch_ws<3>(*a, st);
  if (a->s1 && a->s2) return launch_ws<4>(*a, st);
  if (a->s1) return launch_ws<1>(*a, st);
  return launch_ws<0>(*a, st);
}
