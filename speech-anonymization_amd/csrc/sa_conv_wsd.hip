// Fused DATA-GRADIENT convolutions 128 -> 128 (bf16x3 policy: fp32 storage, split-bf16 operands) in
// the weight-stationary, persistent, LDS-DMA-pipelined form -- the backward of
// models/ConvAutoEncoder.py:150-158 (encoder.11), :161 (decoder.0) and :33-43 (the three TDNN layers of
// the sex classifier) reached from speechbrain_convae_train.py:241.
//
// One launch = normalisation-backward PROLOGUE of the layer above (d y = c1*dz + c2*y + c3 [* (y > 0)]
// formed while the rows are staged; by-products bf16(d y) for the weight gradient and the column sums
// for the bias gradient) + data-gradient convolution + activation / normalisation backward EPILOGUE of
// the layer below (g' = (acc + g2) * swish'(z), statistics (sum g', sum g'*xhat)).  Same arguments,
// slab geometry and output BITS (y, a_out) as sa_conv_gemm_kernel<bf16x3_t,128,128,1,1,64,PRO2>.
//
// Structure (what sa_conv_ws.hip does for the forward launches, re-cut so that the two extra input
// tensors of a fused data gradient fit the register file):
//   * one persistent 4-wave workgroup per CU, one wave per SIMD, the whole 512-register file; wave w
//     owns output columns [32w, 32w+32) and keeps its weight fragments (taps x 8 k-steps x hi, lo) in
//     registers for the launch (5 taps: 256 AGPRs + 64 VGPRs);
//   * a 64-row tile is computed as two 32-row SECTIONS one after the other, each on ONE 32x32
//     accumulator: acc[0] for rows 0..31, acc[1] for rows 32..63.  The epilogue of a section runs in
//     the issue gaps of the NEXT section's MFMA loop straight from its accumulator -- no copy, and 32
//     accumulator registers instead of the forward kernel's 128, which is what pays for the epilogue's
//     state (stored-tensor values in flight, per-value pipeline registers);
//   * input rows (d z and the stored forward tensor y of the layer above) arrive by LDS-DMA two tiles
//     ahead; the stored tensor of the layer below (ep_x) and the second gradient (ep_g2) are loaded
//     by the lanes that need them (a register of a 32x32 accumulator = two 128-byte row segments per
//     load / store) one section ahead of their use;
//   * every section is NT*24 single-MFMA statements with a filler slot behind each:
//       slots E0..      epilogue of the previous section, software-pipelined over its 16 values (one
//                       dependence level of up to four values per slot), behind COUNTED s_waitcnt
//                       vmcnt(N): N = the vector-memory operations this wave is guaranteed to have
//                       issued after the value's load (WsdSched::nwait) -- later DMA, stores and loads
//                       stay in flight;
//       slots T0..      transform of the next tile's pieces (prologue arithmetic, hi / lo split, operand
//                       planes, bf16 d y, refill DMA of the tile after next) and the loads for the
//                       next section's epilogue;
//     LDS instructions sit in slot 0 of a 3-MFMA step only (hipcc waits lgkmcnt(0) in front of a
//     step's first MFMA, whose A fragments were requested one step ahead);
//   * tiles at the ends of an utterance, partial output tiles, the first tile of a workgroup and a
//     tile range crossing an utterance take a plain path (no overlap, vmcnt(0) everywhere).
// tools/wsd_audit.py checks the ISA (no spills, accumulators pinned, no compiler write to them inside
// a tile body, loaded registers untouched between load and wait).
#include "sa_conv_ws_common.h"

// -DSA_WSD_ABL=<mask>: timing-only ablation builds (tools/wsd_ablate.py; WRONG results, never shipped):
//   1 no epilogue stream (waits, levels, stores, loads, statistics)   2 no transform stream (incl. refill DMA)
//   4 no MFMA   8 no counted waits   16 no epilogue loads and waits   32 no epilogue stores
#ifndef SA_WSD_ABL
#define SA_WSD_ABL 0
#endif

// -DSA_WSD_STAMPS: diagnostic build (tools/wsd_stamps.py): s_memtime at the section boundaries of one
// workgroup's wave 0; no stamp exists in the normal build.
#ifdef SA_WSD_STAMPS
__device__ unsigned long long sa_wsd_dbg[64 * 16];
__device__ unsigned long long sa_wsd_wg[512 * 2];             // s_memrealtime at entry / exit of every workgroup
#define WSD_WG_STAMP(i) do { if (lane_ == 0 && wave_ == 0 && blockIdx.x < 512) { \
  unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
  sa_wsd_wg[blockIdx.x * 2 + (i)] = t_; } } while (0)
#define WSD_STAMP(it, i) do { if (lane_ == 0 && wave_ == 0 && blockIdx.x == 7 && (it) < 64) { \
  unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
  sa_wsd_dbg[(it) * 16 + (i)] = t_; } } while (0)
#define WSD_STAMP_RT(it, i) do { if (lane_ == 0 && wave_ == 0 && blockIdx.x == 7 && (it) < 64) { \
  unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
  sa_wsd_dbg[(it) * 16 + (i)] = t_; } } while (0)
extern "C" int sa_wsd_dbg_read(unsigned long long* out) {
  return -(int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sa_wsd_dbg), sizeof(sa_wsd_dbg));
}
extern "C" int sa_wsd_wg_read(unsigned long long* out) {
  return -(int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sa_wsd_wg), sizeof(sa_wsd_wg));
}
#else
#define WSD_STAMP(it, i)
#define WSD_STAMP_RT(it, i)
#define WSD_WG_STAMP(i)
#endif

extern unsigned long long g_ws_xcd_weights;                  // (sa_conv_ws.hip)
// entry / exit time (s_memrealtime, 100 MHz) of every workgroup of the LAST launch: what
// sa_conv_ws_calibrate_read hands to the host for the per-XCD weights of the tile ranges
__device__ unsigned long long sa_wsd_life[512 * 2];
#define WSD_LIFE(i) do { if (threadIdx.x == 0 && blockIdx.x < 512) { \
  unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
  sa_wsd_life[blockIdx.x * 2 + (i)] = t_; } } while (0)

namespace {

// ---- the slot schedule of one instantiation (compile-time; shared by the filler and the wait counts) ----
// PRO: 0 plain rows (decoder.0's data gradient: its input needs no apply), 2 normalisation-backward prologue
// EP : 1 (acc) * swish'(z), xhat from x            [InstanceNorm + x*sigmoid(x) block]
//      2 acc, xhat from x                          [ReLU + BatchNorm blocks; per-channel mean / rstd]
//      3 acc, xhat from swish(z)                   [the classifier's input BatchNorm on the activation]
//      4 (acc + k1*g2 + k2*swish(z) + k3) * swish'(z)   [mode 1 with the pending BatchNorm apply as addend]
template <int NT_, int HALO_, int PRO_, int EP_>
struct WsdSched {
  typedef WsGeo<128, NT_, HALO_> G;
  static constexpr int NS = NT_ * G::KSTEPS;              // k-steps per section
  static constexpr int SECT = NS * 3;                     // MFMAs = slots per section (120 / 72)
  static constexpr int DPW = G::DPW;
  static constexpr int NP0 = (DPW + 1) / 2, NP1 = DPW - NP0;   // pieces transformed in section 0 / 1
  static constexpr int LPV = EP_ == 4 ? 2 : 1;            // loads per epilogue value
  static constexpr int EII = EP_ == 1 || EP_ == 4 ? 3 : EP_ == 3 ? 2 : 1; // slots between two values
  static constexpr int ELV = EP_ == 1 ? 11 : EP_ == 4 ? 13 : EP_ == 3 ? 9 : 3;  // dependence levels of a value
  static constexpr int LVG = EP_ == 1 ? 9 : EP_ == 4 ? 11 : EP_ == 3 ? 5 : 1;   // level that leaves the final value in the load's register
  // first epilogue slot (>= 3: MFMA -> VALU distance), chosen so that the heavier levels of the three
  // values in flight do not share a slot with the step's A-fragment reads
  static constexpr int E0 = EP_ == 4 ? 5 : 3;             // (EP 1 at 4: measured slower, 2568 against 2316 cycles to the store burst)
  static constexpr int EEND = E0 + 15 * EII + ELV;        // first slot behind the epilogue arithmetic: the statistics
  // Vector-memory loads retire in order among themselves (LDS-DMA included) and stores among themselves,
  // but NOT loads relative to stores: a counted vmcnt(N) is exact only while every operation younger
  // than the awaited load is a load.  So a section first consumes its values (waits, arithmetic; the
  // final value of value v replaces x in the load's own register), then stores all 16 in a burst, two
  // per slot from slot SB, and only then issues the loads for the next section, one value per slot
  // from slot LB -- no store of a section is younger than any of its loads.
  static constexpr int SB = E0 + 15 * EII + LVG + 1, LB = SB + 8;
  static constexpr int SUBS = PRO_ == 0 ? 6 : NT_ == 5 ? 12 : 9;          // slots per transform piece
  static constexpr int T0 = SECT - NP0 * SUBS;            // first transform slot (both sections)
  // piece-relative slots: raw read of the NEXT piece, operand-plane writes, d y cache, refill DMA
  // (the raw read sits as late as the LDS latency allows: its eight registers then reuse the ones the
  // split has just released)
  static constexpr int KR = SUBS == 12 ? 9 : SUBS == 9 ? 6 : 3, KWH = SUBS == 6 ? 3 : 6, KWL = SUBS == 12 ? 9 : KWH;
  static constexpr int KC = SUBS == 6 ? 5 : 7, KD0 = SUBS == 6 ? 4 : SUBS == 9 ? 7 : 8, KD1 = SUBS == 9 ? 8 : 10;
  static_assert(T0 % 3 == 0 && SUBS % 3 == 0, "LDS instructions in slot 0 of a step");
  static_assert(EEND < SECT && LB + 16 <= SECT, "epilogue, store burst and loads fit a section");
  // refill DMAs guaranteed in slots > f of section s (the tile's last piece exists on some waves only)
  static constexpr int dma_after(int s, int f) {
    int n = 0;
    const int np = s == 0 ? NP0 : NP1;
    for (int p = 0; p < np; ++p) {
      const int j = s == 0 ? p : NP0 + p;
      if (j == DPW - 1) continue;
      n += (T0 + p * SUBS + KD0 > f) ? 1 : 0;
      if (PRO_ == 2) n += (T0 + p * SUBS + KD1 > f) ? 1 : 0;
    }
    return n;
  }
  // counted wait of value v, consumed in the section behind section s: its loads were issued in slot
  // LB + v of section s; guaranteed younger operations = the later loads and the later refill DMAs of
  // section s -- loads all of them.  (The d y cache and statistics stores in between are conditional
  // and not counted: a pending one only makes the wait stricter.)
  static constexpr int nwait(int s, int v) { return (15 - v) * LPV + dma_after(s, LB + v); }
};

// LDS-DMA piece with the destination formed in the statement: M0 = base + immediate (s_add_u32 writes
// SCC).  M0 is not saved and restored around it: hipcc itself neither reads nor writes M0 anywhere in this
// kernel (gfx950 LDS instructions do not use it; tools/wsd_audit.py checks the ISA for a compiler "m0").
__device__ static inline void wsd_dma16i(const void* gbase, unsigned voff, unsigned lds_base, int imm) {
  asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               :: "v"(voff), "s"(gbase), "s"(lds_base), "n"(imm) : "memory", "scc");
}

// The stored-tensor values of the pending section (and the second gradient, EP 4) are loaded by asm a
// whole section ahead of their use, across the tile loop's back edge.  hipcc does not know that such a
// register is in flight: as a C++ variable it gets copied into the loop-carried register at the loop
// header while the load has not landed (found as intermittently stale values; "+v" ties and register
// variables do not prevent the copy).  So these values live in RESERVED registers: amdgpu_num_vgpr
// keeps hipcc's allocation below them, and every access -- load, the levels that read or replace the
// value, store, statistics -- is an asm statement that names the register literally: v240 + i for
// the stored tensor, a240 + i for the second gradient (amdgpu_num_vgpr(240) caps both files; a global
// load may target an accumulator register, v_accvgpr_read fetches the value once it has landed).  In
// the kernels without a second gradient a240..a255 hold two more (tap, k-step) pairs of weights,
// placed and named by hand.  tools/wsd_audit.py checks that no compiler-generated instruction touches
// the reserved registers.
#define WSD_XR0 240
template <int NT, int HALO, int PRO, int EP>
__device__ __forceinline__ void wsd_body(const SaConvArgs& a, int bcost, int total_tiles, unsigned long long xw) {
  typedef WsGeo<128, NT, HALO> G;
  typedef WsdSched<NT, HALO, PRO, EP> Sch;
  constexpr int WS_C = 128, WS_TM = 64, WS_KSTEPS = G::KSTEPS, WS_PITCH = G::PITCH, RPP = G::RPP, LPR = G::LPR;
  constexpr int WS_NTAPS = NT, WS_ROWS = G::ROWS, WS_PLANE = G::PLANE, WS_NDMA = G::NDMA, WS_DPW = G::DPW,
                WS_RAW_BYTES = G::RAW_BYTES, WS_BUF_BYTES = G::BUF_BYTES,
                // (tap, k-step) pairs whose fragments live in AGPRs: amdgpu_num_vgpr(240) caps the accumulator
                // file at 240 registers too (30 pairs); the last 16 (a240..a255) hold the second gradient (EP 4)
                WS_NAGPR_FRAGS = NT * G::KSTEPS < 30 ? NT * G::KSTEPS : 30,
                // ... or, without one, pairs 30 and 31 by hand: hi a[240:243] / lo a[244:247], hi a[248:251] / lo a[252:255]
                WS_NHAND = (EP != 4 && NT * G::KSTEPS > 30) ? 2 : 0;
  constexpr bool PRO2 = PRO == 2;
  constexpr bool ACT = EP == 1 || EP == 3 || EP == 4;       // the epilogue evaluates z = x*s1 + t1 and sigmoid(z)
  constexpr float NL2E = -1.4426950408889634f;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // LDS: [operand buffer 0][operand buffer 1][raw tile x][raw tile nb_x (PRO2)][column-sum scratch (PRO2)]
  bf16_t* const planes = reinterpret_cast<bf16_t*>(smem);
  unsigned char* const raw = smem + 2 * WS_BUF_BYTES;
  const unsigned raw_lds = (unsigned)(uintptr_t)(lds_byte*)raw;
  unsigned raw_lds_w = raw_lds + (unsigned)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * 1024;   // this wave's piece 0
  asm volatile("" : "+s"(raw_lds_w));
  const int tid = threadIdx.x, lane_ = tid & 63;
  const int wave_ = __builtin_amdgcn_readfirstlane(tid >> 6);
  // This workgroup's contiguous tile range.  Tiles are not equally expensive: an iteration whose tile
  // sits at the start of an utterance or within the last few tiles runs un-overlapped, about 2.25 times
  // as long (tools/wsd_stamps.py), and the launch ends with its slowest workgroup (at B = 10, a dozen
  // tiles per workgroup, a cost of the utterance END alone left lifetimes between 82 and 152 us).  The
  // ranges are cut at equal COST in quarter tiles: 4 per overlapped tile, `pq` per plain one, each tile
  // carrying its own cost (plain tiles: [0, nh) and [ntiles - nt, ntiles) of every utterance).
#ifdef SA_WS_OLD_RANGES
  int first, last;
  {                                                         // (A/B build only: round 2's ranges, one surcharge behind each utterance)
    const unsigned long long U = (unsigned long long)a.ntiles + 5u;
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
    // (the thresholds of `fast` below, from the launch geometry alone)
    const int e_lo = a.rowmin < 0 ? (-a.rowmin + WS_TM - 1) / WS_TM : 0;
    int e_hi = (a.Lin - WS_ROWS - a.rowmin) >> 6;
    if (e_hi > a.ntiles - 2) e_hi = a.ntiles - 2;
    int f_hi = e_hi - 2;
    const int p_hi = (a.Lout - WS_TM) >> 6;
    if (f_hi > p_hi) f_hi = p_hi;
    if (f_hi > a.ntiles - 3) f_hi = a.ntiles - 3;
    const int f_lo = e_lo > 2 ? e_lo - 1 : 1;
    int nh = f_lo, nt = a.ntiles - 1 - f_hi;
    if (f_hi < f_lo) { nh = a.ntiles; nt = 0; }             // (a short utterance: every tile is plain)
    const unsigned long long pq = (unsigned)(bcost & 0xffff), ni = (unsigned)(a.ntiles - nh - nt);
    const unsigned long long U = pq * (unsigned)(nh + nt) + 4 * ni;
    const unsigned long long ctot = (unsigned long long)(total_tiles / a.ntiles) * U;
    auto inv = [&](unsigned long long c) {                  // tiles wholly in front of cost position c
      const unsigned long long k = c / U, r = c - k * U;
      unsigned long long t;
      if (r < pq * (unsigned)nh) t = r / pq;
      else if (r < pq * (unsigned)nh + 4 * ni) t = (unsigned)nh + (r - pq * (unsigned)nh) / 4;
      else t = (unsigned)nh + ni + (r - pq * (unsigned)nh - 4 * ni) / pq;
      t += k * (unsigned)a.ntiles;
      return (int)(t < (unsigned)total_tiles ? t : (unsigned)total_tiles);
    };
    // (workgroup i runs on XCD i % 8, and the XCDs of one chip do not run this kernel at one speed: its share
    // of the cost follows the weight of its XCD -- sa_conv_ws_set_xcd_weights, measured per process)
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
  WSD_LIFE(0);
  WSD_STAMP(63, 0); WSD_STAMP_RT(63, 8); WSD_WG_STAMP(0);

  // ---- the weights: this wave's 32 output columns, all taps / channels, hi and lo images ----
  bf16x8 Bh[WS_NTAPS][WS_KSTEPS], Bl[WS_NTAPS][WS_KSTEPS];
  {
    const bf16x8* wp = reinterpret_cast<const bf16x8*>(a.wp);
#pragma unroll
    for (int t = 0; t < WS_NTAPS; ++t) {
      const bf16x8* wt = wp + ((size_t)a.taps.widx[0][t] * WS_KSTEPS * 4 + wave_) * 64 + lane_;
#pragma unroll
      for (int k = 0; k < WS_KSTEPS; ++k) {
        Bh[t][k] = wt[(size_t)k * 4 * 64];
        Bl[t][k] = wt[(size_t)a.wlo_off + (size_t)k * 4 * 64];
      }
      // one tap at a time, moved to its home before the next tap is fetched; the empty asm also makes
      // hipcc wait for the loads HERE and not in front of their first use inside the tile loop
#pragma unroll
      for (int k = 0; k < WS_KSTEPS; ++k) {
        const int pr = t * WS_KSTEPS + k;
        if (pr < WS_NAGPR_FRAGS) {
          asm volatile("" : "+a"(Bh[t][k]), "+a"(Bl[t][k]));
        } else if (pr < WS_NAGPR_FRAGS + WS_NHAND) {
          typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
          const u32x4 uh = __builtin_bit_cast(u32x4, Bh[t][k]), ul = __builtin_bit_cast(u32x4, Bl[t][k]);
          if (pr == 30)
            asm volatile("v_accvgpr_write_b32 a240, %0\n\tv_accvgpr_write_b32 a241, %1\n\tv_accvgpr_write_b32 a242, %2\n\tv_accvgpr_write_b32 a243, %3\n\t"
                         "v_accvgpr_write_b32 a244, %4\n\tv_accvgpr_write_b32 a245, %5\n\tv_accvgpr_write_b32 a246, %6\n\tv_accvgpr_write_b32 a247, %7"
                         :: "v"(uh[0]), "v"(uh[1]), "v"(uh[2]), "v"(uh[3]), "v"(ul[0]), "v"(ul[1]), "v"(ul[2]), "v"(ul[3]));
          else
            asm volatile("v_accvgpr_write_b32 a248, %0\n\tv_accvgpr_write_b32 a249, %1\n\tv_accvgpr_write_b32 a250, %2\n\tv_accvgpr_write_b32 a251, %3\n\t"
                         "v_accvgpr_write_b32 a252, %4\n\tv_accvgpr_write_b32 a253, %5\n\tv_accvgpr_write_b32 a254, %6\n\tv_accvgpr_write_b32 a255, %7"
                         :: "v"(uh[0]), "v"(uh[1]), "v"(uh[2]), "v"(uh[3]), "v"(ul[0]), "v"(ul[1]), "v"(ul[2]), "v"(ul[3]));
        } else {
          asm volatile("" : "+v"(Bh[t][k]), "+v"(Bl[t][k]));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  WSD_STAMP(63, 1);
  // (data gradients carry no bias: wsd_variant() sends a launch with one to the one-tile kernel)
  constexpr bool MASK = PRO2 && EP != 1;                    // ReLU mask of the BatchNorm blocks' prologue (wsd_variant() checks)
  const bool has_stats = a.stats != nullptr, has_ao = a.a_out != nullptr;

  // ---- lane constants of the filler slots (opaque: hipcc keeps THESE and derives per-slot addresses
  // from them by immediates instead of hoisting one address chain per slot) ----
  const int half_ = lane_ >> 5, l31_ = lane_ & 31;             // accumulator layout: rows +4, column
  const int rowp_ = lane_ / LPR, cq_ = lane_ % LPR;            // transform layout: row of the piece, channel quad
  unsigned row0 = RPP * wave_ + rowp_;
  unsigned raw_off = wave_ * 1024 + lane_ * 16;            // byte offset in the raw tile and in the row block of a DMA tile: + j*4096
  unsigned pl_off = (row0 * WS_PITCH + cq_ * 4) * 2;       // byte offset in an operand plane: + j*4*RPP*PITCH*2
  unsigned ao_off = (row0 * WS_C + cq_ * 4) * 2;           // byte offset in the a_out row block: + j*4*RPP*C*2
  unsigned y_off = ((4 * half_) * WS_C + wave_ * 32 + l31_) * 4;   // byte offset in a [64][128] fp32 row block: + ro*512
  asm volatile("" : "+v"(row0), "+v"(raw_off), "+v"(pl_off), "+v"(ao_off), "+v"(y_off));

#define WS_IDS int lane = lane_, wave = wave_; asm volatile("" : "+v"(lane), "+s"(wave)); (void)wave; (void)lane

  // a tile = (utterance, tile of the utterance) + its first staged input row and first output row in the
  // [B*Lin] / [B*Lout] row spaces, advanced incrementally (the per-tile bookkeeping is scalar code in
  // front of the tile body: nothing overlaps it)
  struct Tile { int b, tile, irow, orow; };
  auto tile_of = [&](int t) {
    Tile r; r.b = t / a.ntiles; r.tile = t - r.b * a.ntiles;
    r.irow = r.b * a.Lin + r.tile * WS_TM + a.rowmin; r.orow = r.b * a.Lout + r.tile * WS_TM;
    return r;
  };
  const int iwrap = a.Lin - (a.ntiles - 1) * WS_TM, owrap = a.Lout - (a.ntiles - 1) * WS_TM;
  auto next_tile = [&](Tile T) {
    Tile r; const bool wrap = T.tile + 1 == a.ntiles;
    r.b = wrap ? T.b + 1 : T.b; r.tile = wrap ? 0 : T.tile + 1;
    r.irow = T.irow + (wrap ? iwrap : WS_TM); r.orow = T.orow + (wrap ? owrap : WS_TM);
    return r;
  };
  // rows outside the utterance among the staged ones, or the trailing rows it owns beyond its 64:
  // tile*64 + rowmin < 0, tile*64 + rowmin + ROWS > Lin, or the last tile -- two thresholds on the tile index
  const int edge_lo = a.rowmin < 0 ? (-a.rowmin + WS_TM - 1) / WS_TM : 0;
  int edge_hi = (a.Lin - WS_ROWS - a.rowmin) >> 6;          // (floor)
  if (edge_hi > a.ntiles - 2) edge_hi = a.ntiles - 2;
  const int part_hi = (a.Lout - WS_TM) >> 6;
  auto is_edge = [&](Tile T) { return T.tile < edge_lo || T.tile > edge_hi; };
  auto is_partial = [&](Tile T) { return T.tile > part_hi; };
  // an iteration is overlapped when the previous, the current and the next two tiles are interior tiles
  // of ONE utterance: tile c-1 .. c+2 with c in [fast_lo, fast_hi]
  const int fast_lo = edge_lo > 2 ? edge_lo - 1 : 1;
  int fast_hi = edge_hi - 2;
  if (fast_hi > part_hi) fast_hi = part_hi;
  if (fast_hi > a.ntiles - 3) fast_hi = a.ntiles - 3;

  // ---- constants: prologue coefficients per (utterance | -, channel quad); epilogue per column ----
  float k1[4], k2[4], k3[4];
  int cur_b = -1;
  auto load_consts = [&](int b) {
    if constexpr (PRO2) {
      WS_IDS;
      const int ch = (lane % LPR) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const size_t q = (size_t)b * a.nb_bstride + ch + j;
        k1[j] = a.nb_c1[q]; k2[j] = a.nb_c2[q]; k3[j] = a.nb_c3[q];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(k1[j]), "+v"(k2[j]), "+v"(k3[j]));   // waited for here
    }
    cur_b = b;
  };
  float es1 = 1.0f, et1 = 0.0f, emu = 0.0f, ers = 1.0f, gk1 = 1.0f, gk2 = 0.0f, gk3 = 0.0f;
  int cur_eb = -1;
  auto load_ep_consts = [&](int b) {
    WS_IDS;
    const int col = wave * 32 + (lane & 31);
    if constexpr (ACT) { es1 = a.ep_s1[(size_t)b * WS_C + col]; et1 = a.ep_t1[(size_t)b * WS_C + col]; }
    emu = a.ep_mean[(size_t)b * a.ep_bstride + col];
    ers = a.ep_rstd[(size_t)b * a.ep_bstride + col];
    if constexpr (EP == 4) { gk1 = a.ep_g2k1[col]; gk2 = a.ep_g2k2[col]; gk3 = a.ep_g2k3[col]; }
    asm volatile("" : "+v"(es1), "+v"(et1), "+v"(emu), "+v"(ers), "+v"(gk1), "+v"(gk2), "+v"(gk3));
    cur_eb = b;
  };

  // ---- uniform per-tile bases (set once per iteration, opaque) ----
  const char* xbase_d = nullptr; const char* x2base_d = nullptr;    // rows of the DMA tile (x, nb_x)
  char* aobase_t = nullptr;                                          // a_out rows of the transform tile
  auto row_ptr = [&](const void* p, int row, int row_bytes) {                  // (B*L rows < 2^31: checked at launch)
    const char* r = reinterpret_cast<const char*>(p) + (long)row * row_bytes;
    asm volatile("" : "+s"(r));
    return r;
  };
  // a [64][128] fp32 row block addressed by 13-bit immediates: value (h, i) of a tile lives in row
  // 32h + 8(i >> 2) + (i & 3) (+ 4 for the upper lane half): pointer = row 16q + 8 of the block with
  // q = 2h + (i >> 3), immediate ((i >> 2) & 1 ? 0 : -4096) + (i & 3) * 512.  The slots walk the values in
  // order, so each stream (y stores, ep_x loads, ep_g2 loads) keeps ONE running pointer, advanced by 16
  // rows after every eighth value.
  auto block8 = [&](const void* p, Tile T) {               // row 8 of the tile's output row block
    const char* r = reinterpret_cast<const char*>(p) + ((long)T.orow * WS_C + 8 * WS_C) * 4;
    asm volatile("" : "+s"(r));
    return r;
  };
  const char* yb_p = nullptr; const char* yb_c = nullptr; const char* xb_c = nullptr; const char* gb_c = nullptr;
  const char* ycur = nullptr; const char* xcur = nullptr; const char* gcur = nullptr;
#define WSD_IMM(i) ((((i) >> 2) & 1 ? 0 : -4096) + ((i) & 3) * 512)
#define WSD_Q(h, i) (2 * (h) + ((i) >> 3))
#define WSD_STEP16(p) do { p += 16 * WS_C * 4; asm volatile("" : "+s"(p)); } while (0)

  // ---- LDS-DMA of piece j of this wave (rows RPP*i .., i = wave + 4j) of tile T ----
  auto dma_piece = [&](Tile T, bool edge, int j, int part = 2) {
    const int i = wave_ + 4 * j;
    if (j == WS_DPW - 1 && i >= WS_NDMA) return;
    if (!edge) {
      // (LDS destination = this wave's first piece + an immediate, formed in the statement: no scalar
      // register per piece)
      if (part != 1) wsd_dma16i(xbase_d + j * 4096, raw_off, raw_lds_w, j * 4096);
      if constexpr (PRO2) {
        if (part != 0) wsd_dma16i(x2base_d + j * 4096, raw_off, raw_lds_w, WS_RAW_BYTES + j * 4096);
      }
    } else {
      WS_IDS;
      int g = T.tile * WS_TM + a.rowmin + RPP * i + lane / LPR;
      g = g < 0 ? 0 : (g >= a.Lin ? a.Lin - 1 : g);        // rows outside the utterance: any valid address (zeroed in the transform)
      const size_t off = ((size_t)T.b * a.Lin + g) * (WS_C * 4) + (lane % LPR) * 16;
      ws_dma16(reinterpret_cast<const char*>(a.x) + off, raw_lds + i * 1024);
      if constexpr (PRO2) ws_dma16(reinterpret_cast<const char*>(a.nb_x) + off, raw_lds + WS_RAW_BYTES + i * 1024);
    }
  };

  // ---- transform of a piece, in sub-steps (each small enough for one filler slot) ----
  f32x4 vx[1], vy[1];
  float f[4], csum[4], z[4], w[4];
  uint2 phi, plo;
  unsigned pl_cur = 0;                                     // LDS byte offset of this lane's first row in the transform tile's buffer
  auto has_piece = [&](int j) { return !(j == WS_DPW - 1 && wave_ + 4 * j >= WS_NDMA); };
  auto piece_read = [&](int j) {
    if (!has_piece(j)) return;
    vx[0] = *reinterpret_cast<const f32x4*>(raw + (raw_off + j * 4096));
    if constexpr (PRO2) vy[0] = *reinterpret_cast<const f32x4*>(raw + (raw_off + WS_RAW_BYTES + j * 4096));
  };
  auto own_range = [&](Tile T, int& lo, int& hi) {
    lo = T.tile * WS_TM;
    hi = T.tile == a.ntiles - 1 ? a.Lin : lo + WS_TM;
    if (hi > a.Lin) hi = a.Lin;
  };
  auto owns = [&](Tile T, bool edge, int j) -> bool {
    if (!edge) {                                           // own rows = staged rows [-rowmin, -rowmin + TM)
      if (j > 0 && j < WS_DPW - 2) return true;            // the middle pieces are owned in every geometry
      const int r = (int)row0 + 4 * RPP * j;
      return r >= -a.rowmin && r < -a.rowmin + WS_TM;
    }
    int lo, hi;
    own_range(T, lo, hi);
    const int g = T.tile * WS_TM + a.rowmin + (int)row0 + 4 * RPP * j;
    return g >= lo && g < hi;
  };
  // hi = bf16(v), lo = bf16(v - hi) of a channel pair (same roundings as sa_split4; opaque inputs: the
  // split must see the ROUNDED value, not an fma that formed it)
  auto split_pair = [&](float a0, float a1, unsigned& hi, unsigned& lo) {
    asm volatile("" : "+v"(a0), "+v"(a1));
    hi = ws_cvt_pk_bf16(a0, a1);
    lo = ws_cvt_pk_bf16(a0 - __uint_as_float(hi << 16), a1 - __uint_as_float(hi & 0xffff0000u));
  };
  auto piece_elem = [&](Tile T, bool edge, int j, int q) {            // plain path: channel q of the lane's four
    if (!has_piece(j)) return;
    float v = vx[0][q];
    if constexpr (PRO2) {
      const float y = vy[0][q];
      v = fmaf(k1[q], v, fmaf(k2[q], y, k3[q]));
      if constexpr (MASK) v = y > 0.0f ? v : 0.0f;
      if (owns(T, edge, j)) csum[q] += v;
    }
    if (edge) {                                            // rows outside the utterance are zero operands
      const int g = T.tile * WS_TM + a.rowmin + (int)row0 + 4 * RPP * j;
      if (!(g >= 0 && g < a.Lin)) v = 0.0f;
    }
    f[q] = v;
    if (q == 1) split_pair(f[0], f[1], phi.x, plo.x);
    if (q == 3) split_pair(f[2], f[3], phi.y, plo.y);
  };
  // the same transform for an interior tile, cut by dependence level (one level of all four channels per slot)
  auto piece_level = [&](Tile T, int j, int lv) {
    if (!has_piece(j)) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if constexpr (!PRO2) {
        if (lv == 0) f[q] = vx[0][q];
      } else {
        if (lv == 0) z[q] = fmaf(k2[q], vy[0][q], k3[q]);
        if (lv == 1) f[q] = fmaf(k1[q], vx[0][q], z[q]);
        if (lv == 2) { if constexpr (MASK) f[q] = vy[0][q] > 0.0f ? f[q] : 0.0f; }
        // (asm: hipcc otherwise sinks the sums of all pieces into the slot that folds them)
        if (lv == 3) { if (owns(T, false, j)) asm volatile("v_add_f32 %0, %0, %1" : "+v"(csum[q]) : "v"(f[q])); }
      }
    }
  };
  auto piece_split = [&](int j, int lv) {
    if (!has_piece(j)) return;
    if (lv == 0) {
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
  auto piece_write = [&](int j, int part = 2) {             // operand planes of the transform tile (0 hi, 1 lo, 2 both)
    if (!has_piece(j)) return;
    unsigned char* dst = smem + (pl_cur + j * (4 * RPP * WS_PITCH * 2));
    if (part != 1) *reinterpret_cast<uint2*>(dst) = phi;
    if (part != 0) *reinterpret_cast<uint2*>(dst + WS_PLANE * 2) = plo;
  };
  auto piece_cache = [&](Tile T, bool edge, int j) {          // bf16 d y for sa_wgrad: hi values of the owned rows
    if (!has_piece(j)) return;
    if (has_ao && owns(T, edge, j)) ws_store_b64(aobase_t + j * (4 * RPP * WS_C * 2), ao_off, phi);
  };
  // per-tile column sums of d y (bias gradient): fold the two row halves of the wave, one LDS slot per
  // wave, summed in wave order by 128 threads after the tile barrier
  auto colsum_put = [&](int it) {
    if constexpr (PRO2) {
      if (a.nb_colsum) {
        float* colred = reinterpret_cast<float*>(raw + 2 * WS_RAW_BYTES) + (size_t)(it & 1) * 4 * WS_C;
        f32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {                      // lanes l and l + 32 hold the same channel quad
          const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(csum[q]), __float_as_uint(csum[q]), false, false);
          v[q] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        }
        if (lane_ < 32) *reinterpret_cast<f32x4*>(colred + wave_ * WS_C + (lane_ & 31) * 4) = v;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) csum[q] = 0.0f;
    }
  };
  // (in two halves: the LDS reads, and -- a slot or more later in an overlapped body -- the sum and store)
  float cso[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  auto colsum_rd = [&](int it) {                            // after the barrier that follows colsum_put(it)
    if constexpr (PRO2) {
      if (a.nb_colsum && tid < WS_C) {
        const float* colred = reinterpret_cast<const float*>(raw + 2 * WS_RAW_BYTES) + (size_t)(it & 1) * 4 * WS_C;
#pragma unroll
        for (int w_ = 0; w_ < 4; ++w_) cso[w_] = colred[w_ * WS_C + tid];
      }
    }
  };
  auto colsum_wr = [&](int t) {
    if constexpr (PRO2) {
      if (a.nb_colsum && tid < WS_C) a.nb_colsum[(size_t)t * WS_C + tid] = (cso[0] + cso[1]) + (cso[2] + cso[3]);
    }
  };
  auto colsum_out = [&](int t, int it) { colsum_rd(it); colsum_wr(t); };

  // ---- epilogue: one accumulator register = one value (row ro + 4*half, this lane's column) ----
  f32x16 acc[2];
  asm volatile("" ::: "v255", "a255");                    // (the kernel descriptor must cover the reserved registers)
  float ez[8], ew[8], es[8], et[8], eg[8], exn[8], ea[8];   // per-value pipeline registers (value v uses v & 7)
  float ssum = 0.0f, ssq = 0.0f;
  // one dependence level of value (h, i); xq / gq / yq: the row-block bases of its tile.  The very
  // operations of sa_conv_gemm's ep_rows, in its order (sa_swish / sa_swish_grad spelled out).
  auto epi_level = [&](auto h_c, auto i_c, auto lv_c) {
    constexpr int h = decltype(h_c)::value, i = decltype(i_c)::value, lv = decltype(lv_c)::value, r = i & 7;
    (void)&ssum; (void)&ssq; (void)&y_off;
  // (asm: hipcc otherwise sinks all the accumulations of a section, and the xhat arithmetic that feeds
  // them, into the statistics slot behind its branch)
#define WSD_ACC_SUM() asm volatile("v_add_f32 %0, %0, v%c1" : "+v"(ssum) : "n"(WSD_XR0 + i))
#define WSD_ACC_SQ() asm volatile("v_fmac_f32 %0, v%c2, %1" : "+v"(ssq) : "v"(exn[r]), "n"(WSD_XR0 + i))
  // x = the reserved register of value i: z = x*s1 + t1 | x - mean | x <- a*b | x <- a + b | ea += gk1*g2
#define WSD_X_FMA(dst, m, c) asm volatile("v_fma_f32 %0, v%c3, %1, %2" : "=v"(dst) : "v"(m), "v"(c), "n"(WSD_XR0 + i))
#define WSD_X_SUB(dst, c) asm volatile("v_sub_f32 %0, v%c2, %1" : "=v"(dst) : "v"(c), "n"(WSD_XR0 + i))
#define WSD_X_SETMUL(a_, b_) asm volatile("v_mul_f32 v%c2, %0, %1" :: "v"(a_), "v"(b_), "n"(WSD_XR0 + i))
#define WSD_X_SET(a_) asm volatile("v_mov_b32 v%c1, %0" :: "v"(a_), "n"(WSD_XR0 + i))
#define WSD_G2_READ(dst) asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(dst) : "n"(WSD_XR0 + i))
    // the last level that needs x writes the final value over it (in its reserved register): the store burst and the
    // statistics read it there
    if constexpr (EP == 1) {
      if constexpr (lv == 0) WSD_X_FMA(ez[r], es1, et1);
      if constexpr (lv == 1) ew[r] = ez[r] * NL2E;
      if constexpr (lv == 2) ew[r] = __builtin_amdgcn_exp2f(ew[r]);
      if constexpr (lv == 3) ew[r] = 1.0f + ew[r];
      if constexpr (lv == 4) es[r] = __builtin_amdgcn_rcpf(ew[r]);
      if constexpr (lv == 5) { et[r] = 1.0f - es[r]; eg[r] = acc[h][i]; }
      if constexpr (lv == 6) et[r] = fmaf(ez[r], et[r], 1.0f);
      if constexpr (lv == 7) es[r] = es[r] * et[r];
      if constexpr (lv == 8) WSD_X_SUB(exn[r], emu);
      if constexpr (lv == 9) { WSD_X_SETMUL(eg[r], es[r]); exn[r] = exn[r] * ers; }
      if constexpr (lv == 10) { WSD_ACC_SUM(); WSD_ACC_SQ(); }
    } else if constexpr (EP == 4) {
      if constexpr (lv == 0) WSD_X_FMA(ez[r], es1, et1);
      if constexpr (lv == 1) ew[r] = ez[r] * NL2E;
      if constexpr (lv == 2) ew[r] = __builtin_amdgcn_exp2f(ew[r]);
      if constexpr (lv == 3) ew[r] = 1.0f + ew[r];
      if constexpr (lv == 4) { es[r] = __builtin_amdgcn_rcpf(ew[r]); WSD_G2_READ(exn[r]); }
      if constexpr (lv == 5) { et[r] = 1.0f - es[r]; ea[r] = ez[r] * es[r]; }
      if constexpr (lv == 6) { et[r] = fmaf(ez[r], et[r], 1.0f); ea[r] = fmaf(gk2, ea[r], gk3); }
      if constexpr (lv == 7) { es[r] = es[r] * et[r]; ea[r] = fmaf(gk1, exn[r], ea[r]); }
      if constexpr (lv == 8) eg[r] = acc[h][i];
      if constexpr (lv == 9) eg[r] = eg[r] + ea[r];
      if constexpr (lv == 10) WSD_X_SUB(exn[r], emu);
      if constexpr (lv == 11) { WSD_X_SETMUL(eg[r], es[r]); exn[r] = exn[r] * ers; }
      if constexpr (lv == 12) { WSD_ACC_SUM(); WSD_ACC_SQ(); }
    } else if constexpr (EP == 3) {
      if constexpr (lv == 0) WSD_X_FMA(ez[r], es1, et1);
      if constexpr (lv == 1) ew[r] = ez[r] * NL2E;
      if constexpr (lv == 2) ew[r] = __builtin_amdgcn_exp2f(ew[r]);
      if constexpr (lv == 3) ew[r] = 1.0f + ew[r];
      if constexpr (lv == 4) es[r] = __builtin_amdgcn_rcpf(ew[r]);
      if constexpr (lv == 5) { ea[r] = ez[r] * es[r]; WSD_X_SET(acc[h][i]); }
      if constexpr (lv == 6) exn[r] = ea[r] - emu;
      if constexpr (lv == 7) { exn[r] = exn[r] * ers; WSD_ACC_SUM(); }
      if constexpr (lv == 8) WSD_ACC_SQ();
    } else {
      if constexpr (lv == 0) WSD_X_SUB(exn[r], emu);
      if constexpr (lv == 1) { WSD_X_SET(acc[h][i]); exn[r] = exn[r] * ers; }
      if constexpr (lv == 2) { WSD_ACC_SUM(); WSD_ACC_SQ(); }
    }
#undef WSD_X_FMA
#undef WSD_X_SUB
#undef WSD_X_SETMUL
#undef WSD_X_SET
#undef WSD_G2_READ
#undef WSD_ACC_SUM
#undef WSD_ACC_SQ
  };
  // store of value i (its final value sits in its reserved register); yp: the stream pointer of its row group
  auto epi_store = [&](auto i_c, const char* yp) {
    constexpr int i = decltype(i_c)::value;
    (void)&y_off;
    asm volatile("global_store_dword %0, v%c3, %1 offset:%2" :: "v"(y_off), "s"(yp), "n"(WSD_IMM(i)), "n"(WSD_XR0 + i) : "memory");
  };
  // loads of value i of a section (xp / gp: the stream pointers of its row group) into the reserved registers
  auto epi_load = [&](auto i_c, const char* xp, const char* gp) {
    constexpr int i = decltype(i_c)::value;
    (void)&ssum; (void)&ssq; (void)&y_off;
    asm volatile("global_load_dword v%c3, %0, %1 offset:%2" :: "v"(y_off), "s"(xp), "n"(WSD_IMM(i)), "n"(WSD_XR0 + i) : "memory");
    if constexpr (EP == 4)
      asm volatile("global_load_dword a%c3, %0, %1 offset:%2" :: "v"(y_off), "s"(gp), "n"(WSD_IMM(i)), "n"(WSD_XR0 + i) : "memory");
  };
  // whole section h of tile T outside the slots (plain loads, bounds-checked; everything older has landed)
  auto epi_plain = [&](auto h_c, Tile T, const char* yb, const char* xb, const char* gb) {     // (row 8 of the row blocks)
    constexpr int h = decltype(h_c)::value;
    if (cur_eb != T.b) load_ep_consts(T.b);
    auto in_bounds = [&](int i) {
      WS_IDS;
      return T.tile * WS_TM + 32 * h + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5) < a.Lout;
    };
    ws_static_for<0, 16>([&](auto i_c) {                    // all loads first, one wait
      constexpr int i = decltype(i_c)::value;
      if (in_bounds(i)) epi_load(i_c, xb + WSD_Q(h, i) * (16 * WS_C * 4), EP == 4 ? gb + WSD_Q(h, i) * (16 * WS_C * 4) : nullptr);
    });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ws_static_for<0, 16>([&](auto i_c) {
      constexpr int i = decltype(i_c)::value;
      (void)&ssum; (void)&ssq; (void)&y_off;
      if (in_bounds(i)) {
        ws_static_for<0, Sch::ELV>([&](auto lv_c) { epi_level(h_c, i_c, lv_c); });
        epi_store(i_c, yb + WSD_Q(h, i) * (16 * WS_C * 4));
      }
    });
  };
  // statistics of a tile: fold the two lane halves (a lane owns one column, the other half holds the
  // rows +4) and store; both halves then hold the same sums and write the same slab entry
  char* stbase_p = nullptr;
  auto epi_stats = [&]() {
    if (has_stats) {
      const auto rs = __builtin_amdgcn_permlane32_swap(__float_as_uint(ssum), __float_as_uint(ssum), false, false);
      const auto rq = __builtin_amdgcn_permlane32_swap(__float_as_uint(ssq), __float_as_uint(ssq), false, false);
      const float st_s = __uint_as_float(rs[0]) + __uint_as_float(rs[1]);
      const float st_q = __uint_as_float(rq[0]) + __uint_as_float(rq[1]);
      WS_IDS;
      ws_store_b64(stbase_p, (unsigned)((wave * 32 + (lane & 31)) * 8), make_uint2(__float_as_uint(st_s), __float_as_uint(st_q)));
    }
    ssum = 0.0f; ssq = 0.0f;
  };
  auto drain_vmem = [&]() {                                 // every asm load has landed in its register
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  int toff[WS_NTAPS];
#pragma unroll
  for (int t = 0; t < WS_NTAPS; ++t) toff[t] = (a.taps.off[0][t] - a.rowmin) * WS_PITCH;

  // ================= prologue: first tile staged without overlap =================
  Tile Tc = tile_of(first), Tn = tile_of(first + 1 < last ? first + 1 : last - 1), Tp = Tc;
#pragma unroll
  for (int q = 0; q < 4; ++q) csum[q] = 0.0f;
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
    load_consts(Tc.b);
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
  // iteration (t, it): section 0 = MFMAs of rows 0..31 of tile t -> acc[0], in its slots the epilogue of
  // rows 32..63 of tile t-1 (acc[1]), the loads for the epilogue of rows 0..31 of tile t, the transform
  // of pieces 0..NP0-1 of tile t+1; section 1 = rows 32..63 -> acc[1], epilogue of rows 0..31 of tile t,
  // loads for rows 32..63, pieces NP0.. of tile t+1.  One barrier per tile.
  WSD_STAMP(63, 2);
  yb_c = block8(a.y, Tc);
  bool cs_pending = false;
  bool ptr_step = false;                                   // the per-tile pointers can be advanced by their strides
  bool pend_issued = false;                                // the loads of the pending section (rows 32..63 of Tp) are in flight / in xr
  asm volatile("" : "=v"(acc[1]));                         // (read by the dummy pending section of a range's first tile)
  for (int t = first, it = 0; t < last; ++t, ++it) {
    int lanem = lane_;
    asm volatile("" : "+v"(lanem));
    const bf16_t* ab = planes + (size_t)(it & 1) * 2 * WS_PLANE + (lanem & 31) * WS_PITCH + (lanem >> 5) * 8;
    bf16x8 ah[2], al;
    auto a_ptr = [&](int S) {                               // step S of the tile: section S / NS, (tap, k-step) S % NS
      const int h = S / Sch::NS, s = S % Sch::NS;
      return ab + toff[s / WS_KSTEPS] + h * 32 * WS_PITCH + (s % WS_KSTEPS) * 16;
    };
    auto load_a = [&](int S) {
      al = *reinterpret_cast<const bf16x8*>(a_ptr(S) + WS_PLANE);
      ah[S & 1] = *reinterpret_cast<const bf16x8*>(a_ptr(S));
    };
    // column sums of the tile transformed by the previous iteration (= this one's tile): summed over the
    // waves and stored by 128 threads -- in the first slots of an overlapped body, else here
    WSD_STAMP(it, 8);
    load_a(0);
    __builtin_amdgcn_sched_barrier(0);
    const bool doE = t > first;
    // The first iteration of a range is overlapped too: there is no previous tile, so the slots of the
    // pending section (rows 32..63 of tile t-1) run on an undefined accumulator with Tp = Tc -- they store
    // into rows 32..63 and the statistics slab of THIS tile, which the next iteration's real pending
    // section overwrites (same wave, same addresses, stores retire in order); its loads read this tile's rows.
    const bool fast = (doE || !(bcost & 0x10000)) && Tc.tile >= fast_lo && Tc.tile <= fast_hi;
    const Tile Tnn = t + 2 < last ? next_tile(Tn) : Tn;       // (clamped: Tn is already the last tile then)
    const bool edgeT = is_edge(Tn), edgeD = is_edge(Tnn);   // (plain iterations only)
    WSD_STAMP(it, 9);
    if (Tn.b != cur_b) load_consts(Tn.b);
    yb_p = yb_c;                                           // (the previous iteration's current tile)
    if (fast && ptr_step) {
      // every tile involved is one tile further along its utterance: constant strides, no multiplies
      xbase_d += WS_TM * WS_C * 4; asm volatile("" : "+s"(xbase_d));
      if constexpr (PRO2) { x2base_d += WS_TM * WS_C * 4; asm volatile("" : "+s"(x2base_d)); }
      aobase_t += WS_TM * WS_C * 2; asm volatile("" : "+s"(aobase_t));
      yb_c += WS_TM * WS_C * 4; asm volatile("" : "+s"(yb_c));
      xb_c += WS_TM * WS_C * 4; asm volatile("" : "+s"(xb_c));
      if constexpr (EP == 4) { gb_c += WS_TM * WS_C * 4; asm volatile("" : "+s"(gb_c)); }
      stbase_p += WS_C * 8; asm volatile("" : "+s"(stbase_p));
    } else {
      xbase_d = row_ptr(a.x, Tnn.irow, WS_C * 4);
      if constexpr (PRO2) x2base_d = row_ptr(a.nb_x, Tnn.irow, WS_C * 4);
      aobase_t = const_cast<char*>(row_ptr(a.a_out, Tn.irow, WS_C * 2));
      yb_c = block8(a.y, Tc);
      xb_c = block8(a.ep_x, Tc);
      if constexpr (EP == 4) gb_c = block8(a.ep_g2, Tc);
      stbase_p = const_cast<char*>(row_ptr(a.stats, __builtin_amdgcn_readfirstlane(t - (int)doE), WS_C * 8));     // tile t-1 of the launch (first iteration: see `fast`)
    }
    // (the strides hold from a fast iteration to the next fast one unless the tile after next was clamped)
    ptr_step = fast && doE && t + 3 < last;                // (the first iteration's statistics pointer is the dummy's: recomputed next time)
    pl_cur = (it + 1) & 1 ? pl_off + WS_BUF_BYTES : pl_off;
    asm volatile("" : "+v"(pl_cur));
    WSD_STAMP(it, 10);
    if (cs_pending && !fast) colsum_out(t, it);
    if (doE && !fast) {                                     // pending section of the previous tile, not overlapped
      drain_vmem();
      asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[1]));
      epi_plain(std::integral_constant<int, 1>{}, Tp, yb_p, block8(a.ep_x, Tp), EP == 4 ? block8(a.ep_g2, Tp) : nullptr);
      epi_stats();
      pend_issued = false;
    }
    if (fast) {
      if (cur_eb != Tc.b) load_ep_consts(Tc.b);
      if (!pend_issued) {                                   // first overlapped tile behind a plain one
        xcur = block8(a.ep_x, Tp) + 2 * (16 * WS_C * 4);
        if constexpr (EP == 4) gcur = block8(a.ep_g2, Tp) + 2 * (16 * WS_C * 4);
        ws_static_for<0, 16>([&](auto i_c) {
          epi_load(i_c, xcur, gcur);
          if constexpr (decltype(i_c)::value == 7) { WSD_STEP16(xcur); if constexpr (EP == 4) WSD_STEP16(gcur); }
        });
        drain_vmem();
      }
    }
    WSD_STAMP(it, 0);
    WSD_STAMP_RT(it, 5);
    // One tile: 2 x SECT single-MFMA asm statements with filler slot f behind statement f of a section.
    auto tile_body = [&](auto fast_c) {
      constexpr bool FAST = decltype(fast_c)::value;
      auto filler = [&](auto sec_c, auto f_c) {
        constexpr int sec = decltype(sec_c)::value, fs = decltype(f_c)::value;
        if constexpr (!FAST) return;
        if constexpr (sec == 0 && fs == 0) { if (cs_pending) colsum_rd(it); }
        if constexpr (sec == 0 && fs == 3) { if (cs_pending) colsum_wr(t); }
        constexpr int he = sec ^ 1;                         // the section whose accumulator the epilogue reads
        // running pointers of the three streams: set in the slot in front of a stream's first value,
        // advanced behind its eighth
        if constexpr (fs == Sch::SB - 1) { ycur = sec == 0 ? yb_p + 2 * (16 * WS_C * 4) : yb_c; asm volatile("" : "+s"(ycur)); }
        if constexpr (fs == Sch::LB - 1) {
          if constexpr (sec == 0) { xcur = xb_c; asm volatile("" : "+s"(xcur)); if constexpr (EP == 4) { gcur = gb_c; asm volatile("" : "+s"(gcur)); } }
          else { WSD_STEP16(xcur); if constexpr (EP == 4) WSD_STEP16(gcur); }
        }
        // ---- epilogue stream: value v enters at slot E0 + v*EII, one level per slot ----
        if constexpr (fs >= Sch::E0 && fs < Sch::EEND && !(SA_WSD_ABL & 1)) {
          ws_static_for<0, 16>([&](auto v_c) {
            constexpr int v = decltype(v_c)::value, lv = fs - (Sch::E0 + v * Sch::EII);
            (void)&ssum; (void)&ssq; (void)&y_off;
            if constexpr (lv >= 0 && lv < Sch::ELV) {
              if constexpr (lv == 0) {
                // the previous section issued this value's load(s) in its slot LB + v
                // (the comment names the registers for tools/wsd_audit.py)
                constexpr int N = Sch::nwait(sec ^ 1, v);
                if constexpr (!(SA_WSD_ABL & (8 | 16)))
                asm volatile("s_waitcnt vmcnt(%0) ; landed v%c1" :: "n"(N), "n"(WSD_XR0 + v) : "memory");
              }
              epi_level(std::integral_constant<int, he>{}, v_c, std::integral_constant<int, lv>{});
            }
          });
        }
        // ---- store burst: values 2k, 2k+1 in slot SB + k ----
        if constexpr (fs >= Sch::SB && fs < Sch::SB + 8 && !(SA_WSD_ABL & (1 | 32))) {
          constexpr int k2 = 2 * (fs - Sch::SB);
          epi_store(std::integral_constant<int, k2>{}, ycur);
          epi_store(std::integral_constant<int, k2 + 1>{}, ycur);
          if constexpr (k2 == 6) WSD_STEP16(ycur);
        }
        if constexpr (sec == 0 && fs == Sch::EEND && !(SA_WSD_ABL & 1)) epi_stats();
        if constexpr (sec == 1 && fs == 0) WSD_STAMP(it, 1);
        if constexpr (fs == Sch::SB) WSD_STAMP(it, 6 + sec);
        // ---- transform stream ----
        constexpr int np = sec == 0 ? Sch::NP0 : Sch::NP1, jb = sec == 0 ? 0 : Sch::NP0;
        if constexpr (fs == Sch::T0 - 3 && !(SA_WSD_ABL & 2)) piece_read(jb);
        if constexpr (fs >= Sch::T0 && fs < Sch::T0 + np * Sch::SUBS && !(SA_WSD_ABL & 2)) {
          constexpr int p = (fs - Sch::T0) / Sch::SUBS, k = (fs - Sch::T0) % Sch::SUBS, j = jb + p;
          if constexpr (PRO2) {
            if constexpr (k < 3) piece_level(Tn, j, k);
            if constexpr (k == 3) piece_split(j, 0);
            if constexpr (k == 4) { piece_split(j, 1); piece_level(Tn, j, 3); }
            if constexpr (k == 5) piece_split(j, 2);
            if constexpr (k == Sch::KWH) { piece_write(j, 0); piece_split(j, 3); }
          } else {
            if constexpr (k == 0) { piece_level(Tn, j, 0); piece_split(j, 0); }
            if constexpr (k == 1) piece_split(j, 1);
            if constexpr (k == 2) piece_split(j, 2);
            if constexpr (k == Sch::KWH) { piece_write(j, 0); piece_split(j, 3); }
          }
          if constexpr (k == Sch::KR && p + 1 < np) piece_read(j + 1);
          if constexpr (k == Sch::KWL) piece_write(j, 1);
          if constexpr (k == Sch::KC) piece_cache(Tn, false, j);
          if constexpr (k == Sch::KD0) dma_piece(Tnn, false, j, 0);
          if constexpr (PRO2 && k == Sch::KD1) dma_piece(Tnn, false, j, 1);
        }
        // column sums of the transformed tile: behind the last piece's sums, an LDS slot of section 1
        if constexpr (PRO2 && sec == 1 && fs == Sch::T0 + Sch::NP1 * Sch::SUBS) colsum_put(it + 1);
        // ---- loads for the next section's epilogue: rows 32*sec .. of the current tile ----
        if constexpr (fs >= Sch::LB && fs < Sch::LB + 16 && !(SA_WSD_ABL & (1 | 16))) {
          epi_load(std::integral_constant<int, fs - Sch::LB>{}, xcur, gcur);
          if constexpr (fs - Sch::LB == 7) { WSD_STEP16(xcur); if constexpr (EP == 4) WSD_STEP16(gcur); }
        }
      };
      // The MFMAs are inline asm so that the weight fragments are AGPR operands where they live; an
      // accumulate chain needs no wait states; the A fragments come from ds_read (waited for by hipcc,
      // which sees the operand); an accumulator is read by VALU code from slot E0 of the NEXT section
      // on (>= 3 MFMAs = 96 cycles behind its last MFMA).
      ws_static_for<0, 2 * Sch::NS>([&](auto S_c) {
        constexpr int S = decltype(S_c)::value, sec = S / Sch::NS, s = S % Sch::NS;
        constexpr int tp = s / WS_KSTEPS, k = s % WS_KSTEPS, sl = S & 1;
        constexpr bool more = S + 1 < 2 * Sch::NS;
        __builtin_amdgcn_sched_barrier(0);
#if SA_WSD_ABL & 4
#define WSD_MFMA(A, BC, B) asm volatile("" : "+v"(acc[sec]) : "v"(A), BC(B))
#else
#define WSD_MFMA(A, BC, B) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[sec]) : "v"(A), BC(B))
#endif
#define WSD_SLOT(I) __builtin_amdgcn_sched_barrier(0); \
        filler(std::integral_constant<int, sec>{}, std::integral_constant<int, 3 * s + (I)>{}); __builtin_amdgcn_sched_barrier(0)
#if SA_WSD_ABL & 4
#define WSD_MFMA_LIT(A, R) asm volatile("" : "+v"(acc[sec]) : "v"(A))
#else
#define WSD_MFMA_LIT(A, R) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[" R "], %0" : "+v"(acc[sec]) : "v"(A))
#endif
        constexpr bool hand = s >= WS_NAGPR_FRAGS && s < WS_NAGPR_FRAGS + WS_NHAND;
        if constexpr (s == 0) {
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc[sec]) : "v"(al), "a"(Bh[0][0]));
        } else if constexpr (s < WS_NAGPR_FRAGS) {
          WSD_MFMA(al, "a", Bh[tp][k]);
        } else if constexpr (hand) {
          if constexpr (s == 30) WSD_MFMA_LIT(al, "240:243"); else WSD_MFMA_LIT(al, "248:251");
        } else {
          WSD_MFMA(al, "v", Bh[tp][k]);
        }
        if constexpr (more) { __builtin_amdgcn_sched_barrier(0); load_a(S + 1); }
        WSD_SLOT(0);
        if constexpr (s < WS_NAGPR_FRAGS) {
          WSD_MFMA(ah[sl], "a", Bl[tp][k]); WSD_SLOT(1);
          WSD_MFMA(ah[sl], "a", Bh[tp][k]); WSD_SLOT(2);
        } else if constexpr (hand) {
          if constexpr (s == 30) { WSD_MFMA_LIT(ah[sl], "244:247"); WSD_SLOT(1); WSD_MFMA_LIT(ah[sl], "240:243"); WSD_SLOT(2); }
          else { WSD_MFMA_LIT(ah[sl], "252:255"); WSD_SLOT(1); WSD_MFMA_LIT(ah[sl], "248:251"); WSD_SLOT(2); }
        } else {
          WSD_MFMA(ah[sl], "v", Bl[tp][k]); WSD_SLOT(1);
          WSD_MFMA(ah[sl], "v", Bh[tp][k]); WSD_SLOT(2);
        }
#undef WSD_MFMA_LIT
#undef WSD_MFMA
#undef WSD_SLOT
      });
    };
    if (fast) {
      tile_body(std::true_type{});
      pend_issued = true;
      WSD_STAMP(it, 2);
    } else {
      tile_body(std::false_type{});
      // MFMA result -> VALU reader wait states (the last MFMA of section 1 has just been issued; acc[0]
      // has been complete for a whole section, but hipcc knows neither)
      asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(acc[0]), "+v"(acc[1]));
      epi_plain(std::integral_constant<int, 0>{}, Tc, yb_c, xb_c, gb_c);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // transform of the next tile (masked at the ends of an utterance)
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
      pend_issued = false;
    }
    if (!fast) colsum_put(it + 1);
    WSD_STAMP(it, 3);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // planes[(it+1) & 1] complete; planes[it & 1] free
    WSD_STAMP(it, 4);
    cs_pending = t + 1 < last;                            // column sums of tile t+1 wait in the LDS scratch
    Tp = Tc; Tc = Tn; Tn = Tnn;
  }
  // ================= tail: rows 32..63 of the last tile =================
  {
    drain_vmem();
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(acc[0]), "+v"(acc[1]));
    stbase_p = const_cast<char*>(row_ptr(a.stats, last - 1, WS_C * 8));
    epi_plain(std::integral_constant<int, 1>{}, Tp, block8(a.y, Tp), block8(a.ep_x, Tp), EP == 4 ? block8(a.ep_g2, Tp) : nullptr);
    epi_stats();
  }
  WSD_STAMP(63, 3); WSD_STAMP_RT(63, 9); WSD_WG_STAMP(1);
  WSD_LIFE(1);
#undef WS_IDS
#undef WSD_IMM
#undef WSD_Q
#undef WSD_STEP16
#undef WSD_XR0
}

template <int NT, int HALO, int PRO, int EP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) __attribute__((amdgpu_num_vgpr(240)))
void sa_conv_wsd_kernel(SaConvArgs a, int bcost, int total_tiles, unsigned long long xw) {
  wsd_body<NT, HALO, PRO, EP>(a, bcost, total_tiles, xw);
}

// extra cost of an utterance end, in tiles (see wsd_body; tools/wsd_ablate.py --bcost sweeps it)
int g_wsd_bcost = 9;

template <int NT, int HALO, int PRO, int EP>
int launch_wsd(const SaConvArgs& a, hipStream_t st) {
  typedef WsGeo<128, NT, HALO> G;
  SaConvArgs args = a;
  args.ntiles = sa_div_up(a.Lout, G::BM);
  int omin = 1 << 30, omax = -(1 << 30), wmax = 0;
  for (int t = 0; t < NT; ++t) {
    omin = a.taps.off[0][t] < omin ? a.taps.off[0][t] : omin;
    omax = a.taps.off[0][t] > omax ? a.taps.off[0][t] : omax;
    wmax = a.taps.widx[0][t] > wmax ? a.taps.widx[0][t] : wmax;
  }
  if (omax - omin != HALO) return -22;
  args.rowmin = omin;
  args.nrows = G::ROWS;
  args.wlo_off = (wmax + 1) * G::KSTEPS * 4 * 64;            // fragment units: size of the hi image
  if ((a.a_out || a.nb_colsum) && (omin > 0 || omax < 0 || (args.ntiles - 1) * G::BM + omin + G::ROWS < a.Lin))
    return -22;                                             // every input row must be staged by the tile that owns it
  if ((long)a.B * a.Lin >= (1L << 31) - 64 || (long)a.B * a.Lout >= (1L << 31) - 64) return -22;   // 32-bit row indices in the kernel
  const size_t lds = 2 * G::BUF_BYTES + (PRO == 2 ? 2 : 1) * G::RAW_BYTES + (PRO == 2 ? 2 * 4 * 128 * 4 : 0);
  if (lds > 160 * 1024) return -12;
  auto kern = sa_conv_wsd_kernel<NT, HALO, PRO, EP>;
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
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, st, args, g_wsd_bcost, total, g_ws_xcd_weights);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// which instantiation serves the launch: 0 = none
int wsd_variant(int dtype, int cin, int cout, int sa, int u, const SaConvArgs* a) {
  if (dtype != SA_BF16X3 || cin != 128 || cout != 128 || sa != 1 || u != 1) return 0;
  if (a->ep_mode != 1 && a->ep_mode != 2) return 0;
  if (!a->ep_x || !a->ep_mean || !a->ep_rstd || !a->stats) return 0;
  if (a->s1 || a->t1 || a->s2 || a->t2 || a->swish || a->relu || a->pro_stats || a->wscale || a->bias) return 0;
  if (a->tile_rows && a->tile_rows != 64) return 0;
  const int nt = a->taps.ntaps[0];
  if (nt != 5 && nt != 3) return 0;
  int omin = 1 << 30, omax = -(1 << 30);
  for (int t = 0; t < nt; ++t) {
    omin = a->taps.off[0][t] < omin ? a->taps.off[0][t] : omin;
    omax = a->taps.off[0][t] > omax ? a->taps.off[0][t] : omax;
  }
  const int halo = omax - omin;
  if (nt == 5 ? halo != 4 : (halo != 4 && halo != 6)) return 0;
  if (omin > 0 || (omin & 1) || omax < 0) return 0;        // ownership tests work on whole DMA pieces (row pairs)
  int pro, ep;
  if (a->nb_x) {
    if (!a->nb_c1 || !a->nb_c2 || !a->nb_c3) return 0;
    pro = 2;
  } else {
    if (a->a_out || a->nb_colsum) return 0;
    pro = 0;
  }
  if (a->ep_mode == 1) {
    if (!a->ep_s1 || !a->ep_t1 || a->ep_xp_is_act || a->ep_bstride != 128) return 0;
    if (a->ep_g2) { if (!a->ep_g2k1 || !a->ep_g2k2 || !a->ep_g2k3) return 0; ep = 4; } else ep = 1;
  } else {
    if (a->ep_g2 || a->ep_bstride != 0) return 0;
    if (a->ep_xp_is_act) { if (!a->ep_s1 || !a->ep_t1) return 0; ep = 3; } else { if (a->ep_s1 || a->ep_t1) return 0; ep = 2; }
  }
  // the ReLU mask of the prologue is compiled in: the BatchNorm blocks (EP 2, 3) have it, the InstanceNorm block not
  if (pro == 2 && (a->nb_relu_mask != 0) != (ep != 1)) return 0;
  // the instantiations the train step needs
  if (nt == 5 && pro == 2 && ep == 1) return 1;            // encoder.11
  if (nt == 5 && pro == 2 && ep == 3) return 2;            // sex_classifier.tdnn.0
  if (nt == 5 && pro == 0 && ep == 4) return 3;            // decoder.0
  if (nt == 3 && halo == 4 && pro == 2 && ep == 2) return 4;   // sex_classifier.tdnn.3
  if (nt == 3 && halo == 6 && pro == 2 && ep == 2) return 5;   // sex_classifier.tdnn.6
  return 0;
}

}  // namespace

extern "C" int sa_conv_wsd_set_bcost(int tiles) {
  // (bit 16, timing A/B only: the first tile of every range takes the plain path, as before round 3's overlap of it)
  if ((tiles & 0xffff) < 4 || (tiles & 0xffff) > 64 || (tiles & ~0x1ffff)) return -22;
  g_wsd_bcost = tiles;
  return 0;
}

extern "C" int sa_conv_ws_calibrate_read(unsigned long long* life512x2) {
  if (!life512x2) return -22;
  return -(int)hipMemcpyFromSymbol(life512x2, HIP_SYMBOL(sa_wsd_life), sizeof(sa_wsd_life));
}

// Does the fused data-gradient kernel serve this launch?  (sa_conv_gemm.hip asks before routing.)
bool sa_conv_wsd_covers(int dtype, int cin, int cout, int sa, int u, const SaConvArgs* a) {
  return wsd_variant(dtype, cin, cout, sa, u, a) != 0;
}

int sa_conv_wsd_dispatch(int cin, int cout, const SaConvArgs* a, hipStream_t st) {
  switch (wsd_variant(SA_BF16X3, cin, cout, 1, 1, a)) {
    case 1: return launch_wsd<5, 4, 2, 1>(*a, st);
    case 2: return launch_wsd<5, 4, 2, 3>(*a, st);
    case 3: return launch_wsd<5, 4, 0, 4>(*a, st);
    case 4: return launch_wsd<3, 4, 2, 2>(*a, st);
    case 5: return launch_wsd<3, 6, 2, 2>(*a, st);
    default: return -38;
  }
}
