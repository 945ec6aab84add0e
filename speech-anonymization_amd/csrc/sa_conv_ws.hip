// Implicit-GEMM 1-D convolution on MFMA (gfx950), channels-last -- the WEIGHT-STATIONARY form of
// the 128 -> 128 channel, stride-1, 5-tap layers in the bf16x3 policy (fp32 storage, split-bf16
// operands): models/ConvAutoEncoder.py:150-158 (encoder 128->128 blocks) and :161-166 (decoder),
// forward and data gradient -- 10 of the 22 conv launches of a train step and 37 % of its GPU time.
//
// Same operation, arguments, numerics and statistics-slab geometry as the 64-row one-tile kernel
// (sa_conv_gemm.hip); different execution structure, chosen from the measurements in
// profiles/r02_conv_structure_experiments.md (the one-tile kernel's phases do not overlap: row loads
// 76 us + weight-fragment stream 61 us + MFMA 122 us + epilogue 37 us + launch floor 45 us):
//
//   * ONE persistent 4-wave workgroup per CU (one wave per SIMD, the whole 512-register file per
//     lane); a workgroup walks a contiguous range of 64-row tiles.
//   * The weights never move: wave w owns output columns [32w, 32w+32) and keeps its B fragments
//     for all 5 taps x 128 input channels x (hi, lo) = 80 fragments = 320 registers for the whole
//     launch (the one-tile kernel re-streams 327 KB of fragments from L2 per tile).
//   * Input rows arrive by LDS-DMA (global_load_lds_dwordx4, no register destination): the DMA of
//     tile g+1 is issued before the MFMA loop of tile g and has that whole loop to land, so no wave
//     ever waits on HBM; the only global loads of the loop are these.
//   * Transform (normalisation affine + x*sigmoid(x), hi/lo split) goes LDS raw -> registers ->
//     LDS operand planes (double-buffered: one barrier per tile); the epilogue goes straight from
//     the accumulator registers (a register of a 32x32 accumulator = two 128-byte row segments).
//
//   per tile g, per wave:   wait DMA(g) | transform(g) -> planes[g&1] | issue DMA(g+1) |
//                           epilogue(g-1) from registers | barrier | MFMA(g)
//
// Launches it does not cover (pro_stats, the generic two-affine prologue) stay on sa_conv_gemm.hip.
#include <type_traits>
#include "sa_conv_cfg.h"

#ifndef SA_ABL
#define SA_ABL 0
#endif

// -DSA_WS_STAMPS: diagnostic build (tools/ws_stamps.py): s_memtime at the phase boundaries of one
// workgroup's wave 0; no stamp exists in the normal build.
#ifdef SA_WS_STAMPS
__device__ unsigned long long sa_ws_dbg[64 * 8];
#define WS_STAMP(it, i) do { if (lane == 0 && wave == 0 && blockIdx.x == 7 && (it) < 64) { \
  unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
  sa_ws_dbg[(it) * 8 + (i)] = t_; } } while (0)
extern "C" int sa_ws_dbg_read(unsigned long long* out) {
  return -(int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sa_ws_dbg), sizeof(sa_ws_dbg));
}
#else
#define WS_STAMP(it, i)
#endif

namespace {

constexpr int WS_C = 128;                  // channels in = channels out
constexpr int WS_TM = 64;                  // output rows per tile
constexpr int WS_NTAPS = 5;
constexpr int WS_HALO = 4;                 // largest - smallest tap offset
constexpr int WS_ROWS = WS_TM + WS_HALO;   // staged input rows per tile
constexpr int WS_KSTEPS = WS_C / 16;
constexpr int WS_PITCH = WS_C + 8;         // bf16 elements per LDS operand row (272 B: conflict-free ds_read_b128)
constexpr int WS_PLANE = WS_ROWS * WS_PITCH;            // bf16 elements per plane
constexpr int WS_NDMA = WS_ROWS / 2;       // 1-KiB DMA pieces (2 rows each) per tile and tensor
constexpr int WS_DPW = (WS_NDMA + 3) / 4;  // pieces per wave (waves 0,1: 9; waves 2,3: 8)
constexpr int WS_RAW_BYTES = WS_ROWS * WS_C * 4;        // one raw fp32 tile
constexpr int WS_BUF_BYTES = 2 * WS_PLANE * 2;          // one operand buffer (hi + lo planes)
constexpr int WS_NAGPR_FRAGS = 30;         // (tap, k-step) pairs whose hi + lo weight fragments live in AGPRs (240 of 256)

typedef __attribute__((address_space(3))) unsigned char lds_byte;

// one 1-KiB piece: lane l's 16 bytes land at lds_dst + 16*l (cdna_hip_programming.md 5.7: M0 is
// written in the statement that reads it; the s_nop is the M0 -> LDS-DMA wait state)
__device__ static inline void ws_dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// MODE: 0 no transform, 1 affine (per utterance, channel) + x*sigmoid(x), 2 normalisation-backward
// prologue (nb_*: d y = c1*dz + c2*y + c3 [* (y > 0)] over two input tensors)
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void sa_conv_ws_kernel(SaConvArgs a, int tiles_per_wg, int total_tiles) {
  constexpr bool PRO2 = MODE == 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // LDS: [operand buffer 0][operand buffer 1][raw tile x][raw tile nb_x (PRO2)][column-sum scratch (PRO2)]
  bf16_t* const planes = reinterpret_cast<bf16_t*>(smem);
  unsigned char* const raw = smem + 2 * WS_BUF_BYTES;
  const unsigned raw_lds = (unsigned)(uintptr_t)(lds_byte*)raw;
  const int tid = threadIdx.x, lane_ = tid & 63, lane = lane_;
  const int wave_ = __builtin_amdgcn_readfirstlane(tid >> 6), wave = wave_;
  const int first = blockIdx.x * tiles_per_wg;
  int last = first + tiles_per_wg;
  if (last > total_tiles) last = total_tiles;
  if (first >= last) return;

  // ---- the weights: this wave's 32 output columns, all taps / channels, hi and lo images ----
  bf16x8 Bh[WS_NTAPS][WS_KSTEPS], Bl[WS_NTAPS][WS_KSTEPS];
  {
    const bf16x8* wp = reinterpret_cast<const bf16x8*>(a.wp);
#pragma unroll
    for (int t = 0; t < WS_NTAPS; ++t) {
      const bf16x8* wt = wp + ((size_t)a.taps.widx[0][t] * WS_KSTEPS * 4 + wave) * 64 + lane;
#pragma unroll
      for (int k = 0; k < WS_KSTEPS; ++k) {
        Bh[t][k] = wt[(size_t)k * 4 * 64];
        Bl[t][k] = wt[(size_t)a.wlo_off + (size_t)k * 4 * 64];
      }
      // one tap at a time, moved to its AGPR home before the next tap is fetched (all 80 loads at
      // once would need 320 VGPRs)
#pragma unroll
      for (int k = 0; k < WS_KSTEPS; ++k)
        if (t * WS_KSTEPS + k < WS_NAGPR_FRAGS) asm volatile("" : "+a"(Bh[t][k]), "+a"(Bl[t][k]));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const int col = wave * 32 + (lane & 31);
  const float bv = a.bias ? a.bias[col] : 0.0f;
  const int ch = (lane & 31) * 4;                          // this lane's 4 channels in the transform

  // per-utterance transform constants (reloaded when the tile range crosses an utterance)
  float s1[4], t1[4], k1[4], k2[4], k3[4];
  int cur_b = -1;
  auto load_consts = [&](int b) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (MODE == 1) {
        s1[j] = a.s1[(size_t)b * WS_C + ch + j];
        t1[j] = a.t1[(size_t)b * WS_C + ch + j];
      }
      if constexpr (PRO2) {
        const size_t q = (size_t)b * a.nb_bstride + ch + j;
        k1[j] = a.nb_c1[q]; k2[j] = a.nb_c2[q]; k3[j] = a.nb_c3[q];
      }
    }
    cur_b = b;
  };

  // ---- LDS-DMA of the rows of tile t (this wave's pieces) ----
  // (lane / wave ids are made opaque per call: hipcc otherwise hoists the address chains of every
  // piece, row and accumulator register out of the tile loop and keeps -- spills -- them all)
  auto issue_dma = [&](int t) {
    int lane = lane_, wave = wave_;
    asm volatile("" : "+v"(lane), "+s"(wave));
    const int b = t / a.ntiles, tile = t % a.ntiles;
    const int gbase = tile * WS_TM + a.rowmin;
    const char* xb = reinterpret_cast<const char*>(a.x) + (size_t)b * a.Lin * (WS_C * 4) + (lane & 31) * 16;
    const char* x2 = PRO2 ? reinterpret_cast<const char*>(a.nb_x) + (size_t)b * a.Lin * (WS_C * 4) + (lane & 31) * 16 : nullptr;
#pragma unroll
    for (int j = 0; j < WS_DPW; ++j) {
      const int i = wave + 4 * j;                          // piece: rows 2i, 2i+1
      if (i < WS_NDMA) {
        int g = gbase + 2 * i + (lane >> 5);
        g = g < 0 ? 0 : (g >= a.Lin ? a.Lin - 1 : g);      // rows outside the utterance: any valid address (zeroed in the transform)
        ws_dma16(xb + (size_t)g * (WS_C * 4), raw_lds + i * 1024);
        if constexpr (PRO2) ws_dma16(x2 + (size_t)g * (WS_C * 4), raw_lds + WS_RAW_BYTES + i * 1024);
      }
    }
  };

  // ---- transform of tile t: raw (this wave's own pieces) -> operand planes[t & 1] ----
  float csum[4];
  auto transform = [&](int t, int it) {
    int lane = lane_, wave = wave_;
    asm volatile("" : "+v"(lane), "+s"(wave));
    const int ch = (lane & 31) * 4;
    const int b = t / a.ntiles, tile = t % a.ntiles, m0 = tile * WS_TM;
    if (MODE != 0 && b != cur_b) load_consts(b);
    const int gbase = m0 + a.rowmin;
    const int own_lo = m0;
    int own_hi = tile == a.ntiles - 1 ? a.Lin : m0 + WS_TM;
    if (own_hi > a.Lin) own_hi = a.Lin;
    bf16_t* dstb = planes + (size_t)(it & 1) * 2 * WS_PLANE + ch;
    bf16_t* ao = a.a_out ? reinterpret_cast<bf16_t*>(a.a_out) + (size_t)b * a.Lin * WS_C + ch : nullptr;
#pragma unroll
    for (int j = 0; j < 4; ++j) csum[j] = 0.0f;
    // the pieces go through registers one at a time, the next one's LDS reads in flight behind the
    // current one's arithmetic (pinned: left alone hipcc reads all nine pieces up front -- 72
    // registers in the PRO2 form -- and spills weight fragments to make room)
    f32x4 vx[2], vy[2];
    auto read_piece = [&](int slot, int j) {
      const int i = wave + 4 * j;
      vx[slot] = *reinterpret_cast<const f32x4*>(raw + i * 1024 + lane * 16);
      if constexpr (PRO2) vy[slot] = *reinterpret_cast<const f32x4*>(raw + WS_RAW_BYTES + i * 1024 + lane * 16);
    };
    read_piece(0, 0);
#pragma unroll
    for (int j = 0; j < WS_DPW; ++j) {
      const int i = wave + 4 * j;
      if (j + 1 < WS_DPW) read_piece((j + 1) & 1, j + 1);   // (piece 34/35 of waves 2, 3: inside the raw tile of the next tensor / scratch, unused)
      __builtin_amdgcn_sched_barrier(0);
      if (i < WS_NDMA) {
        const int r = 2 * i + (lane >> 5), g = gbase + r;
        float f[4];
        const f32x4 v = vx[j & 1];
        f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3];
        const bool in = g >= 0 && g < a.Lin;
        if constexpr (MODE == 1) {
#pragma unroll
          for (int q = 0; q < 4; ++q) f[q] = sa_swish(fmaf(f[q], s1[q], t1[q]));
        }
        if constexpr (PRO2) {
          const f32x4 y = vy[j & 1];
          const bool own = g >= own_lo && g < own_hi;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float u = fmaf(k1[q], f[q], fmaf(k2[q], y[q], k3[q]));
            if (a.nb_relu_mask && !(y[q] > 0.0f)) u = 0.0f;
            f[q] = u;
            if (own) csum[q] += u;
          }
        }
        if (!in) { f[0] = 0.0f; f[1] = 0.0f; f[2] = 0.0f; f[3] = 0.0f; }
        uint2 hi, lo;
        sa_split4(f, hi, lo);
        bf16_t* dst = dstb + (size_t)r * WS_PITCH;
        *reinterpret_cast<uint2*>(dst) = hi;
        *reinterpret_cast<uint2*>(dst + WS_PLANE) = lo;
        // bf16 operand cache for sa_wgrad: the hi values of the rows this tile owns
        if (ao && g >= own_lo && g < own_hi) *reinterpret_cast<uint2*>(ao + (size_t)g * WS_C) = hi;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (PRO2) {
      if (a.nb_colsum) {
        // column sums of d y over the owned rows: fold the two row halves of the wave; one LDS slot
        // per wave, summed after the tile barrier
        float* colred = reinterpret_cast<float*>(raw + 2 * WS_RAW_BYTES) + (size_t)(it & 1) * 4 * WS_C;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float v = csum[q] + __shfl_xor(csum[q], 32, 64);
          if (lane < 32) colred[wave * WS_C + ch + q] = v;
        }
      }
    }
  };

  // ---- epilogue of tile t from the accumulator registers ----
  f32x16 acc[2];
  auto epilogue = [&](int t) {
    int lane = lane_, wave = wave_;
    asm volatile("" : "+v"(lane), "+s"(wave));
    const int col = wave * 32 + (lane & 31);
    const int b = t / a.ntiles, tile = t % a.ntiles, m0 = tile * WS_TM;
    // uniform base + one 32-bit lane offset: the 32 stores of a tile are then immediate offsets from
    // 8 registers (as 64-bit per-store addresses they cost 64 registers, and weight fragments spill)
    char* ybase = reinterpret_cast<char*>(a.y) + ((size_t)b * a.Lout + m0) * (WS_C * 4);
    const unsigned loff = ((4 * (lane >> 5)) * WS_C + col) * 4;
    float ssum = 0.0f, ssq = 0.0f;
    auto body = [&](auto full_c) {
      constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int ro = m * 32 + (i & 3) + 8 * (i >> 2);           // row in the tile, before the lane half's +4
          float val = acc[m][i] + bv;
          if (a.relu) val = fmaxf(val, 0.0f);
          if (FULL || m0 + ro + 4 * (lane >> 5) < a.Lout) {
            *reinterpret_cast<float*>(ybase + (loff + (unsigned)ro * (WS_C * 4))) = val;
            ssum += val; ssq = fmaf(val, val, ssq);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    const bool full = m0 + WS_TM <= a.Lout;                // uniform
    if ((SA_ABL & 4) && a.B > 0) {
      if (acc[0][0] + acc[1][5] == 1.2345e-33f) ybase[loff] = 0;
    } else if (full) {
      body(std::true_type{});
    } else {
      body(std::false_type{});
    }
    if (a.stats) {
      ssum += __shfl_xor(ssum, 32, 64);
      ssq += __shfl_xor(ssq, 32, 64);
      if (lane < 32) {
        float* dst = a.stats + (((size_t)b * a.ntiles + tile) * WS_C + col) * 2;
        dst[0] = ssum; dst[1] = ssq;
      }
    }
  };

  // ---- MFMA loop of the tile staged in planes[it & 1] ----
  int toff[WS_NTAPS];
#pragma unroll
  for (int t = 0; t < WS_NTAPS; ++t) toff[t] = (a.taps.off[0][t] - a.rowmin) * WS_PITCH;
  // The MFMAs are inline asm so that the weight fragments are AGPR operands where they live (left to
  // itself hipcc parks them in AGPRs and copies each one to VGPRs in front of every use, with one
  // A-fragment buffer and lgkmcnt(0) per MFMA).  The first 30 (tap, k-step) pairs (60 fragments) take 240 of the 256 AGPRs
  // (hipcc needs a few as spill / reload temporaries), the other 10 pairs stay in VGPRs.  Hazards (cdna_hip_programming.md 5.7 item 2): an accumulate chain needs no wait
  // states; the A fragments come from ds_read (waited for by hipcc, which sees the operand); the
  // accumulators are read by VALU code only after the s_nop block below.
  auto mfma_tile = [&](int it) {
    int lane = lane_;
    asm volatile("" : "+v"(lane));
    const bf16_t* ab = planes + (size_t)(it & 1) * 2 * WS_PLANE + (lane & 31) * WS_PITCH + (lane >> 5) * 8;
    constexpr int NS = WS_NTAPS * WS_KSTEPS;
    bf16x8 ah[2][2], al[2][2];                             // [slot][m-tile]: one step ahead of the MFMAs
    auto load_a = [&](int slot, int s) {
      const int t = s / WS_KSTEPS, k = s % WS_KSTEPS;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const bf16_t* ap = ab + toff[t] + m * 32 * WS_PITCH + k * 16;
        ah[slot][m] = *reinterpret_cast<const bf16x8*>(ap);
        al[slot][m] = *reinterpret_cast<const bf16x8*>(ap + WS_PLANE);
      }
    };
    load_a(0, 0);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (s + 1 < NS) load_a((s + 1) & 1, s + 1);
      __builtin_amdgcn_sched_barrier(0);
      const int t = s / WS_KSTEPS, k = s % WS_KSTEPS, sl = s & 1;
      if constexpr ((SA_ABL & 1) != 0) {
        asm volatile("" :: "v"(ah[sl][0]), "v"(al[sl][0]), "v"(ah[sl][1]), "v"(al[sl][1]));
      } else if (s == 0) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %4, 0\n\tv_mfma_f32_32x32x16_bf16 %1, %3, %4, 0"
                     : "=&v"(acc[0]), "=&v"(acc[1]) : "v"(al[sl][0]), "v"(al[sl][1]), "a"(Bh[0][0]));
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %4, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %3, %4, %1"
                     : "+v"(acc[0]), "+v"(acc[1]) : "v"(ah[sl][0]), "v"(ah[sl][1]), "a"(Bl[0][0]));
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %4, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %3, %4, %1"
                     : "+v"(acc[0]), "+v"(acc[1]) : "v"(ah[sl][0]), "v"(ah[sl][1]), "a"(Bh[0][0]));
      } else if (t * WS_KSTEPS + k < WS_NAGPR_FRAGS) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %4, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %3, %4, %1"
                     : "+v"(acc[0]), "+v"(acc[1]) : "v"(al[sl][0]), "v"(al[sl][1]), "a"(Bh[t][k]));
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %4, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %3, %4, %1"
                     : "+v"(acc[0]), "+v"(acc[1]) : "v"(ah[sl][0]), "v"(ah[sl][1]), "a"(Bl[t][k]));
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %4, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %3, %4, %1"
                     : "+v"(acc[0]), "+v"(acc[1]) : "v"(ah[sl][0]), "v"(ah[sl][1]), "a"(Bh[t][k]));
      } else {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %4, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %3, %4, %1"
                     : "+v"(acc[0]), "+v"(acc[1]) : "v"(al[sl][0]), "v"(al[sl][1]), "v"(Bh[t][k]));
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %4, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %3, %4, %1"
                     : "+v"(acc[0]), "+v"(acc[1]) : "v"(ah[sl][0]), "v"(ah[sl][1]), "v"(Bl[t][k]));
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %4, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %3, %4, %1"
                     : "+v"(acc[0]), "+v"(acc[1]) : "v"(ah[sl][0]), "v"(ah[sl][1]), "v"(Bh[t][k]));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // MFMA result -> VALU reader wait states (the epilogue runs after the next transform anyway)
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[0]), "+v"(acc[1]));
  };

  // ================= the tile walk =================
  issue_dma(first);
  for (int t = first, it = 0; t <= last; ++t, ++it) {
    WS_STAMP(it, 0);
    if (t < last) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // DMA(t) has landed (issued one MFMA loop ago)
      WS_STAMP(it, 1);
      transform(t, it);
      WS_STAMP(it, 2);
      if (t + 1 < last) issue_dma(t + 1);                  // this wave's raw pieces are consumed: refill them
    }
    if (t > first) epilogue(t - 1);
    WS_STAMP(it, 3);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // planes[it & 1] complete; planes[(it+1) & 1] free
    WS_STAMP(it, 4);
    if constexpr (PRO2) {
      if (t < last && a.nb_colsum && tid < WS_C) {
        const float* colred = reinterpret_cast<const float*>(raw + 2 * WS_RAW_BYTES) + (size_t)(it & 1) * 4 * WS_C;
        a.nb_colsum[(size_t)t * WS_C + tid] =
            (colred[tid] + colred[WS_C + tid]) + (colred[2 * WS_C + tid] + colred[3 * WS_C + tid]);
      }
    }
    if (t < last) mfma_tile(it);
    WS_STAMP(it, 5);
  }
}

template <int MODE>
int launch_ws(const SaConvArgs& a, hipStream_t st) {
  SaConvArgs args = a;
  args.ntiles = sa_div_up(a.Lout, WS_TM);
  int omin = 1 << 30, omax = -(1 << 30), wmax = 0;
  for (int t = 0; t < WS_NTAPS; ++t) {
    omin = a.taps.off[0][t] < omin ? a.taps.off[0][t] : omin;
    omax = a.taps.off[0][t] > omax ? a.taps.off[0][t] : omax;
    wmax = a.taps.widx[0][t] > wmax ? a.taps.widx[0][t] : wmax;
  }
  if (omax - omin != WS_HALO) return -22;
  args.rowmin = omin;
  args.nrows = WS_ROWS;
  args.wlo_off = (wmax + 1) * WS_KSTEPS * 4 * 64;        // fragment units: size of the hi image
  if ((a.a_out || a.nb_colsum) &&
      (omin > 0 || omax < 0 || (args.ntiles - 1) * WS_TM + omin + WS_ROWS < a.Lin))
    return -22;                                           // every input row must be staged by the tile that owns it
  const size_t lds = 2 * WS_BUF_BYTES + (MODE == 2 ? 2 : 1) * WS_RAW_BYTES + (MODE == 2 ? 2 * 4 * WS_C * 4 : 0);
  auto kern = sa_conv_ws_kernel<MODE>;
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
  const int per = sa_div_up(total, n_cu);
  const int nwg = sa_div_up(total, per);
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, st, args, per, total);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

}  // namespace

// Does the weight-stationary kernel serve this launch?  (sa_conv_gemm.hip asks before routing.)
bool sa_conv_ws_covers(int dtype, int cin, int cout, int sa, int u, const SaConvArgs* a) {
  if (dtype != SA_BF16X3 || cin != WS_C || cout != WS_C || sa != 1 || u != 1) return false;
  if (a->taps.ntaps[0] != WS_NTAPS || a->pro_stats || a->s2 || a->t2 || a->wscale) return false;
  if (a->ep_mode) return false;
  if (a->tile_rows && a->tile_rows != WS_TM) return false;
  if (a->nb_x) return !a->s1 && !a->swish && a->nb_c1 && a->nb_c2 && a->nb_c3;
  if (a->s1) return a->t1 && a->swish;                    // affine + x*sigmoid(x)
  return !a->swish;
}

int sa_conv_ws_dispatch(const SaConvArgs* a, hipStream_t st) {
  if (a->nb_x) return launch_ws<2>(*a, st);
  if (a->s1) return launch_ws<1>(*a, st);
  return launch_ws<0>(*a, st);
}
