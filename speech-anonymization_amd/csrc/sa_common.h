// Common device helpers for the speech-anonymization HIP library (gfx950 / CDNA4 only).
//
// Data layout used by every kernel in this library: activations are CHANNELS-LAST,
// [B][L][C] with C contiguous ("rows" = positions on the flattened T*80 axis the reference
// ConvAutoencoder convolves over, models/ConvAutoEncoder.py:181-188).  The reduction
// dimension of conv forward / dgrad (channels) is then contiguous, which is what the MFMA
// operand fragments want (8 consecutive k per lane for v_mfma_f32_32x32x16_bf16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/sa_hip.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SA_WAVE 64


// ---- 16-byte chunk <-> float[VEC] -------------------------------------------------
template <typename T> struct Tr;

template <> struct Tr<float> {
  static constexpr int VEC = 4;
  static constexpr int KS = 2;           // k per v_mfma_f32_32x32x2_f32
  typedef float Frag;
  __device__ static inline void unpack(const uint4& u, float* f) {
    f[0] = __uint_as_float(u.x); f[1] = __uint_as_float(u.y);
    f[2] = __uint_as_float(u.z); f[3] = __uint_as_float(u.w);
  }
  __device__ static inline uint4 pack(const float* f) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]),
                      __float_as_uint(f[2]), __float_as_uint(f[3]));
  }
  __device__ static inline float to_f(float v) { return v; }
  __device__ static inline float from_f(float v) { return v; }
  __device__ static inline f32x16 mfma(Frag a, Frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  }
};

__device__ static inline uint32_t sa_pack_bf16x2(float lo, float hi) {
  union { bf16_t b[2]; uint32_t u; } cv;
  cv.b[0] = (bf16_t)lo; cv.b[1] = (bf16_t)hi;     // v_cvt_pk_bf16_f32 (RNE, NaN-safe)
  return cv.u;
}

template <> struct Tr<bf16_t> {
  static constexpr int VEC = 8;
  static constexpr int KS = 16;          // k per v_mfma_f32_32x32x16_bf16
  typedef bf16x8 Frag;
  __device__ static inline void unpack(const uint4& u, float* f) {
    f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
    f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
    f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
    f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
  }
  __device__ static inline uint4 pack(const float* f) {
    return make_uint4(sa_pack_bf16x2(f[0], f[1]), sa_pack_bf16x2(f[2], f[3]),
                      sa_pack_bf16x2(f[4], f[5]), sa_pack_bf16x2(f[6], f[7]));
  }
  __device__ static inline float to_f(bf16_t v) { return (float)v; }
  __device__ static inline bf16_t from_f(float v) { return (bf16_t)v; }
  __device__ static inline f32x16 mfma(Frag a, Frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};

// x * sigmoid(x): the reference's "GLU" (models/ConvAutoEncoder.py:119-120)
// (v_exp_f32 + v_rcp_f32: ~1 ulp each; a full IEEE division would cost ~10 VALU ops per element
// in every prologue)
__device__ static inline float sa_swish(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
// d/dv [v * sigmoid(v)] = s * (1 + v * (1 - s))
// (spelled out operation by operation: the weight-stationary data-gradient kernel, sa_conv_wsd.hip,
// issues exactly these instructions from its filler slots and must produce the same bits)
__device__ static inline float sa_swish_grad(float v) {
  const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f));
  const float t = 1.0f - s;
  const float u = fmaf(v, t, 1.0f);
  return s * u;
}

// Row of the C/D accumulator of a 32x32 MFMA held in register `reg` of lane `lane`
// (col = lane & 31): cdna_hip_programming.md section 3.
__device__ static inline int sa_acc_row(int reg, int lane) {
  return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
}

__device__ static inline float sa_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ static inline double sa_wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ static inline float sa_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ABI structs (SaTaps, SaConvArgs, SaWgradArgs, SaEwArgs) live in include/sa_hip.h
static inline int sa_div_up(int a, int b) { return (a + b - 1) / b; }

// ---- precision policies of the MFMA kernels -------------------------------------------
//   SA_F32    : fp32 storage, fp32 LDS operands, v_mfma_f32_32x32x2_f32 (exact fp32)
//   SA_BF16   : bf16 storage, bf16 LDS operands, one v_mfma_f32_32x32x16_bf16 per k-step
//   SA_BF16X3 : fp32 storage; each operand is split on the fly into hi = bf16(v) and
//               lo = bf16(v - hi) (two LDS planes / two weight images) and every k-step issues
//               hi*hi + lo*hi + hi*lo on the bf16 MFMA: ~16 mantissa bits per operand at 3/16 of
//               the fp32-MFMA cost.  This is the mode that meets the 1e-4 parity bar.
struct bf16x3_t {};
//   SA_BF16X1F: fp32 storage, operands rounded once to bf16 in LDS, one bf16 MFMA per k-step.
//               Used for WEIGHT GRADIENTS in bf16x3 models: a wgrad inner product sums ~10^5
//               independent roundings that average out and its result is not propagated through
//               further layers, so the split buys nothing there (tools/precision_probe.py).
struct bf16x1f_t {};

//   SA_FP8    : bf16 storage; LDS operands and the weight image are OCP e4m3 (gfx950's fp8), one
//               v_mfma_f32_32x32x16_fp8_fp8 per k-step (same rate as bf16), fp32 accumulation.
//               Activations are quantised as they are staged (they are post-normalisation, O(1):
//               no scale; values beyond +-448 saturate), the weights carry a per-tensor power-of-two
//               scale that the epilogue divides out.  Forward launches only.
struct fp8_t {};
typedef long fp8x8;                       // 8 e4m3 values: one MFMA operand fragment

template <typename T> struct Pol;
template <> struct Pol<float> {
  typedef float store_t; typedef float lds_t; typedef float Frag;
  static constexpr int NPL = 1, VEC = 4, KS = 2, PAD = 1;
};
template <> struct Pol<bf16_t> {
  typedef bf16_t store_t; typedef bf16_t lds_t; typedef bf16x8 Frag;
  static constexpr int NPL = 1, VEC = 8, KS = 16, PAD = 8;
};
template <> struct Pol<bf16x1f_t> {
  typedef float store_t; typedef bf16_t lds_t; typedef bf16x8 Frag;
  static constexpr int NPL = 1, VEC = 4, KS = 16, PAD = 8;
};
template <> struct Pol<bf16x3_t> {
  typedef float store_t; typedef bf16_t lds_t; typedef bf16x8 Frag;
  static constexpr int NPL = 2, VEC = 4, KS = 16, PAD = 8;
};

template <> struct Pol<fp8_t> {
  typedef bf16_t store_t; typedef unsigned char lds_t; typedef fp8x8 Frag;
  static constexpr int NPL = 1, VEC = 8, KS = 16, PAD = 8;      // pitch = C + 8 bytes: conflict-free ds_read_b64
};
template <> struct Tr<unsigned char> {
  __device__ static inline f32x16 mfma(fp8x8 a, fp8x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a, b, c, 0, 0, 0);
  }
};
// 8 floats -> 8 e4m3 (v_cvt_pk_fp8_f32, RNE; clamped to the finite range first)
__device__ static inline uint2 sa_pack_fp8x8(const float* f) {
  float c[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) c[j] = __builtin_amdgcn_fmed3f(f[j], -448.0f, 448.0f);
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[4], c[5], hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[6], c[7], hi, true);
  return make_uint2((unsigned)lo, (unsigned)hi);
}

__device__ static inline uint2 sa_pack_bf16x4(const float* f) {
  return make_uint2(sa_pack_bf16x2(f[0], f[1]), sa_pack_bf16x2(f[2], f[3]));
}
// split 4 floats into hi / lo bf16 quadruples
__device__ static inline void sa_split4(const float* f, uint2& hi, uint2& lo) {
  float h[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const bf16_t hb = (bf16_t)f[j];
    h[j] = (float)hb;
    l[j] = f[j] - h[j];
  }
  hi = sa_pack_bf16x4(h);
  lo = sa_pack_bf16x4(l);
}
