// Feature front-end: STFT power spectrum + 80-bin log-Mel filterbank + global normalisation.
//
// Replaces hparams.compute_features (speechbrain.lobes.features.Fbank: sample_rate 16000,
// n_fft 400, n_mels 80 -- speechbrain_configs/convae.yaml:93-95,289-292) and
// modules.normalize (speechbrain InputNormalization, norm_type "global",
// convae.yaml:269-271) at the reference call sites speechbrain_convae_train.py:58-63,82-87.
//
// Kernel 1 (sa_fbank_kernel): one workgroup = 32 frames of one utterance.  The real-input DFT
// is folded: with e[k] = x[k] + x[400-k], o[k] = x[k] - x[400-k] (k = 1..199; e[0] = x[0],
// e[200] = x[200]) the spectrum is re = e . cos, im = o . sin over k = 0..200 -- half the
// multiplies and half the table of the direct [400 x 402] product.  Both GEMMs run on the bf16
// MFMA with split operands: the DFT with a 3-way split (h + m + l = 24 significant bits per
// operand, the six products down to 2^-24 are kept: hh, hm, mh, hl, lh, mm -- fp32-class
// accuracy, a plain bf16 spectrum would put a -48 dB quantisation floor under every frame), the
// all-positive Mel product with the 2-way split of the conv kernels.  Frames are built in LDS
// from coalesced waveform loads, the power spectrum overlays them, and the result is written as
// 10*log10(max(.,1e-10)) with the per-tile maximum for the top-dB clamp.
// Kernel 2/3: per-utterance clamp floor + length-masked mean / unbiased std per Mel bin;
// running global statistics; (x - glob_mean)/glob_std with zero rows appended up to T'.
#include "sa_common.h"

#define SA_NFFT 400
#define SA_HOP 160
#define SA_NBIN 201
#define SA_FK 208            // folded reduction length (201 used), 13 k-steps of 16
#define SA_FKS 13
#define SA_BINT 7            // 32-bin tiles (224 columns, 201 used)
#define SA_MELT 3            // 32-column Mel tiles (96 columns, 80 used)
#define SA_NMEL 80
#define SA_FB_FRAMES 32

// v = h + m + l with h, m, l bf16 (24 significant bits)
__device__ static inline void sa_split3(float v, bf16_t& h, bf16_t& m, bf16_t& l) {
  h = (bf16_t)v;
  const float r1 = v - (float)h;
  m = (bf16_t)r1;
  l = (bf16_t)(r1 - (float)m);
}

// dft: fragment-major bf16 image [cos|sin][h|m|l][13 k-steps][7 bin tiles][64 lanes][8], element
//      (k = ks*16 + 8*(lane>>5) + j, bin = q*32 + (lane&31)); mel: [h|l][13][3][64][8] likewise
//      (k = frequency bin, column = Mel filter).  Built once by the host (features.py).
__global__ __launch_bounds__(256) void sa_fbank_kernel(const float* __restrict__ wav, int N, int T,
                                                       const float* __restrict__ window,
                                                       const bf16x8* __restrict__ dft,
                                                       const bf16x8* __restrict__ mel,
                                                       float* __restrict__ feats,
                                                       float* __restrict__ tilemax, int ntiles) {
  constexpr int PL = SA_FB_FRAMES * SA_FK;                        // elements per plane
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* E = reinterpret_cast<bf16_t*>(smem);                    // [3][32][208] even part
  bf16_t* O = E + 3 * PL;                                         // [3][32][208] odd part
  bf16_t* P = E;                                                  // [2][32][208] power (overlay)
  __shared__ float wmax[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y, t0 = blockIdx.x * SA_FB_FRAMES;
  const float* wb = wav + (size_t)b * N;
  // ---- folded, windowed frames -> split planes (two k per thread and step) ----
  for (int i = tid; i < SA_FB_FRAMES * (SA_FK / 2); i += 256) {
    const int m = i / (SA_FK / 2), k0 = 2 * (i % (SA_FK / 2));
    const int base = (t0 + m) * SA_HOP - SA_NFFT / 2;
    bf16_t eh[2], em[2], el[2], oh[2], om[2], ol[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const int k = k0 + d;
      float xk = 0.0f, xr = 0.0f;
      if (t0 + m < T && k <= SA_NFFT / 2) {
        const int n1 = base + k, n2 = base + SA_NFFT - k;
        if (n1 >= 0 && n1 < N) xk = wb[n1] * window[k];
        if (k >= 1 && k < SA_NFFT / 2 && n2 >= 0 && n2 < N) xr = wb[n2] * window[SA_NFFT - k];
      }
      sa_split3(xk + xr, eh[d], em[d], el[d]);
      sa_split3(xk - xr, oh[d], om[d], ol[d]);
    }
    const int at = m * SA_FK + k0;
    auto put2 = [&](bf16_t* dst, const bf16_t* v) { dst[0] = v[0]; dst[1] = v[1]; };
    put2(E + at, eh); put2(E + PL + at, em); put2(E + 2 * PL + at, el);
    put2(O + at, oh); put2(O + PL + at, om); put2(O + 2 * PL + at, ol);
  }
  __syncthreads();
  // ---- GEMM 1: re = e . cos, im = o . sin; bin tiles q = wave, wave + 4 ----
  f32x16 pw[2];
  const int arow = (lane & 31) * SA_FK + 8 * (lane >> 5);
#pragma unroll
  for (int qi = 0; qi < 2; ++qi) {
    const int q = wave + 4 * qi;
#pragma unroll
    for (int i = 0; i < 16; ++i) pw[qi][i] = 0.0f;
    if (q < SA_BINT) {
      f32x16 cm, cs, sm, ss;                                       // main (h*h) and small-term sums
#pragma unroll
      for (int i = 0; i < 16; ++i) { cm[i] = 0.0f; cs[i] = 0.0f; sm[i] = 0.0f; ss[i] = 0.0f; }
      const bf16x8* ct = dft + (size_t)q * 64 + lane;
      constexpr size_t PLANE = (size_t)SA_FKS * SA_BINT * 64;      // fragments per table plane
#pragma unroll 2
      for (int ks = 0; ks < SA_FKS; ++ks) {
        const bf16x8* cp = ct + (size_t)ks * SA_BINT * 64;
        const bf16x8 ch = cp[0], cmm = cp[PLANE], cl = cp[2 * PLANE];
        const bf16x8 sh = cp[3 * PLANE], smm = cp[4 * PLANE], sl = cp[5 * PLANE];
        const bf16_t* ap = E + arow + ks * 16;
        const bf16x8 eh = *reinterpret_cast<const bf16x8*>(ap);
        const bf16x8 em = *reinterpret_cast<const bf16x8*>(ap + PL);
        const bf16x8 el = *reinterpret_cast<const bf16x8*>(ap + 2 * PL);
        const bf16x8 oh = *reinterpret_cast<const bf16x8*>(ap + 3 * PL);
        const bf16x8 om = *reinterpret_cast<const bf16x8*>(ap + 4 * PL);
        const bf16x8 ol = *reinterpret_cast<const bf16x8*>(ap + 5 * PL);
        cs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(em, cmm, cs, 0, 0, 0);
        cs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(el, ch, cs, 0, 0, 0);
        cs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(eh, cl, cs, 0, 0, 0);
        cs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(em, ch, cs, 0, 0, 0);
        cs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(eh, cmm, cs, 0, 0, 0);
        cm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(eh, ch, cm, 0, 0, 0);
        ss = __builtin_amdgcn_mfma_f32_32x32x16_bf16(om, smm, ss, 0, 0, 0);
        ss = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ol, sh, ss, 0, 0, 0);
        ss = __builtin_amdgcn_mfma_f32_32x32x16_bf16(oh, sl, ss, 0, 0, 0);
        ss = __builtin_amdgcn_mfma_f32_32x32x16_bf16(om, sh, ss, 0, 0, 0);
        ss = __builtin_amdgcn_mfma_f32_32x32x16_bf16(oh, smm, ss, 0, 0, 0);
        sm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(oh, sh, sm, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float re = cm[i] + cs[i], im = sm[i] + ss[i];
        pw[qi][i] = fmaf(re, re, im * im);
      }
    }
  }
  __syncthreads();                                   // every wave is done with E / O
  // ---- power spectrum -> hi / lo planes (bins >= 201 are exact zeros: zero table columns) ----
#pragma unroll
  for (int qi = 0; qi < 2; ++qi) {
    const int bin = (wave + 4 * qi) * 32 + (lane & 31);
    if (wave + 4 * qi < SA_BINT && bin < SA_FK) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const bf16_t h = (bf16_t)pw[qi][i];
        const int at = sa_acc_row(i, lane) * SA_FK + bin;
        P[at] = h;
        P[PL + at] = (bf16_t)(pw[qi][i] - (float)h);
      }
    }
  }
  __syncthreads();
  // ---- GEMM 2: [32 x 208] x [208 x 96] -> Mel energies -> dB ----
  float mx = -INFINITY;
  if (wave < SA_MELT) {
    f32x16 am, as;
#pragma unroll
    for (int i = 0; i < 16; ++i) { am[i] = 0.0f; as[i] = 0.0f; }
    constexpr size_t MPLANE = (size_t)SA_FKS * SA_MELT * 64;
#pragma unroll 2
    for (int ks = 0; ks < SA_FKS; ++ks) {
      const bf16x8* mp = mel + ((size_t)ks * SA_MELT + wave) * 64 + lane;
      const bf16x8 mh = mp[0], ml = mp[MPLANE];
      const bf16_t* ap = P + arow + ks * 16;
      const bf16x8 ph = *reinterpret_cast<const bf16x8*>(ap);
      const bf16x8 pl = *reinterpret_cast<const bf16x8*>(ap + PL);
      as = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pl, mh, as, 0, 0, 0);
      as = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ph, ml, as, 0, 0, 0);
      am = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ph, mh, am, 0, 0, 0);
    }
    const int col = wave * 32 + (lane & 31);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int t = t0 + sa_acc_row(i, lane);
      if (col < SA_NMEL && t < T) {
        const float db = 10.0f * log10f(fmaxf(am[i] + as[i], 1e-10f));
        feats[((size_t)b * T + t) * SA_NMEL + col] = db;
        mx = fmaxf(mx, db);
      }
    }
  }
  mx = sa_wave_max(mx);
  if (lane == 0) wmax[wave] = mx;
  __syncthreads();
  if (tid == 0)
    tilemax[(size_t)b * ntiles + blockIdx.x] = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
}

extern "C" int sa_fbank_ntiles(int T) { return sa_div_up(T, SA_FB_FRAMES); }

// feats: [B][T][80] raw dB (T = 1 + N/160); tilemax: [B][ntiles]
extern "C" int sa_fbank_table_elems(int which) {       // bf16 elements of the dft (0) / mel (1) image
  return which == 0 ? 2 * 3 * SA_FKS * SA_BINT * 64 * 8 : 2 * SA_FKS * SA_MELT * 64 * 8;
}

extern "C" int sa_fbank(const float* wav, int B, int N, const float* window, const void* dft,
                        const void* mel, float* feats, float* tilemax, void* stream) {
  if (!wav || !window || !dft || !mel || !feats || !tilemax || B <= 0 || N <= 0) return -22;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int T = 1 + N / SA_HOP, nt = sa_div_up(T, SA_FB_FRAMES);
  const size_t lds = (size_t)6 * SA_FB_FRAMES * SA_FK * sizeof(bf16_t);
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sa_fbank_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    if (e != hipSuccess) return -(int)e;
    attr = true;
  }
  hipLaunchKernelGGL(sa_fbank_kernel, dim3(nt, B), dim3(256), lds, st, wav, N, T, window,
                     reinterpret_cast<const bf16x8*>(dft), reinterpret_cast<const bf16x8*>(mel), feats,
                     tilemax, nt);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// Per-utterance statistics of the top-dB-clamped features over the first round(len*T) frames:
// floor[b] = max - top_db (per utterance, or the batch max when batch_max != 0), and partial
// (sum, sumsq) per Mel bin for SA_UTT_CHUNKS frame chunks of each utterance (fp64).
#define SA_UTT_CHUNKS 16
__global__ __launch_bounds__(256) void sa_fbank_utt_partial_kernel(const float* __restrict__ feats,
                                                                   const float* __restrict__ tilemax,
                                                                   int ntiles, int B, int T,
                                                                   const float* __restrict__ lens,
                                                                   float top_db, int batch_max,
                                                                   float* __restrict__ floor_out,
                                                                   double* __restrict__ part) {
  __shared__ float smax[256];
  __shared__ double acc[3][SA_NMEL][2];
  const int tid = threadIdx.x, ch = blockIdx.x, b = blockIdx.y;
  float mx = -INFINITY;
  if (batch_max) { for (int i = tid; i < B * ntiles; i += 256) mx = fmaxf(mx, tilemax[i]); }
  else { for (int i = tid; i < ntiles; i += 256) mx = fmaxf(mx, tilemax[(size_t)b * ntiles + i]); }
  smax[tid] = mx;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (tid < s) smax[tid] = fmaxf(smax[tid], smax[tid + s]); __syncthreads(); }
  const float fl = smax[0] - top_db;
  if (tid == 0 && ch == 0) floor_out[b] = fl;
  int n = (int)rintf(lens[b] * (float)T);
  if (n > T) n = T;
  const int per = (T + SA_UTT_CHUNKS - 1) / SA_UTT_CHUNKS;
  const int t0 = ch * per;
  int t1 = t0 + per; if (t1 > n) t1 = n;
  if (tid < 240) {
    const int f = tid % SA_NMEL, p = tid / SA_NMEL;
    double s = 0.0, q = 0.0;
    for (int t = t0 + p; t < t1; t += 3) {
      const float v = fmaxf(feats[((size_t)b * T + t) * SA_NMEL + f], fl);
      s += v; q += (double)v * v;
    }
    acc[p][f][0] = s; acc[p][f][1] = q;
  }
  __syncthreads();
  if (tid < SA_NMEL) {
    double* d = part + (((size_t)b * SA_UTT_CHUNKS + ch) * SA_NMEL + tid) * 2;
    d[0] = acc[0][tid][0] + acc[1][tid][0] + acc[2][tid][0];
    d[1] = acc[0][tid][1] + acc[1][tid][1] + acc[2][tid][1];
  }
}

// InputNormalization "global": per-utterance mean / unbiased std (floored at 1e-10) from the
// chunk partials, averaged over the batch, folded into the running state
// [count, glob_mean[80], glob_std[80]] (speechbrain semantics restated in oracle/features.py).
// 80 features x 8 utterance lanes: lane q handles utterances q, q+8, ... (all loads of one
// utterance are independent), the 8 lane sums are added in lane order by the q = 0 thread.
__global__ __launch_bounds__(640) void sa_norm_update_kernel(const double* __restrict__ part,
                                                             const float* __restrict__ lens, int B, int T,
                                                             int update, int epoch, int update_until_epoch,
                                                             float* state) {
  __shared__ float lm[8][SA_NMEL], ls[8][SA_NMEL];
  const int f = threadIdx.x % SA_NMEL, q = threadIdx.x / SA_NMEL;
  const float count = state[0];
  float cm = 0.f, cs = 0.f;
  for (int b = q; b < B; b += 8) {
    int n = (int)rintf(lens[b] * (float)T);
    if (n > T) n = T;
    double s = 0.0, sq = 0.0;
#pragma unroll
    for (int c = 0; c < SA_UTT_CHUNKS; ++c) {
      const double* d = part + (((size_t)b * SA_UTT_CHUNKS + c) * SA_NMEL + f) * 2;
      s += d[0]; sq += d[1];
    }
    const double m = s / n;
    double var = n > 1 ? (sq - s * m) / (n - 1) : 0.0;
    if (var < 0.0) var = 0.0;
    cm += (float)m;
    cs += fmaxf((float)sqrt(var), 1e-10f);
  }
  lm[q][f] = cm; ls[q][f] = cs;
  __syncthreads();
  if (q == 0) {
    cm = 0.f; cs = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) { cm += lm[r][f]; cs += ls[r][f]; }
    cm /= B; cs /= B;
    if (update) {
      if (count == 0.0f) { state[1 + f] = cm; state[1 + SA_NMEL + f] = cs; }
      else if (epoch < update_until_epoch) {
        const float w = 1.0f / (count + 1.0f);
        state[1 + f] = (1.0f - w) * state[1 + f] + w * cm;
        state[1 + SA_NMEL + f] = (1.0f - w) * state[1 + SA_NMEL + f] + w * cs;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && update) state[0] = count + 1.0f;
}

// out[b][t][f] = (max(feats, floor[b]) - glob_mean[f]) / glob_std[f] for t < T, 0 for T <= t < Tp
__global__ void sa_norm_apply_kernel(const float* __restrict__ feats, const float* __restrict__ floor_b,
                                     const float* __restrict__ state, int T, int Tp,
                                     float* __restrict__ out) {
  const int b = blockIdx.y;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;     // float4 index in [Tp*80/4]
  if (i >= (size_t)Tp * SA_NMEL / 4) return;
  const size_t e = i * 4;
  const int t = (int)(e / SA_NMEL), f = (int)(e % SA_NMEL);
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  if (t < T) {
    const float4 x = *reinterpret_cast<const float4*>(feats + ((size_t)b * T) * SA_NMEL + e);
    const float fl = floor_b[b];
    const float* gm = state + 1 + f;
    const float* gs = state + 1 + SA_NMEL + f;
    o.x = (fmaxf(x.x, fl) - gm[0]) / gs[0]; o.y = (fmaxf(x.y, fl) - gm[1]) / gs[1];
    o.z = (fmaxf(x.z, fl) - gm[2]) / gs[2]; o.w = (fmaxf(x.w, fl) - gm[3]) / gs[3];
  }
  *reinterpret_cast<float4*>(out + ((size_t)b * Tp) * SA_NMEL + e) = o;
}

// feats [B][T][80] raw dB + tilemax -> out [B][Tp][80] normalised (+ zero pad rows);
// scratch: caller-allocated, sa_fbank_scratch_bytes(B) bytes (floor[B] + fp64 chunk partials)
extern "C" int sa_fbank_scratch_bytes(int B) {
  return (int)(((B + 1) & ~1) * sizeof(float) + (size_t)B * SA_UTT_CHUNKS * SA_NMEL * 2 * sizeof(double));
}

extern "C" int sa_fbank_normalize(const float* feats, const float* tilemax, int B, int T, int Tp,
                                  const float* lens, float top_db, int batch_max, int update,
                                  int epoch, int update_until_epoch, float* state, float* scratch,
                                  float* out, void* stream) {
  if (!feats || !tilemax || !lens || !state || !scratch || !out || B <= 0 || T <= 0 || Tp < T)
    return -22;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  float* fl = scratch;
  double* part = reinterpret_cast<double*>(scratch + ((B + 1) & ~1));
  hipLaunchKernelGGL(sa_fbank_utt_partial_kernel, dim3(SA_UTT_CHUNKS, B), dim3(256), 0, st, feats,
                     tilemax, sa_div_up(T, SA_FB_FRAMES), B, T, lens, top_db, batch_max, fl, part);
  hipLaunchKernelGGL(sa_norm_update_kernel, dim3(1), dim3(640), 0, st, part, lens, B, T, update, epoch,
                     update_until_epoch, state);
  const size_t n4 = (size_t)Tp * SA_NMEL / 4;
  hipLaunchKernelGGL(sa_norm_apply_kernel, dim3((unsigned)((n4 + 255) / 256), B), dim3(256), 0, st,
                     feats, fl, state, T, Tp, out);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}
