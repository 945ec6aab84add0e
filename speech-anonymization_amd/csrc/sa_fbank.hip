// Feature front-end: STFT power spectrum + 80-bin log-Mel filterbank + global normalisation.
//
// Replaces hparams.compute_features (speechbrain.lobes.features.Fbank: sample_rate 16000,
// n_fft 400, n_mels 80 -- speechbrain_configs/convae.yaml:93-95,289-292) and
// modules.normalize (speechbrain InputNormalization, norm_type "global",
// convae.yaml:269-271) at the reference call sites speechbrain_convae_train.py:58-63,82-87.
//
// Kernel 1 (sa_fbank_kernel): one workgroup = 32 frames of one utterance.  The 400-sample
// Hamming-windowed frames (hop 160, centre padding 200 zeros each side) are built in LDS
// from coalesced waveform loads, multiplied by the [400 x 416] cos|sin DFT table on the exact
// fp32 MFMA (v_mfma_f32_32x32x2_f32; a bf16 spectrum would put a -48 dB quantisation floor
// under every frame), squared into the power spectrum in LDS, multiplied by the [208 x 96]
// padded Mel matrix on the same MFMA, and written as 10*log10(max(.,1e-10)) with the
// per-tile maximum for the top-dB clamp.
// Kernel 2/3: per-utterance clamp floor + length-masked mean / unbiased std per Mel bin;
// running global statistics; (x - glob_mean)/glob_std with zero rows appended up to T'.
#include "sa_common.h"

#define SA_NFFT 400
#define SA_HOP 160
#define SA_NBIN 201
#define SA_DFT_COLS 416      // 13 tiles x (16 cos | 16 sin)
#define SA_MEL_ROWS 208
#define SA_MEL_COLS 96
#define SA_NMEL 80
#define SA_FB_FRAMES 32

__global__ __launch_bounds__(256) void sa_fbank_kernel(const float* __restrict__ wav, int N, int T,
                                                       const float* __restrict__ window,
                                                       const float* __restrict__ dft,
                                                       const float* __restrict__ mel,
                                                       float* __restrict__ feats,
                                                       float* __restrict__ tilemax, int ntiles) {
  constexpr int APITCH = 401, PPITCH = 209;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* As = reinterpret_cast<float*>(smem);                    // [32][401] windowed frames
  float* Ps = As + SA_FB_FRAMES * APITCH;                        // [32][209] power spectrum
  float* wmax = Ps + SA_FB_FRAMES * PPITCH;                      // [4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y, t0 = blockIdx.x * SA_FB_FRAMES;
  const float* wb = wav + (size_t)b * N;
  for (int i = tid; i < SA_FB_FRAMES * SA_NFFT; i += 256) {
    const int m = i / SA_NFFT, k = i % SA_NFFT;
    const int n = (t0 + m) * SA_HOP + k - SA_NFFT / 2;
    float v = 0.0f;
    if (t0 + m < T && n >= 0 && n < N) v = wb[n] * window[k];
    As[m * APITCH + k] = v;
  }
  for (int i = tid; i < SA_FB_FRAMES * PPITCH; i += 256) Ps[i] = 0.0f;
  __syncthreads();
  // ---- GEMM 1: [32 x 400] x [400 x 416] -> re | im, squared into Ps ----
  for (int q = wave; q < SA_DFT_COLS / 32; q += 4) {
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    const float* ap = As + (lane & 31) * APITCH + (lane >> 5);
    const float* bp = dft + (size_t)(lane >> 5) * SA_DFT_COLS + q * 32 + (lane & 31);
#pragma unroll 8
    for (int k0 = 0; k0 < SA_NFFT; k0 += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[k0], bp[(size_t)k0 * SA_DFT_COLS], acc, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float sq = acc[i] * acc[i];
      const float pw = sq + __shfl_xor(sq, 16, 64);
      const int bin = q * 16 + (lane & 15);
      if ((lane & 16) == 0 && bin < SA_NBIN) Ps[sa_acc_row(i, lane) * PPITCH + bin] = pw;
    }
  }
  __syncthreads();
  // ---- GEMM 2: [32 x 208] x [208 x 96] -> Mel energies -> dB ----
  float mx = -INFINITY;
  if (wave < SA_MEL_COLS / 32) {
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    const float* ap = Ps + (lane & 31) * PPITCH + (lane >> 5);
    const float* bp = mel + (size_t)(lane >> 5) * SA_MEL_COLS + wave * 32 + (lane & 31);
#pragma unroll 8
    for (int k0 = 0; k0 < SA_MEL_ROWS; k0 += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[k0], bp[(size_t)k0 * SA_MEL_COLS], acc, 0, 0, 0);
    const int col = wave * 32 + (lane & 31);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int t = t0 + sa_acc_row(i, lane);
      if (col < SA_NMEL && t < T) {
        const float db = 10.0f * log10f(fmaxf(acc[i], 1e-10f));
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
extern "C" int sa_fbank(const float* wav, int B, int N, const float* window, const float* dft,
                        const float* mel, float* feats, float* tilemax, void* stream) {
  if (!wav || !window || !dft || !mel || !feats || !tilemax || B <= 0 || N <= 0) return -22;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int T = 1 + N / SA_HOP, nt = sa_div_up(T, SA_FB_FRAMES);
  const size_t lds = (size_t)(SA_FB_FRAMES * 401 + SA_FB_FRAMES * 209 + 4) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sa_fbank_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    if (e != hipSuccess) return -(int)e;
    attr = true;
  }
  hipLaunchKernelGGL(sa_fbank_kernel, dim3(nt, B), dim3(256), lds, st, wav, N, T, window, dft, mel,
                     feats, tilemax, nt);
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
