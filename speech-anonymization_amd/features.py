"""Fbank + InputNormalization with the constructor / call surface the reference's YAML uses.

Drop-in for ``compute_features: !new:speechbrain.lobes.features.Fbank {sample_rate, n_fft,
n_mels}`` and ``normalize: !new:speechbrain.processing.features.InputNormalization
{norm_type: global, update_until_epoch: 4}`` (speechbrain_configs/convae.yaml:269-271,
289-292), called as ``feats = compute_features(wavs)``, ``feats = normalize(feats, wav_lens,
epoch=...)`` at speechbrain_convae_train.py:58-60,82-84.

All arithmetic is HIP (csrc/sa_fbank.hip).  The top-dB clamp needs the per-utterance maximum,
so ``Fbank.__call__`` returns the RAW dB features tagged with their per-tile maxima and the
clamp is applied together with the normalisation (one fused pass).  ``Fbank(...)(wav)`` used
on its own (``.clamped()``) gives the reference's clamped features.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib as L

N_FFT, HOP, NBIN, FK, KSTEPS, BIN_TILES, MEL_ROWS, MEL_COLS = 400, 160, 201, 208, 13, 7, 208, 96


def _hamming(n):
    i = np.arange(n, dtype=np.float64)
    return (0.54 - 0.46 * np.cos(2.0 * np.pi * i / n)).astype(np.float32)      # periodic


def _split_bf16(v, planes):
    """fp32 tensor -> `planes` bf16 tensors whose sum reproduces it to 8*planes bits (RNE)."""
    out, r = [], v.float()
    for _ in range(planes):
        h = r.to(torch.bfloat16)
        out.append(h)
        r = r - h.float()
    return out


def _fragment_image(mat):
    """[K][N] (K % 16 == 0, N % 32 == 0) -> fragment-major [K/16][N/32][64 lanes][8]: element
    (k = ks*16 + 8*(lane>>5) + j, n = q*32 + (lane&31)) -- the B operand of v_mfma_f32_32x32x16_bf16."""
    K, N = mat.shape
    m = mat.reshape(K // 16, 2, 8, N // 32, 32)          # ks, lane>>5, j, q, lane&31
    return m.permute(0, 3, 1, 4, 2).contiguous()         # ks, q, lane>>5, lane&31, j


def _dft_image():
    """Folded real DFT tables cos / sin(2 pi k b / 400), k = 0..200 (rows up to 208 zero), b = 0..200
    (columns up to 224 zero), each split 3-way into bf16: [cos|sin][h|m|l][13][7][64][8]."""
    k = np.arange(FK, dtype=np.float64)[:, None]
    b = np.arange(BIN_TILES * 32, dtype=np.float64)[None, :]
    ang = 2.0 * np.pi * ((k * b) % N_FFT) / N_FFT
    valid = (k <= N_FFT // 2) & (b < NBIN)
    tabs = []
    for f in (np.cos, np.sin):
        t = torch.from_numpy(np.where(valid, f(ang), 0.0).astype(np.float32))
        tabs.append(torch.stack([_fragment_image(p) for p in _split_bf16(t, 3)]))
    return torch.stack(tabs).reshape(-1)


def _mel_image(mel):
    """padded Mel matrix [208][96] fp32 -> [h|l][13][3][64][8] bf16."""
    return torch.stack([_fragment_image(p) for p in _split_bf16(mel, 2)]).reshape(-1)


def _mel_matrix(n_mels, n_fft, sample_rate, f_min=0.0, f_max=None):
    """speechbrain Filterbank (triangular, HTK mel, centre +- left spacing), padded [208][96].
    Computed in fp32 with the same op order as the published implementation."""
    f_max = sample_rate / 2 if f_max is None else f_max
    to_mel = lambda hz: 2595.0 * math.log10(1.0 + hz / 700.0)
    mel = torch.linspace(to_mel(f_min), to_mel(f_max), n_mels + 2)
    hz = 700.0 * (10.0 ** (mel / 2595.0) - 1.0)
    band = (hz[1:] - hz[:-1])[:-1]
    f_central = hz[1:-1]
    n_stft = n_fft // 2 + 1
    all_freqs = torch.linspace(0, sample_rate // 2, n_stft)
    slope = (all_freqs.repeat(n_mels, 1) - f_central.repeat(n_stft, 1).t()) / band.repeat(n_stft, 1).t()
    fb = torch.max(torch.zeros(1), torch.min(slope + 1.0, -slope + 1.0)).t()    # [201][80]
    out = torch.zeros(MEL_ROWS, MEL_COLS)
    out[:n_stft, :n_mels] = fb
    return out


class FbankFeatures:
    """Raw dB features + what the fused clamp/normalise pass needs."""

    def __init__(self, raw, tilemax, top_db, batch_max):
        self.raw, self.tilemax, self.top_db, self.batch_max = raw, tilemax, top_db, batch_max

    @property
    def shape(self):
        return self.raw.shape

    def clamped(self):
        """max(x, amax - top_db): the tensor speechbrain's Fbank returns."""
        if self.batch_max:
            floor = self.tilemax.max() - self.top_db
            return torch.maximum(self.raw, floor)
        floor = self.tilemax.amax(dim=1) - self.top_db
        return torch.maximum(self.raw, floor.view(-1, 1, 1))


class Fbank(torch.nn.Module):
    def __init__(self, sample_rate=16000, n_fft=400, n_mels=80, top_db=80.0,
                 top_db_mode="utterance"):
        super().__init__()
        if (sample_rate, n_fft, n_mels) != (16000, 400, 80):
            raise L.SaHipError("the HIP Fbank is built for sample_rate 16000, n_fft 400, n_mels 80 "
                               "(speechbrain_configs/convae.yaml:93-95)")
        self.top_db, self.batch_max = float(top_db), top_db_mode != "utterance"
        self.register_buffer("window", torch.from_numpy(_hamming(N_FFT)), persistent=False)
        self.register_buffer("dft", _dft_image(), persistent=False)
        self.register_buffer("mel", _mel_image(_mel_matrix(n_mels, n_fft, sample_rate)), persistent=False)

    @torch.no_grad()
    def forward(self, wav):
        lib = L.load()
        if self.window.device != wav.device:
            self.to(wav.device)
        wav = wav.contiguous().float()
        B, N = wav.shape
        T = 1 + N // HOP
        raw = torch.empty(B, T, 80, dtype=torch.float32, device=wav.device)
        tmax = torch.empty(B, lib.sa_fbank_ntiles(T), dtype=torch.float32, device=wav.device)
        L.check(lib.sa_fbank(L.ptr(wav), B, N, L.ptr(self.window), L.ptr(self.dft), L.ptr(self.mel),
                             L.ptr(raw), L.ptr(tmax), L.stream()), "sa_fbank")
        return FbankFeatures(raw, tmax, self.top_db, self.batch_max)


class InputNormalization(torch.nn.Module):
    """norm_type="global" only (what every YAML of the reference uses).  State dict keys follow
    the reference's normalizer.ckpt: count, glob_mean, glob_std, spk_dict_*."""

    def __init__(self, norm_type="global", update_until_epoch=3, pad_multiple=None):
        super().__init__()
        if norm_type != "global":
            raise L.SaHipError("only norm_type='global' is implemented (convae.yaml:269-271)")
        self.update_until_epoch = update_until_epoch
        self.pad_multiple = pad_multiple
        # [count, glob_mean[80], glob_std[80]]
        self.register_buffer("state", torch.zeros(161), persistent=False)

    @property
    def count(self):
        return int(self.state[0].item())

    @property
    def glob_mean(self):
        return self.state[1:81]

    @property
    def glob_std(self):
        return self.state[81:161]

    @torch.no_grad()
    def forward(self, feats, lengths, epoch=0, pad_multiple=None):
        """feats: FbankFeatures (fused clamp) or a plain [B,T,80] tensor.  Returns the normalised
        features; with pad_multiple=m, T is zero-padded up to a multiple of m in the same pass
        (speechbrain_convae_train.py:62-63)."""
        lib = L.load()
        if isinstance(feats, FbankFeatures):
            raw, tmax, top_db, bmax = feats.raw, feats.tilemax, feats.top_db, feats.batch_max
        else:
            raw = feats.contiguous().float()
            B, T, _ = raw.shape
            tmax = torch.full((B, lib.sa_fbank_ntiles(T)), -1e30, device=raw.device)
            top_db, bmax = 0.0, False                       # floor = -1e30: clamp is a no-op
        B, T, _ = raw.shape
        if self.state.device != raw.device:
            self.to(raw.device)
        m = pad_multiple if pad_multiple is not None else self.pad_multiple
        Tp = T if not m or T % m == 0 else T + (m - T % m)
        lens = lengths.to(device=raw.device, dtype=torch.float32).contiguous()
        out = torch.empty(B, Tp, 80, dtype=torch.float32, device=raw.device)
        scratch = torch.empty(lib.sa_fbank_scratch_bytes(B) // 4, dtype=torch.float32, device=raw.device)
        L.check(lib.sa_fbank_normalize(L.ptr(raw), L.ptr(tmax), B, T, Tp, L.ptr(lens),
                                       C.c_float(top_db), int(bmax), int(self.training), int(epoch),
                                       int(self.update_until_epoch), L.ptr(self.state),
                                       L.ptr(scratch), L.ptr(out), L.stream()), "sa_fbank_normalize")
        return out

    def state_dict(self, *a, **k):
        return {"count": self.count, "glob_mean": self.glob_mean.clone(),
                "glob_std": self.glob_std.clone(), "spk_dict_mean": {}, "spk_dict_std": {},
                "spk_dict_count": {}}

    def load_state_dict(self, sd, strict=True):
        with torch.no_grad():
            self.state[0] = float(sd["count"])
            self.state[1:81] = sd["glob_mean"].to(self.state.device).float()
            self.state[81:161] = sd["glob_std"].to(self.state.device).float()
