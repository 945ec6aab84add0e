"""Oracle: CPU restatement of the feature front-end.

TEST INFRASTRUCTURE (see oracle/__init__.py).  The arithmetic lives in speechbrain
(un-vendored; pinned at speechbrain/speechbrain@ff3bca4c in the reference's env.log) and
is restated here from the published v0.5.x algorithm -- "parity unpinned" except for the
statistical pin of the reference's normalizer.ckpt (tests/golden/normalizer_pin.json).
Reference call sites: speechbrain_convae_train.py:58-63,82-87; config
speechbrain_configs/convae.yaml:93-95,269-271,289-292.

  Fbank(sample_rate=16000, n_fft=400, n_mels=80):
      torch.stft(n_fft=400, hop=160, win=400, hamming(periodic), center=True,
                 pad_mode="constant", onesided) -> re^2+im^2 -> @ fbank[201,80]
      (triangular filters on the HTK mel scale, centre +- the LEFT mel spacing)
      -> 10*log10(clamp(.,1e-10)) -> max(x, amax - 80)   [amax per utterance by default]
  InputNormalization(norm_type="global", update_until_epoch=4): see class below.
  pad T up to a multiple of 36 with zeros (speechbrain_convae_train.py:62-63).
"""
import math
import torch


def hamming_window(n=400):
    """torch.hamming_window(n) (periodic): 0.54 - 0.46 cos(2 pi i / n)."""
    return torch.hamming_window(n)


def mel_filterbank(n_mels=80, n_fft=400, sample_rate=16000, f_min=0.0, f_max=None):
    """speechbrain Filterbank triangular matrix [n_fft//2+1, n_mels]."""
    if f_max is None:
        f_max = sample_rate / 2
    to_mel = lambda hz: 2595.0 * math.log10(1.0 + hz / 700.0)
    mel = torch.linspace(to_mel(f_min), to_mel(f_max), n_mels + 2)
    hz = 700.0 * (10.0 ** (mel / 2595.0) - 1.0)
    band = (hz[1:] - hz[:-1])[:-1]
    f_central = hz[1:-1]
    n_stft = n_fft // 2 + 1
    all_freqs = torch.linspace(0, sample_rate // 2, n_stft)
    fc = f_central.repeat(n_stft, 1).transpose(0, 1)
    bd = band.repeat(n_stft, 1).transpose(0, 1)
    slope = (all_freqs.repeat(n_mels, 1) - fc) / bd
    fb = torch.max(torch.zeros(1), torch.min(slope + 1.0, -slope + 1.0))
    return fb.transpose(0, 1).contiguous()


class Fbank:
    def __init__(self, sample_rate=16000, n_fft=400, n_mels=80, top_db=80.0,
                 top_db_mode="utterance"):
        self.n_fft, self.hop, self.win = n_fft, sample_rate // 100, sample_rate // 40
        self.window = hamming_window(self.win)
        self.fb = mel_filterbank(n_mels, n_fft, sample_rate)
        self.top_db, self.top_db_mode = top_db, top_db_mode
        self.amin = 1e-10

    def power_spectrum(self, wav):
        st = torch.stft(wav, self.n_fft, self.hop, self.win, self.window, center=True,
                        pad_mode="constant", normalized=False, onesided=True,
                        return_complex=True)
        st = torch.view_as_real(st).transpose(2, 1)          # [B, T, 201, 2]
        return st.pow(2).sum(-1)

    @torch.no_grad()
    def __call__(self, wav):
        spec = self.power_spectrum(wav)
        fb = torch.matmul(spec, self.fb)
        x_db = 10.0 * torch.log10(torch.clamp(fb, min=self.amin))
        if self.top_db_mode == "utterance":
            floor = x_db.amax(dim=(-2, -1)) - self.top_db
            x_db = torch.max(x_db, floor.view(-1, 1, 1))
        else:                                                 # older speechbrain: batch max
            x_db = torch.max(x_db, x_db.max() - self.top_db)
        return x_db


def power_spectrum_direct_dft(wav, n_fft=400, hop=160):
    """O(N^2) float64 DFT cross-check of Fbank.power_spectrum (numpy, no torch.stft)."""
    import numpy as np
    w = wav.double().numpy()
    B, N = w.shape
    T = 1 + N // hop
    pad = n_fft // 2
    wp = np.pad(w, ((0, 0), (pad, pad)))
    win = 0.54 - 0.46 * np.cos(2 * np.pi * np.arange(n_fft) / n_fft)
    k = np.arange(n_fft // 2 + 1)[:, None] * np.arange(n_fft)[None, :]
    C, S = np.cos(2 * np.pi * k / n_fft), np.sin(2 * np.pi * k / n_fft)
    out = np.zeros((B, T, n_fft // 2 + 1))
    for t in range(T):
        fr = wp[:, t * hop:t * hop + n_fft] * win
        out[:, t] = (fr @ C.T) ** 2 + (fr @ S.T) ** 2
    return torch.from_numpy(out)


class InputNormalization:
    """speechbrain.processing.features.InputNormalization, norm_type="global".

    Per utterance mean / unbiased std over the first round(len*T) frames (std floored at
    1e-10) -> averaged over the batch -> while ``training and epoch < update_until_epoch``
    folded into the running glob_mean/glob_std with weight 1/(count+1) (first call: copy)
    -> x = (x - glob_mean) / glob_std.  State keys as in the reference's normalizer.ckpt.
    """

    def __init__(self, norm_type="global", update_until_epoch=4):
        assert norm_type == "global"
        self.update_until_epoch = update_until_epoch
        self.eps = 1e-10
        self.count = 0
        self.glob_mean = torch.tensor([0.0])
        self.glob_std = torch.tensor([0.0])
        self.training = True

    def batch_stats(self, x, lengths):
        means, stds = [], []
        for i in range(x.shape[0]):
            n = int(torch.round(lengths[i] * x.shape[1]))
            means.append(x[i, 0:n].mean(dim=0))
            stds.append(torch.max(x[i, 0:n].std(dim=0), torch.tensor(self.eps)))
        return torch.stack(means).mean(dim=0), torch.stack(stds).mean(dim=0)

    @torch.no_grad()
    def __call__(self, x, lengths, epoch=0):
        cur_mean, cur_std = self.batch_stats(x, lengths)
        if self.training:
            if self.count == 0:
                self.glob_mean, self.glob_std = cur_mean, cur_std
            elif epoch < self.update_until_epoch:
                w = 1.0 / (self.count + 1)
                self.glob_mean = (1 - w) * self.glob_mean + w * cur_mean
                self.glob_std = (1 - w) * self.glob_std + w * cur_std
            self.count += 1
        return (x - self.glob_mean) / self.glob_std

    def state_dict(self):
        return {"count": self.count, "glob_mean": self.glob_mean, "glob_std": self.glob_std,
                "spk_dict_mean": {}, "spk_dict_std": {}, "spk_dict_count": {}}


def pad_to_multiple(feats, m=36):
    """speechbrain_convae_train.py:62-63 (pads even when the remainder is zero? no:
    only ``if feats.shape[1] % 36 != 0``)."""
    T = feats.shape[1]
    if T % m != 0:
        feats = torch.nn.functional.pad(feats, (0, 0, 0, m - T % m, 0, 0))
    return feats


def synthetic_wave(B, N, seed=8886, rank=0):
    """0.1*randn + three sinusoids (220 Hz, 1 kHz, 3.4 kHz; 0.2/0.1/0.05)
    clipped to [-1, 1], fp32, 16 kHz.  numpy RNG so the GPU box draws the same samples."""
    import numpy as np
    rs = np.random.RandomState(seed + rank)
    t = np.arange(N, dtype=np.float64) / 16000.0
    w = 0.1 * rs.standard_normal((B, N))
    for f, a in ((220.0, 0.2), (1000.0, 0.1), (3400.0, 0.05)):
        w = w + a * np.sin(2 * np.pi * f * t)[None, :]
    return torch.from_numpy(np.clip(w, -1.0, 1.0).astype("float32"))


def synthetic_feats(B, T, seed=1):
    """Normalised-Fbank-like test features [B, T, 80] whose utterances differ in STRUCTURE
    (harmonic ridges at utterance-specific Mel positions, different temporal modulation), not
    just in scale: InstanceNorm removes per-utterance scale/offset, and with white-noise inputs
    the classifier's pooled statistics become nearly identical across the batch, which makes
    its train-mode BatchNorm (statistics over B samples) amplify rounding noise by 1/sqrt(var)
    -- an ill-conditioned test problem rather than a property of real speech batches."""
    import numpy as np
    rs = np.random.RandomState(seed)
    t = np.arange(T, dtype=np.float64)[:, None]
    f = np.arange(80, dtype=np.float64)[None, :]
    out = np.zeros((B, T, 80))
    for b in range(B):
        f0 = 6.0 + 9.0 * b + 3.0 * rs.rand()                       # ridge spacing (Mel bins)
        mod = 0.05 + 0.04 * b                                       # temporal modulation rate
        ridges = np.cos(2 * np.pi * f / f0 + 0.7 * np.sin(2 * np.pi * mod * t))
        tilt = -(0.4 + 0.3 * b) * (f / 80.0)
        env = 0.6 + 0.4 * np.sin(2 * np.pi * (0.011 + 0.007 * b) * t + b)
        out[b] = env * ridges + tilt + (0.25 + 0.1 * b) * rs.standard_normal((T, 80))
    return torch.from_numpy(out.astype("float32"))
