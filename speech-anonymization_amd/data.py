"""Data side of the hooks: the LibriSpeech CSV manifest the reference consumes (columns
``ID,duration,wav,spk_id,wrd`` + the fork-added ``gender``; speechbrain_convae_train.py:419-511),
16-bit PCM WAV reading (stdlib ``wave``; soundfile / torchaudio are not dependencies), duration
sorting, gender M/F -> 0/1 (:465-472) and zero-padded batches with relative lengths like
speechbrain's PaddedBatch."""
import csv
import wave

import numpy as np
import torch

from .brain import Batch

SEX = {"M": 0, "F": 1}


def read_audio(path):
    with wave.open(path, "rb") as w:
        assert w.getsampwidth() == 2, "16-bit PCM WAV expected"
        x = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").astype(np.float32) / 32768.0
        if w.getnchannels() > 1:
            x = x.reshape(-1, w.getnchannels()).mean(axis=1)
    return torch.from_numpy(x)


def write_audio(path, sig, sample_rate=16000):
    with wave.open(path, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(sample_rate)
        w.writeframes((sig.clamp(-1, 1) * 32767.0).round().to(torch.int16).numpy().tobytes())


class CsvDataset:
    def __init__(self, csv_path, replacements=None, sorting="random"):
        self.items = []
        for row in csv.DictReader(open(csv_path)):
            wav = row["wav"]
            for k, v in (replacements or {}).items():
                wav = wav.replace("$" + k, v).replace("{" + k + "}", v)
            g = row.get("gender", "M")
            self.items.append(dict(id=row["ID"], duration=float(row["duration"]), wav=wav,
                                   wrd=row.get("wrd", ""), gender=SEX.get(g, g)))
        if sorting in ("ascending", "descending"):
            self.items.sort(key=lambda r: r["duration"], reverse=sorting == "descending")
        elif sorting != "random":
            raise NotImplementedError("sorting must be random, ascending or descending")

    def __len__(self):
        return len(self.items)


def shard_indices(n_items, shuffle=False, seed=0, epoch=0, rank=0, world=1):
    """Index list of one rank for one epoch, torch DistributedSampler semantics: the (optionally
    shuffled, seed + epoch like speechbrain's ReproducibleRandomSampler) list is padded by
    wrapping around to a multiple of `world`, then dealt round-robin -- every rank gets the SAME
    number of items, hence the same number of batches (a rank with one batch more would wait
    forever in the SyncBatchNorm / gradient all-reduce of a step its peers never run)."""
    idx = list(range(n_items))
    if shuffle:
        np.random.RandomState(seed + epoch).shuffle(idx)
    if world > 1 and idx:
        total = -(-len(idx) // world) * world
        idx = (idx * (total // len(idx) + 1))[:total]
    return idx[rank::world]


def batches(dataset, batch_size, shuffle=False, seed=0, rank=0, world=1, epoch=0):
    """yields Batch objects; utterances are sharded by index across data-parallel ranks."""
    idx = shard_indices(len(dataset), shuffle, seed, epoch, rank, world)
    for i in range(0, len(idx), batch_size):
        rows = [dataset.items[j] for j in idx[i:i + batch_size]]
        sigs = [read_audio(r["wav"]) for r in rows]
        n = max(len(s) for s in sigs)
        wav = torch.zeros(len(sigs), n)
        for k, s in enumerate(sigs):
            wav[k, :len(s)] = s
        lens = torch.tensor([len(s) / n for s in sigs])
        yield Batch(wav, lens, torch.tensor([int(r["gender"]) for r in rows]), ids=[r["id"] for r in rows])


def synthetic_dataset(n_utts, batch_size, n_samples=161120, seed=8886, rank=0, world=1):
    """SURVEY.md 8(d) synthetic waveforms, generated on the fly (no dataset on the GPU box)."""
    g = torch.Generator().manual_seed(seed + rank)
    t = torch.arange(n_samples, dtype=torch.float64) / 16000.0
    for i in range(0, n_utts // world, batch_size):
        b = min(batch_size, n_utts // world - i)
        w = 0.1 * torch.randn(b, n_samples, generator=g, dtype=torch.float64)
        for f, a in ((220.0, 0.2), (1000.0, 0.1), (3400.0, 0.05)):
            w += a * torch.sin(2 * torch.pi * f * t)[None, :]
        yield Batch(w.clamp(-1, 1).float(), torch.ones(b), torch.arange(b) % 2)
