"""Oracle: CPU restatement of the reference's ConvReconstruction (models/EndToEnd.py:36-87).

TEST INFRASTRUCTURE (see oracle/__init__.py).  The conv stack (:40-54) and the forward (:64-87)
follow the reference line by line; the pretrained x-vector classifier the reference loads from
absolute paths (``EncoderClassifier.from_hparams(source="/home/ubuntu/...")``, :57-61) is passed
in: ``OracleEncoderClassifier`` wraps oracle/xvector.py (Xvector + Classifier, eval mode) behind
the call the reference makes on it, ``self.sex_classifier(feats) -> (log_probs, score, index)``
(:81; the fork-only feats entry point of speechbrain's EncoderClassifier -- parity unpinned, see
oracle/xvector.py).  gen_golden.py imports the reference class with ``from_hparams`` returning this
same wrapper and asserts bit-equality.
"""
import torch
import torch.nn as nn

from .convae import GLU
from . import xvector as OX


class OracleEncoderClassifier(nn.Module):
    def __init__(self, embedding_model=None, classifier=None):
        super().__init__()
        self.embedding_model = embedding_model or OX.Xvector()
        self.classifier = classifier or OX.Classifier()
        self.eval()
        for p in self.parameters():
            p.requires_grad = False

    def forward(self, feats, wav_lens=None):
        out_prob = self.classifier(self.embedding_model(feats, wav_lens)).squeeze(1)
        score, index = torch.max(out_prob, dim=-1)
        return out_prob, score, index

    classify_batch_feats = forward


class ConvReconstruction(nn.Module):
    def __init__(self, sex_classifier):
        super().__init__()
        self.encoder = nn.Sequential(
            nn.Conv1d(1, 32, 15, 1, 7), nn.InstanceNorm1d(32, affine=True), GLU(),
            nn.Conv1d(32, 64, 5, 2, 2), nn.InstanceNorm1d(64, affine=True), GLU(),
            nn.Conv1d(64, 64, 5, 1, 2), nn.InstanceNorm1d(64, affine=True), GLU(),
            nn.ConvTranspose1d(64, 32, 5, 2, 2, output_padding=1), nn.InstanceNorm1d(32, affine=True), GLU(),
            nn.Conv1d(32, 1, 15, 1, 7),
        )
        self.sex_classifier = sex_classifier

    def forward(self, input):
        out = input
        input = input.reshape(input.shape[0], input.shape[1] * input.shape[2]).unsqueeze(1)
        input = self.encoder(input).squeeze(1)
        input = input.reshape(input.shape[0], out.shape[1], out.shape[2])
        logits, score, index = self.sex_classifier(input)
        return input, logits


def numpy_params(module, seed):
    """deterministic, platform-independent parameters keyed like module.state_dict() (numpy
    RandomState; same conventions as oracle.convae.numpy_params, plus non-trivial BatchNorm
    running statistics for the eval-mode classifier)."""
    import numpy as np
    rs = np.random.RandomState(seed)
    out = {}
    for k, v in module.state_dict().items():
        shp = tuple(v.shape)
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros((), dtype=torch.long)
        elif k.endswith("running_mean"):
            out[k] = torch.from_numpy((0.2 * rs.standard_normal(shp)).astype("float32"))
        elif k.endswith("running_var"):
            out[k] = torch.from_numpy((0.5 + rs.rand(*shp)).astype("float32"))
        elif v.dim() == 1 and k.endswith("weight"):
            out[k] = torch.from_numpy((1.0 + 0.1 * rs.standard_normal(shp)).astype("float32"))
        elif v.dim() == 1:
            out[k] = torch.from_numpy((0.05 * rs.standard_normal(shp)).astype("float32"))
        else:
            fan_in = int(v[0].numel()) if "encoder.9." not in k else int(v.shape[0] * v.shape[2])
            out[k] = torch.from_numpy((rs.standard_normal(shp) / (fan_in ** 0.5)).astype("float32"))
    return out
