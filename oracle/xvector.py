"""Oracle: CPU restatement of the x-vector gender classifier FORWARD.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows models/external_gender_classifiers.py
(:24-115 Xvector, :118-183 Classifier; configured by speechbrain_configs/evaluator_inference.yaml
:34-48: in 80, five TDNN blocks 512,512,512,512,1500 / k 5,3,3,1,1 / dil 1,2,3,1,1, LeakyReLU,
emb 128, 2 classes).  The layer wrappers are speechbrain's (un-vendored, restated from the
published v0.5.x code): nnet.CNN.Conv1d ([B,T,C] layout, padding "same" with reflect mode),
nnet.normalization.BatchNorm1d, nnet.linear.Linear, nnet.pooling.StatisticsPooling(lengths).
"parity unpinned" for the arithmetic; parameter names / shapes are pinned by the reference's
classifier.ckpt (tests/golden/reference_pins.json: norm.norm.weight, DNN.block_0.linear.w.weight,
out.w.weight ...).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .convae import StatisticsPooling


class Conv1d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, dilation=1):
        super().__init__()
        self.kernel_size, self.dilation = kernel_size, dilation
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size, dilation=dilation)

    def forward(self, x):
        x = x.transpose(1, -1)
        L_in = x.shape[-1]
        L_out = (L_in - self.dilation * (self.kernel_size - 1) - 1) + 1
        p = (L_in - L_out) // 2
        x = F.pad(x, (p, p), mode="reflect")
        return self.conv(x).transpose(1, -1)


class BatchNorm1d(nn.Module):
    def __init__(self, input_size):
        super().__init__()
        self.norm = nn.BatchNorm1d(input_size)

    def forward(self, x):
        return self.norm(x.transpose(-1, 1)).transpose(1, -1) if x.dim() == 3 else self.norm(x)


class Linear(nn.Module):
    def __init__(self, input_size, n_neurons):
        super().__init__()
        self.w = nn.Linear(input_size, n_neurons)

    def forward(self, x):
        return self.w(x)


class Xvector(nn.Module):
    def __init__(self, in_channels=80, lin_neurons=128, tdnn_channels=(512, 512, 512, 512, 1500),
                 tdnn_kernel_sizes=(5, 3, 3, 1, 1), tdnn_dilations=(1, 2, 3, 1, 1), pooling_noise=None):
        super().__init__()
        self.blocks = nn.ModuleList()
        for c, k, d in zip(tdnn_channels, tdnn_kernel_sizes, tdnn_dilations):
            self.blocks.extend([Conv1d(in_channels, c, k, d), nn.LeakyReLU(), BatchNorm1d(c)])
            in_channels = c
        self.blocks.append(StatisticsPooling(pooling_noise))
        self.blocks.append(Linear(in_channels * 2, lin_neurons))

    def forward(self, x, lens=None):
        for layer in self.blocks:
            x = layer(x, lengths=lens) if isinstance(layer, StatisticsPooling) else layer(x)
        return x


class _Block(nn.Module):
    def __init__(self, n_in, n_out):
        super().__init__()
        self.linear, self.act, self.norm = Linear(n_in, n_out), nn.LeakyReLU(), BatchNorm1d(n_out)

    def forward(self, x):
        return self.norm(self.act(self.linear(x)))


class Classifier(nn.Module):
    def __init__(self, emb=128, lin_neurons=128, out_neurons=2):
        super().__init__()
        self.act = nn.LeakyReLU()
        self.norm = BatchNorm1d(emb)
        self.DNN = nn.ModuleDict({"block_0": _Block(emb, lin_neurons)})
        self.out = Linear(lin_neurons, out_neurons)

    def forward(self, x):
        x = self.norm(self.act(x))
        x = self.DNN["block_0"](x)
        return F.log_softmax(self.out(x), dim=-1)
