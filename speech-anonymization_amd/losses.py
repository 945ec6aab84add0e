"""Loss modules with the reference's YAML surface, computed by libsa_hip.so with the gradient
fused into the reduction pass.

  loss_reconstruction: !new:torch.nn.MSELoss / L1Loss   -> ReconLoss("mse" | "l1")
  loss_sex_classification: !new:torch.nn.NLLLoss        -> NLLLoss()
  loss_confusion: !new:torch.nn.MSELoss (vs -0.6931)    -> ConfusionLoss()
  loss_utility: utils.cosine_similarity_loss.CosineSimilarityLoss -> CosineSimilarityLoss()
  loss_mutual_information: !new:utils.mi_loss.MILoss    -> MILoss()
(speechbrain_configs/convae.yaml:78-85, transformer.yaml:71-74; call sites
speechbrain_convae_train.py:105-109.)
"""
import numpy as np
import torch

from . import ops


class _ScalarWithGrad(torch.autograd.Function):
    """loss value computed by a HIP kernel that also produced d loss / d input."""

    @staticmethod
    def forward(ctx, x, loss, grad):
        ctx.save_for_backward(grad)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None


class ReconLoss(torch.nn.Module):
    """nn.L1Loss / nn.MSELoss(reduction="mean") on the flattened [B, T'*80] tensors
    (speechbrain_convae_train.py:105): mean over every element, zero-padded frames included."""

    def __init__(self, kind="mse", reduction="mean"):
        super().__init__()
        assert reduction == "mean" and kind in ("l1", "mse")
        self.kind = kind

    def forward(self, pred, target):
        p = pred.detach().contiguous().float()
        loss, grad = ops.recon_loss(p, target.detach().contiguous().float(), self.kind,
                                    want_grad=pred.requires_grad)
        if not pred.requires_grad:
            return loss.reshape(())
        return _ScalarWithGrad.apply(pred, loss, grad.view_as(pred))


class L1Loss(ReconLoss):
    def __init__(self, reduction="mean"):
        super().__init__("l1", reduction)


class MSELoss(ReconLoss):
    def __init__(self, reduction="mean"):
        super().__init__("mse", reduction)


class NLLLoss(torch.nn.Module):
    def forward(self, logp, label):
        label = torch.as_tensor(label, device=logp.device).long().contiguous()
        out, dn, _ = ops.cls_losses(logp.detach().contiguous(), label, want_grad=logp.requires_grad)
        if not logp.requires_grad:
            return out[0]
        return _ScalarWithGrad.apply(logp, out[0], dn)


class ConfusionLoss(torch.nn.Module):
    """MSELoss(logp, -0.6931 * ones): speechbrain_convae_train.py:108."""

    def forward(self, logp, target=None):
        label = torch.zeros(logp.shape[0], dtype=torch.long, device=logp.device)
        out, _, dc = ops.cls_losses(logp.detach().contiguous(), label, want_grad=logp.requires_grad)
        if not logp.requires_grad:
            return out[1]
        return _ScalarWithGrad.apply(logp, out[1], dc)


class CosineSimilarityLoss(torch.nn.Module):
    """sum(1 - cos(x1, x2; dim=2, eps=1e-6)) / S  (utils/cosine_similarity_loss.py:53-56).
    Differentiable w.r.t. input1 (the reconstruction branch); input2 is the frozen target."""

    def forward(self, input1, input2):
        x1 = input1.detach().contiguous().float()
        loss, dx1 = ops.cosine_loss(x1, input2.detach().contiguous().float(),
                                    want_grad=input1.requires_grad)
        if not input1.requires_grad:
            return loss.reshape(())
        return _ScalarWithGrad.apply(input1, loss, dx1.view_as(input1))


class MILoss(torch.nn.Module):
    """utils/mi_loss.py:14-17 -> GroupSamplingMI(n_samples = batch_size // sets, 100 iterations,
    k = 3) -> ClusterMI.  Returns the list of per-iteration MI estimates like the reference
    (the estimator has no gradient).  The class-balanced resampling draws come from
    np.random.choice exactly as utils/GroupSamplingMI.py:21-26 does."""

    def __init__(self, n_iterations=100, k=3):
        super().__init__()
        self.n_iterations, self.k = n_iterations, k

    @staticmethod
    def sample_index_sets(groups, n_samples, n_iterations):
        groups = np.asarray(groups)
        members = {g: np.nonzero(groups == g)[0] for g in sorted(set(groups.tolist()))}
        sets = []
        for _ in range(n_iterations):
            idx = []
            for g in members:
                idx.extend(members[g][np.random.choice(len(members[g]), n_samples, replace=False)])
            sets.append(idx)
        return np.asarray(sets, dtype=np.int64)

    def forward(self, X, y, batch, batch_size, n_classes=2, samples_set_per_batch=1):
        n_samples = batch_size // samples_set_per_batch
        idx = self.sample_index_sets(batch, n_samples, self.n_iterations)
        Xf = X.detach().reshape(X.shape[0], -1).contiguous().float()
        mi = ops.cluster_mi(Xf, y.long().contiguous(), torch.from_numpy(idx).to(X.device),
                            ncls=n_classes, k=self.k)
        return list(mi)
