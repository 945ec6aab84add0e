"""Oracle: CPU restatement of the loss reductions.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Reference:
  * recon / NLL / confusion / weighted sum : speechbrain_convae_train.py:105-128,
    speechbrain_configs/convae.yaml:78-85
  * CosineSimilarityLoss                   : utils/cosine_similarity_loss.py:48,53-56
  * ClusterMI / GroupSamplingMI / MILoss   : utils/ClusterMI.py:12-65,88-121,
    utils/GroupSamplingMI.py:49-61, utils/mi_loss.py:14-17
"""
import math
import torch
import torch.nn.functional as F

LOG_HALF = -0.6931          # the literal the reference uses (speechbrain_convae_train.py:108)


def recon_loss(recon, feats, kind="l1"):
    """loss_reconstruction(recon.view(B,-1), feats.view(B,-1)), mean over B*T'*80 with the
    zero-padded frames included (speechbrain_convae_train.py:105)."""
    a, b = recon.reshape(recon.shape[0], -1), feats.reshape(feats.shape[0], -1)
    return F.l1_loss(a, b) if kind == "l1" else F.mse_loss(a, b)


def sex_loss(logp, gender):
    return F.nll_loss(logp, gender)


def confusion_loss(logp):
    return F.mse_loss(logp, torch.ones_like(logp) * LOG_HALF)


def total_loss(recon_l, sex_l, util_l, conf_l, w, model_type="convae"):
    """speechbrain_convae_train.py:111-128.  w = dict(recon, sex, utility, confusion)."""
    if model_type == "endtoend":
        if w["recon"] == 0.0 and w["utility"] == 0.0:
            return w["sex"] * sex_l
        return (w["recon"] * recon_l - w["sex"] * sex_l + w["utility"] * util_l
                - w["confusion"] * conf_l)
    return w["recon"] * recon_l + w["sex"] * sex_l + w["utility"] * util_l


def cosine_similarity_loss(x1, x2):
    """sum(1 - cos(x1, x2; dim=2, eps=1e-6)) / S   -- divides by S = shape[1], not B*S."""
    sim = F.cosine_similarity(x1, x2, dim=2, eps=1e-6)
    loss = 1 - sim
    return torch.sum(loss) / loss.shape[1]


def pairwise_cosine_dists(x):
    """What utils/ClusterMI.py:_pairwise_dists produces with cosine_distance_2d: a symmetric
    N x N matrix d[i,j] = 1 - cos(x_i, x_j) with a zero diagonal (the reference fills it
    with N/2 rolls; the values are those of F.cosine_similarity(dim=1, eps=1e-8))."""
    N = x.shape[0]
    d = torch.zeros(N, N)
    for i in range(N):
        for j in range(N):
            if i != j:
                d[i, j] = 1 - F.cosine_similarity(x[i:i + 1], x[j:j + 1], dim=1)[0]
    return d


def cluster_mi(X, y, n_classes=2, k=3):
    """utils/ClusterMI.py:88-121 (Ross 2014 k-NN MI between continuous X and labels y),
    in bits.  Non-differentiable by construction (counts + digamma)."""
    N = X.shape[0]
    N_dig = torch.digamma(torch.tensor(float(N)))
    N_x = torch.tensor([float(torch.sum(y == i)) for i in range(n_classes)])
    avg_N_x = torch.sum(N_x / N * torch.digamma(N_x))
    d = pairwise_cosine_dists(X)
    same = y.view(-1, 1) == y.view(1, -1)
    d_same = torch.where(same, d, 10e6 * torch.ones_like(d))
    anchor = torch.topk(d_same, k + 1, dim=1, largest=False)[0][:, -1]
    m_i = torch.sum(d <= anchor.unsqueeze(1), dim=1) - 1
    avg_m = torch.mean(torch.digamma(m_i.float()))
    mi = N_dig - avg_N_x + torch.digamma(torch.tensor(float(k))) - avg_m
    return mi / math.log(2.0)


def group_sampling_mi(X, y, idx_sets, n_classes=2, k=3):
    """utils/GroupSamplingMI.py:49-61 with the resampled index sets passed in explicitly
    (the reference draws them with np.random.choice; a test feeds both sides the same
    draws).  Returns (list of MI values, mean, unbiased std)."""
    mi = [cluster_mi(X[i], y[i], n_classes, k) for i in idx_sets]
    t = torch.tensor([float(v) for v in mi])
    return mi, t.mean(), t.std()
