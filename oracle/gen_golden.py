"""Generate tests/golden/* from the REFERENCE's own classes (run in the build container,
where /root/reference exists; never on the GPU box).

TEST INFRASTRUCTURE (see oracle/__init__.py).  What it does:
  1. injects a two-symbol speechbrain shim (StatisticsPooling, EncoderClassifier) into
     sys.modules -- the reference's models/ConvAutoEncoder.py imports both at module
     level (:5-6) and speechbrain is an empty submodule / not installed;
  2. imports the reference's ConvAutoencoder, CosineSimilarityLoss, ClusterMI,
     GroupSamplingMI unmodified from /root/reference;
  3. runs them on seeded inputs, asserts the oracle restatement (oracle/convae.py,
     oracle/losses.py) reproduces them (bit-exact for the ConvAE forward/backward),
     and writes the vectors as small .npz/.json fixtures;
  4. reads the data fixtures the reference's results/ hold (normalizer.ckpt via
     torch.load(weights_only=True), train_log.txt lr column, model.ckpt key/shape list).

Usage:  python -m oracle.gen_golden        (from the repo root)
"""
import io
import json
import os
import re
import sys
import types
import contextlib

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
RUN = "sa_channel1_convtranspose_glu_sexclassifier_recon0.1_sex0.9_l1_2_60_epoch_adam_lr_1.0"

from oracle import convae as O, losses as L, train_step as TS       # noqa: E402


def install_shim():
    sb = types.ModuleType("speechbrain")
    nnet = types.ModuleType("speechbrain.nnet")
    pooling = types.ModuleType("speechbrain.nnet.pooling")
    pretrained = types.ModuleType("speechbrain.pretrained")

    class StatisticsPooling(O.StatisticsPooling):          # deterministic form (noise off)
        def __init__(self):
            super().__init__(noise=None)

    class EncoderClassifier:
        """the reference's models/EndToEnd.py builds its frozen gender classifier with
        EncoderClassifier.from_hparams(source="/home/ubuntu/...") (:57-61): absolute paths of its
        authors' machine and weights that are not in the repository.  The shim hands back the
        oracle x-vector wrapper registered in FROM_HPARAMS (seeded weights) instead."""
        FROM_HPARAMS = None

        @classmethod
        def from_hparams(cls, **_kw):
            assert cls.FROM_HPARAMS is not None
            return cls.FROM_HPARAMS

    pooling.StatisticsPooling = StatisticsPooling
    pretrained.EncoderClassifier = EncoderClassifier
    sb.nnet, nnet.pooling, sb.pretrained = nnet, pooling, pretrained
    sys.modules.update({"speechbrain": sb, "speechbrain.nnet": nnet,
                        "speechbrain.nnet.pooling": pooling,
                        "speechbrain.pretrained": pretrained})


def sub(t, n=2048):
    """fixed-stride subsample of a flattened tensor (keeps fixtures small)."""
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].numpy().copy()


def convae_fixture(tag, B, T, seed, recon_kind):
    from models.ConvAutoEncoder import ConvAutoencoder as RefAE
    params = O.numpy_params(8886)
    from oracle.features import synthetic_feats
    rs = np.random.RandomState(seed)
    feats = synthetic_feats(B, T, seed)
    feats[-1, T - 5:] = 0.0                                # a few zero-padded frames
    target = feats + torch.from_numpy(0.1 * rs.standard_normal((B, T, 80)).astype("float32"))
    gender = torch.arange(B) % 2
    w = dict(recon=0.1, sex=0.9, utility=0.0, confusion=0.0)

    def run(model):
        model.load_state_dict(params)
        model.train()
        recon, logp = model(feats)
        rl = L.recon_loss(recon, target, recon_kind)
        sl = torch.nn.NLLLoss()(logp, gender)
        cl = torch.nn.MSELoss()(logp, torch.ones(logp.shape) * (-0.6931))
        loss = w["recon"] * rl + w["sex"] * sl + w["utility"] * 0.0
        loss.backward()
        grads = {k: p.grad.clone() for k, p in model.named_parameters()}
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, betas=(0.9, 0.98), eps=1e-9)
        opt.step()
        return dict(recon=recon.detach(), logp=logp.detach(), rl=rl.detach(), sl=sl.detach(),
                    cl=cl.detach(), loss=loss.detach(), grads=grads, gn=gn,
                    state={k: v.detach().clone() for k, v in model.state_dict().items()})

    ref, ora = run(RefAE()), run(O.ConvAutoencoder())
    for k in ("recon", "logp", "loss", "gn"):
        assert torch.equal(ref[k], ora[k]), f"oracle != reference on {k}"
    for k in ref["grads"]:
        assert torch.equal(ref["grads"][k], ora["grads"][k]), f"oracle != reference grad {k}"
    for k in ref["state"]:
        assert torch.equal(ref["state"][k], ora["state"][k]), f"oracle != reference state {k}"

    d = dict(feats=feats.numpy(), target=target.numpy(), gender=gender.numpy(),
             recon=ref["recon"].numpy(), logp=ref["logp"].numpy(),
             recon_loss=ref["rl"].numpy(), sex_loss=ref["sl"].numpy(),
             confusion_loss=ref["cl"].numpy(), loss=ref["loss"].numpy(),
             grad_norm=ref["gn"].numpy(), recon_kind=np.array(recon_kind))
    for k, g in ref["grads"].items():
        d["grad_sub/" + k] = sub(g)
        d["grad_stat/" + k] = np.array([float(g.double().sum()), float(g.double().norm())])
    for k, v in ref["state"].items():
        if v.dtype.is_floating_point:
            d["state_sub/" + k] = sub(v)
    np.savez_compressed(os.path.join(OUT, f"convae_{tag}.npz"), **d)
    print(f"convae_{tag}: loss={float(ref['loss']):.6f} gn={float(ref['gn']):.6f} "
          f"(oracle == reference: bit-exact)")


def endtoend_fixture(B=3, T=36, seed=4):
    """models/EndToEnd.py:ConvReconstruction (the reference's own class, its classifier =
    the oracle x-vector wrapper via the shim) == oracle/endtoend.py, bit for bit; loss with the
    adversarial signs of speechbrain_convae_train.py:111-121."""
    from oracle import endtoend as OE
    from oracle.features import synthetic_feats
    import speechbrain.pretrained as sbp
    clf = OE.OracleEncoderClassifier()
    clf.load_state_dict(OE.numpy_params(clf, 1230))
    clf.eval()
    sbp.EncoderClassifier.FROM_HPARAMS = clf
    from models.EndToEnd import ConvReconstruction as RefCR
    rs = np.random.RandomState(seed)
    feats = synthetic_feats(B, T, seed)
    target = feats + torch.from_numpy(0.1 * rs.standard_normal((B, T, 80)).astype("float32"))
    gender = torch.arange(B) % 2
    w = dict(recon=0.4, sex=0.5, utility=0.0, confusion=0.1)
    ora0 = OE.ConvReconstruction(clf)
    enc_params = {k: v for k, v in OE.numpy_params(ora0, 8886).items() if k.startswith("encoder.")}

    def run(model):
        model.load_state_dict(enc_params, strict=False)
        model.train()
        model.sex_classifier.eval()
        recon, logp = model(feats)
        rl = L.recon_loss(recon, target, "l1")
        sl = torch.nn.NLLLoss()(logp, gender)
        cl = torch.nn.MSELoss()(logp, torch.ones(logp.shape) * (-0.6931))
        loss = w["recon"] * rl - w["sex"] * sl + w["utility"] * 0.0 - w["confusion"] * cl
        loss.backward()
        grads = {k: p.grad.clone() for k, p in model.named_parameters() if k.startswith("encoder.")}
        return dict(recon=recon.detach(), logp=logp.detach(), loss=loss.detach(), grads=grads)

    ref, ora = run(RefCR()), run(ora0)
    for k in ("recon", "logp", "loss"):
        assert torch.equal(ref[k], ora[k]), f"oracle != reference on {k}"
    for k in ref["grads"]:
        assert torch.equal(ref["grads"][k], ora["grads"][k]), f"oracle != reference grad {k}"
    d = dict(feats=feats.numpy(), target=target.numpy(), gender=gender.numpy(), recon=ref["recon"].numpy(),
             logp=ref["logp"].numpy(), loss=ref["loss"].numpy(),
             weights=np.array([w["recon"], w["sex"], w["utility"], w["confusion"]]))
    for k, g in ref["grads"].items():
        d["grad_sub/" + k] = sub(g)
        d["grad_stat/" + k] = np.array([float(g.double().sum()), float(g.double().norm())])
    np.savez_compressed(os.path.join(OUT, "endtoend_S.npz"), **d)
    print(f"endtoend_S: loss={float(ref['loss']):.6f} (oracle == reference: bit-exact)")


def loss_fixtures():
    from utils.cosine_similarity_loss import CosineSimilarityLoss
    with contextlib.redirect_stdout(io.StringIO()):
        from utils.ClusterMI import ClusterMI
        from utils.GroupSamplingMI import GroupSamplingMI
    rs = np.random.RandomState(7)
    x1 = torch.from_numpy(rs.standard_normal((3, 7, 40)).astype("float32"))
    x2 = x1 + torch.from_numpy(0.5 * rs.standard_normal((3, 7, 40)).astype("float32"))
    x2[0, 0] = 0.0                                          # exercises the eps clamp
    cos_ref = CosineSimilarityLoss()(x1, x2)
    assert torch.allclose(cos_ref, L.cosine_similarity_loss(x1, x2), rtol=0, atol=0)

    X = torch.from_numpy(rs.standard_normal((16, 24)).astype("float32"))
    y = torch.from_numpy(rs.randint(0, 2, 16)).long()
    X[y == 1] += 0.8
    mi_ref = ClusterMI(n_classes=2, k=3)(X, y)
    mi_ora = L.cluster_mi(X, y)
    assert abs(float(mi_ref) - float(mi_ora)) < 1e-6, (mi_ref, mi_ora)
    # GroupSamplingMI: replay the reference's np.random.choice draws
    groups = y.numpy().tolist()
    np.random.seed(123)
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        mi_list, mi_mean, mi_std = GroupSamplingMI(n_samples=5, n_classes=2, n_iterations=6)(X, y, groups)
    np.random.seed(123)
    gs = {g: np.array([i for i, v in enumerate(groups) if v == g]) for g in sorted(set(groups))}
    idx_sets = []
    for _ in range(6):
        idx = []
        for g in gs:
            idx.extend(gs[g][np.random.choice(len(gs[g]), 5, replace=False)])
        idx_sets.append(np.array(idx))
    o_list, o_mean, o_std = L.group_sampling_mi(X, y, [torch.from_numpy(i) for i in idx_sets])
    assert np.allclose([float(v) for v in mi_list], [float(v) for v in o_list], atol=1e-6)
    # second case at the batch size the bench uses (N = 32, D = 40, many exact distance ties: rows
    # are duplicated so that d <= anchor counts are decided by equality, which is index work)
    X32 = torch.from_numpy(rs.standard_normal((32, 40)).astype("float32"))
    y32 = torch.from_numpy(rs.randint(0, 2, 32)).long()
    X32[y32 == 1] += 0.5
    X32[5], X32[17], X32[20] = X32[3].clone(), X32[3].clone(), X32[11].clone()
    y32[5], y32[17], y32[20] = y32[3], y32[3], y32[11]
    mi32_ref = ClusterMI(n_classes=2, k=3)(X32, y32)
    assert abs(float(mi32_ref) - float(L.cluster_mi(X32, y32))) < 1e-6
    groups32 = y32.numpy().tolist()
    np.random.seed(321)
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        mi32_list, _, _ = GroupSamplingMI(n_samples=8, n_classes=2, n_iterations=8)(X32, y32, groups32)
    np.random.seed(321)
    gs32 = {g: np.array([i for i, v in enumerate(groups32) if v == g]) for g in sorted(set(groups32))}
    idx32 = []
    for _ in range(8):
        idx = []
        for g in gs32:
            idx.extend(gs32[g][np.random.choice(len(gs32[g]), 8, replace=False)])
        idx32.append(np.array(idx))
    o32, _, _ = L.group_sampling_mi(X32, y32, [torch.from_numpy(i) for i in idx32])
    assert np.allclose([float(v) for v in mi32_list], [float(v) for v in o32], atol=1e-6)
    np.savez_compressed(os.path.join(OUT, "losses_n32.npz"), X=X32.numpy(), y=y32.numpy(),
                        mi=np.array(float(mi32_ref)), idx_sets=np.stack(idx32),
                        mi_list=np.array([float(v) for v in mi32_list]))
    np.savez_compressed(os.path.join(OUT, "losses.npz"), x1=x1.numpy(), x2=x2.numpy(),
                        cos_loss=cos_ref.numpy(), X=X.numpy(), y=y.numpy(),
                        mi=np.array(float(mi_ref)), idx_sets=np.stack(idx_sets),
                        mi_list=np.array([float(v) for v in mi_list]),
                        mi_mean=np.array(float(mi_mean)), mi_std=np.array(float(mi_std)))
    print(f"losses: cos={float(cos_ref):.6f} mi={float(mi_ref):.6f} (oracle == reference)")


def data_pins():
    pins = {}
    ck = os.path.join(REF, "results", RUN, "8886", "save")
    d = sorted(x for x in os.listdir(ck) if x.startswith("CKPT"))
    norm = torch.load(os.path.join(ck, d[-1], "normalizer.ckpt"), weights_only=True, map_location="cpu")
    pins["normalizer"] = {"source": f"results/{RUN}/8886/save/{d[-1]}/normalizer.ckpt",
                          "keys": sorted(norm.keys()), "count": int(norm["count"]),
                          "glob_mean": [float(v) for v in norm["glob_mean"]],
                          "glob_std": [float(v) for v in norm["glob_std"]]}
    model = torch.load(os.path.join(ck, d[-1], "model.ckpt"), weights_only=True, map_location="cpu")
    pins["historical_model_ckpt"] = {"source": f"results/{RUN}/8886/save/{d[-1]}/model.ckpt",
                                     "shapes": {k: list(v.shape) for k, v in model.items()}}
    cls = torch.load(os.path.join(REF, "results/gender_classifier/1230/save/"
                                  "trained_external_classifier_ckpt/classifier.ckpt"),
                     weights_only=True, map_location="cpu")
    pins["classifier_ckpt"] = {"source": "results/gender_classifier/1230/save/"
                               "trained_external_classifier_ckpt/classifier.ckpt",
                               "shapes": {k: list(v.shape) for k, v in cls.items()}}
    rows = []
    for line in open(os.path.join(REF, "results", RUN, "8886", "train_log.txt")):
        m = re.search(r"lr: ([0-9.e+-]+), steps: (\d+)", line)
        if m:
            rows.append([int(m.group(2)), float(m.group(1))])
    pins["noam_train_log"] = {"source": f"results/{RUN}/8886/train_log.txt", "steps_lr": rows}
    for n, lr in rows:
        assert abs(TS.noam_lr(n) - lr) / lr < 6e-3, (n, lr, TS.noam_lr(n))
    json.dump(pins, open(os.path.join(OUT, "reference_pins.json"), "w"), indent=1)
    print(f"pins: normalizer mean(glob_mean)={np.mean(pins['normalizer']['glob_mean']):.2f}, "
          f"{len(rows)} noam rows reproduce")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    install_shim()
    sys.path.insert(0, REF)
    torch.manual_seed(0)
    torch.set_num_threads(1)        # deterministic CPU reductions
    convae_fixture("S", B=4, T=72, seed=1, recon_kind="l1")
    convae_fixture("S_mse", B=5, T=36, seed=2, recon_kind="mse")
    endtoend_fixture()
    loss_fixtures()
    data_pins()
