"""CPU: the oracle against the golden vectors generated from the reference's own classes
(oracle/gen_golden.py) and against the data fixtures the reference's results/ hold."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import convae as O, features as OF, losses as L, train_step as TS


@pytest.mark.parametrize("tag", ["S", "S_mse"])
def test_convae_matches_reference_vectors(golden_dir, tag):
    torch.set_num_threads(1)
    z = np.load(os.path.join(golden_dir, f"convae_{tag}.npz"))
    m = O.ConvAutoencoder()
    m.load_state_dict(O.numpy_params(8886))
    m.train()
    feats, target = torch.from_numpy(z["feats"]), torch.from_numpy(z["target"])
    gender = torch.from_numpy(z["gender"])
    recon, logp = m(feats)
    assert torch.allclose(recon, torch.from_numpy(z["recon"]), rtol=1e-5, atol=1e-6)
    assert torch.allclose(logp, torch.from_numpy(z["logp"]), rtol=1e-5, atol=1e-6)
    rl = L.recon_loss(recon, target, str(z["recon_kind"]))
    sl = L.sex_loss(logp, gender)
    loss = L.total_loss(rl, sl, 0.0, L.confusion_loss(logp), dict(recon=0.1, sex=0.9, utility=0.0, confusion=0.0))
    assert abs(float(loss) - float(z["loss"])) < 1e-6
    assert abs(float(L.confusion_loss(logp)) - float(z["confusion_loss"])) < 1e-6
    loss.backward()
    for k, p in m.named_parameters():
        g = p.grad.reshape(-1)
        step = max(1, g.numel() // 2048)
        assert torch.allclose(g[::step][:2048], torch.from_numpy(z["grad_sub/" + k]), rtol=1e-4, atol=1e-7), k
        s, n = z["grad_stat/" + k]
        assert abs(float(g.double().norm()) - n) <= 1e-5 * max(1.0, n), k


def test_losses_match_reference_vectors(golden_dir):
    z = np.load(os.path.join(golden_dir, "losses.npz"))
    c = L.cosine_similarity_loss(torch.from_numpy(z["x1"]), torch.from_numpy(z["x2"]))
    assert abs(float(c) - float(z["cos_loss"])) < 1e-6
    X, y = torch.from_numpy(z["X"]), torch.from_numpy(z["y"])
    assert abs(float(L.cluster_mi(X, y)) - float(z["mi"])) < 1e-6
    lst, mean, std = L.group_sampling_mi(X, y, [torch.from_numpy(i) for i in z["idx_sets"]])
    assert np.allclose([float(v) for v in lst], z["mi_list"], atol=1e-6)
    assert abs(float(mean) - float(z["mi_mean"])) < 1e-6 and abs(float(std) - float(z["mi_std"])) < 1e-6


def test_mi_and_endtoend_oracle_match_reference_vectors(golden_dir):
    """the round-2 fixtures: N = 32 MI case with exact distance ties; ConvReconstruction
    (models/EndToEnd.py) with the seeded oracle x-vector as its frozen classifier"""
    z = np.load(os.path.join(golden_dir, "losses_n32.npz"))
    X, y = torch.from_numpy(z["X"]), torch.from_numpy(z["y"])
    assert abs(float(L.cluster_mi(X, y)) - float(z["mi"])) < 1e-6
    lst, _, _ = L.group_sampling_mi(X, y, [torch.from_numpy(i) for i in z["idx_sets"]])
    assert np.allclose([float(v) for v in lst], z["mi_list"], atol=1e-6)
    from oracle import endtoend as OE
    torch.set_num_threads(1)
    z = np.load(os.path.join(golden_dir, "endtoend_S.npz"))
    clf = OE.OracleEncoderClassifier()
    clf.load_state_dict(OE.numpy_params(clf, 1230))
    clf.eval()
    m = OE.ConvReconstruction(clf)
    m.load_state_dict({k: v for k, v in OE.numpy_params(m, 8886).items() if k.startswith("encoder.")}, strict=False)
    m.train(); clf.eval()
    recon, logp = m(torch.from_numpy(z["feats"]))
    w = [float(v) for v in z["weights"]]
    loss = (w[0] * L.recon_loss(recon, torch.from_numpy(z["target"]), "l1") - w[1] * L.sex_loss(logp, torch.from_numpy(z["gender"]))
            - w[3] * L.confusion_loss(logp))
    assert torch.equal(recon.detach(), torch.from_numpy(z["recon"])) and abs(float(loss) - float(z["loss"])) < 1e-6
    loss.backward()
    g = m.encoder[3].weight.grad.reshape(-1)
    step = max(1, g.numel() // 2048)
    assert np.allclose(g[::step][:2048].numpy(), z["grad_sub/encoder.3.weight"], rtol=1e-5, atol=1e-8)


def test_noam_schedule_matches_train_log(golden_dir):
    pins = json.load(open(os.path.join(golden_dir, "reference_pins.json")))
    rows = pins["noam_train_log"]["steps_lr"]
    assert len(rows) >= 20
    for n, lr in rows:
        assert abs(TS.noam_lr(n) - lr) / lr < 6e-3           # the log prints 3 significant digits
    assert abs(TS.noam_lr(1) - 9.13e-9) / 9.13e-9 < 2e-3     # "lr: 9.13e-09, steps: 2" row


def test_stft_against_direct_dft():
    wav = OF.synthetic_wave(2, 4000, seed=3)
    fb = OF.Fbank()
    ps = fb.power_spectrum(wav).double()
    ref = OF.power_spectrum_direct_dft(wav)
    assert ps.shape == ref.shape == (2, 26, 201)
    assert float(((ps - ref) ** 2).sum() / (ref ** 2).sum()) < 1e-10


def test_fbank_scale_against_normalizer_pin(golden_dir):
    """statistical pin: the reference's normalizer.ckpt says LibriSpeech Fbank features average
    -24 dB with a 14 dB spread; speech-like synthetic audio must land on the same dB scale."""
    pins = json.load(open(os.path.join(golden_dir, "reference_pins.json")))["normalizer"]
    assert pins["keys"] == ["count", "glob_mean", "glob_std", "spk_dict_count", "spk_dict_mean", "spk_dict_std"]
    assert len(pins["glob_mean"]) == 80 and -30 < np.mean(pins["glob_mean"]) < -18
    f = OF.Fbank()(OF.synthetic_wave(2, 16000, seed=1) * 0.05)
    assert f.shape == (2, 101, 80) and -80 < float(f.mean()) < 10
    n = OF.InputNormalization(update_until_epoch=4)
    out = n(f, torch.tensor([1.0, 0.5]), epoch=1)
    assert n.count == 1 and sorted(n.state_dict().keys()) == pins["keys"]
    assert abs(float(out[0].mean())) < 1.0


def test_statistics_pooling_noise_range():
    p = O.StatisticsPooling(noise="random")
    x = torch.randn(3, 50, 8)
    d = p(x).squeeze(1)[:, :8] - x.mean(1)
    assert float(d.min()) >= 0.99e-5 and float(d.max()) <= 9.01e-5
