"""One small train step of the whole hot path on cuda:0 (Fbank x2 -> normalise -> ConvAE fwd ->
losses -> bwd -> clip -> Adam -> Noam) checked against the CPU oracle's step on the same
waveforms (used by __graft_entry__.smoke() and tests/test_train_step_gpu.py)."""
import functools

import numpy as np
import torch


def rel_mse(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float(((a - b) ** 2).sum() / (b ** 2).sum().clamp_min(1e-30))


def make_wave(B, N, seed=8886):
    """utterances with different harmonic structure (see oracle.features.synthetic_feats)."""
    rs = np.random.RandomState(seed)
    t = np.arange(N) / 16000.0
    w = np.zeros((B, N))
    for b in range(B):
        f0 = 110.0 * (1 + b)
        for h in range(1, 12):
            w[b] += (0.3 / h) * np.sin(2 * np.pi * f0 * h * t * (1 + 0.02 * np.sin(2 * np.pi * (2 + b) * t)))
        w[b] *= 0.5 + 0.4 * np.sin(2 * np.pi * (1.5 + 0.7 * b) * t)
        w[b] += 0.02 * (1 + b) * rs.standard_normal(N)
    return torch.from_numpy(np.clip(w, -1, 1).astype("float32"))


def build(dtype, device, params=None):
    import speech_anonymization_amd as pkg
    from speech_anonymization_amd import brain as B, convae, losses
    kw = dict(precision=dtype) if isinstance(dtype, str) else dict(dtype=dtype)
    model = convae.ConvAutoencoder(pooling_noise=None, **kw)
    if params is not None:
        model.load_state_dict(params)
    hp = dict(model_type="convae", compute_features=pkg.Fbank(16000, 400, 80).to(device),
              epoch_counter=B.EpochCounter(500), loss_reconstruction=losses.L1Loss(),
              loss_sex_classification=losses.NLLLoss(), loss_confusion=losses.ConfusionLoss(),
              recon_loss_weight=0.1, sex_loss_weight=0.9, utility_loss_weight=0.0,
              confusion_loss_weight=0.0, gradient_accumulation=1,
              noam_annealing=B.NoamScheduler(1.0, 25000, 768))
    hp["epoch_counter"].current = 1
    br = B.SexAnonymizationTraining(
        modules={"normalize": pkg.InputNormalization("global", update_until_epoch=4)},
        opt_class=functools.partial(torch.optim.Adam, lr=0.001, betas=(0.9, 0.98), eps=1e-9),
        hparams=hp, run_opts={"device": str(device)})
    br.modules["ConvAE"] = model.to(device)
    br.on_fit_start()
    br.modules.train()
    return br


def run(dtype="bf16x3", B=4, N=11360, steps=2, verbose=True):
    from oracle.convae import numpy_params
    from oracle.train_step import OracleTrainer
    from speech_anonymization_amd.brain import Batch
    dev = torch.device("cuda:0")
    params = numpy_params(8886)
    wav = make_wave(B, N)
    lens = torch.tensor([1.0, 0.83, 0.61, 1.0][:B])
    gender = torch.arange(B) % 2
    ora = OracleTrainer(params=params, threads=8)
    br = build(dtype, dev, params)
    batch = Batch(wav, lens, gender)
    exact = dtype != torch.bfloat16
    tol = {torch.float32: 3e-5, 'bf16x3': 3e-4}.get(dtype, 5e-2)
    p0 = {k: v.clone() for k, v in params.items()}
    for s in range(steps):
        o_loss, aux = ora.fit_batch(wav, lens, gender)
        br.step += 1
        loss = br.fit_batch(batch)
        torch.cuda.synchronize()
        if verbose:
            print(f"step {s}: loss hip {float(loss):.6f} oracle {float(o_loss):.6f}")
        # Adam's first update is -lr*g/(|g|+1e-9): parameters whose gradient is rounding noise
        # (conv biases in front of InstanceNorm) move by +-lr on the sign of that noise, in the
        # reference as well, so from the second step on the two runs agree only to ~lr.
        lim = tol if s == 0 else (2e-2 if exact else 0.25)
        assert abs(float(loss) - float(o_loss)) < lim * max(1.0, abs(float(o_loss))), (s, loss, o_loss)
        if s == 0:
            g0 = aux["grads"]
            hsd = {k: v.detach().cpu() for k, v in br.modules["ConvAE"].state_dict().items()}
            osd = ora.model.state_dict()
            agree = total = 0
            for k, g in g0.items():
                m = g.abs() > (1e-3 if exact else 3e-2) * g.abs().max()
                du_h, du_o = (hsd[k] - p0[k])[m], (osd[k] - p0[k])[m]
                agree += int((torch.sign(du_h) == torch.sign(du_o)).sum())
                total += int(m.sum())
                assert float((hsd[k] - p0[k]).abs().max()) <= 1.001e-3, k     # |Adam step 1| <= lr
            if verbose:
                print(f"first Adam update: {agree}/{total} significant elements move the same way")
            assert agree >= (0.999 if exact else 0.75) * total
    assert br.hparams.noam_annealing.n_steps == steps
    assert abs(br.optimizer.param_groups[0]["lr"] - ora.opt.param_groups[0]["lr"]) < 1e-15
    assert abs(br.optimizer.param_groups[0]["lr"] - 768 ** -0.5 * steps * 25000 ** -1.5) < 1e-15
    nrm = br.modules["normalize"]
    assert nrm.count == ora.normalize.count == 2 * steps      # Fbank+normalise run twice per step
    assert rel_mse(nrm.glob_mean, ora.normalize.glob_mean) < 1e-8
    return float(loss)


if __name__ == "__main__":
    run()
