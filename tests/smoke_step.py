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
        # Adam's first update is -lr*g/(|g|+1e-9), i.e. sign-like with |step| = lr = 1e-3 (the Noam
        # rate only applies from the second step on): every element whose gradient is small
        # relative to the rounding noise of its own computation moves by a full +-lr on the sign
        # of that noise, in the reference as well, so free-running trajectories agree only to
        # ~lr from the second step on.  (Conv biases in front of InstanceNorm are NOT the cause:
        # the norm removes them, they cannot move the loss.)  run_teacher_forced() below pins the
        # later steps tightly by restarting every step from the oracle's state.
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


def hip_step(br, batch):
    """SexAnonymizationTraining.fit_batch spelled out so that the gradients can be looked at
    before optimizer.zero_grad() drops them (same calls, same order)."""
    from speech_anonymization_amd.brain import Stage
    br.step += 1
    br.apply_epoch_schedule()
    out = br.compute_forward(batch, Stage.TRAIN)
    loss = br.compute_objectives(out, batch, Stage.TRAIN)
    (loss / br.hparams.gradient_accumulation).backward()
    torch.cuda.synchronize()
    if getattr(br.hparams, "epoch_parity_schedule", False):
        br._zero_grads_of_frozen()        # torch 1.10: a frozen parameter keeps a zero-filled .grad
    grads = {k: (p.grad.detach().clone() if p.grad is not None else None)
             for k, p in br.modules["ConvAE"].named_parameters()}
    br.check_gradients(loss)
    br.optimizer.step()
    br.optimizer.zero_grad()
    br.hparams.noam_annealing(br.optimizer)
    return loss.detach(), grads


def load_oracle_state(br, ora):
    """teacher forcing: parameters, BatchNorm buffers, Adam moments / step counts and the
    normaliser's running statistics of the oracle into the HIP brain."""
    br.modules["ConvAE"].load_state_dict(ora.model.state_dict())
    br.optimizer.load_state_dict(ora.opt.state_dict())
    br.modules["normalize"].load_state_dict(dict(count=ora.normalize.count, glob_mean=ora.normalize.glob_mean,
                                                 glob_std=ora.normalize.glob_std))


def compare_grads(h, o, tol, exact, tag, tol_cls=None):
    """tol: gradients that do not pass through the classifier (decoder); tol_cls: the rest (the
    classifier's train-mode BatchNorm over a handful of utterances amplifies rounding noise ~1e3x
    in everything upstream of it, in the reference's own fp32 arithmetic too)."""
    from tests.test_convae_gpu import NULL_BIAS
    worst = 0.0
    for k, g in o.items():
        if g is None:
            assert h[k] is None, (tag, k, "no gradient in the oracle (frozen / not in the graph), has one here")
            continue
        assert h[k] is not None, (tag, k)
        if k in NULL_BIAS:
            if o[NULL_BIAS[k]] is not None:
                lim = (1e-3 if exact else 3e-2) * float(o[NULL_BIAS[k]].abs().max())
                assert float(h[k].abs().max()) <= lim, (tag, k)       # both exactly 0 when nothing flows
            continue
        e = rel_mse(h[k], g)
        lim = tol if k.startswith("decoder") or tol_cls is None else tol_cls
        if e >= lim and k.endswith(".bias") and o.get(k[:-4] + "weight") is not None:
            # a bias in front of [ReLU ->] BatchNorm / InstanceNorm has a gradient that cancels
            # over the batch wherever the ReLU is open (exactly, for the norm-only layers listed
            # in NULL_BIAS): what is left can be mostly rounding noise, in the oracle too -- such a
            # bias is held to the scale of its layer's weight gradient instead of to itself
            scale = float(o[k[:-4] + "weight"].abs().max())
            assert float((h[k].cpu() - g).abs().max()) <= 1e-4 * scale, (tag, k, e, lim)
            continue
        worst = max(worst, e)
        assert e < lim, (tag, k, e, lim)
    return worst


def run_teacher_forced(dtype="bf16x3", B=8, N=11360, steps=4, verbose=True, model_type="convae",
                       weights=None, epoch_parity_schedule=False, epochs=None):
    """Steps 1..steps, each started from the ORACLE's state of the previous step (parameters,
    Adam moments, BatchNorm buffers, normaliser): every step's loss and all 56 gradients are
    compared at the first-step tolerance, so the trajectory is pinned step by step without the
    sign-flip divergence of a free-running comparison.  `weights` = dict(recon, sex, utility,
    confusion) and model_type="endtoend" exercise the adversarial-sign loss
    (speechbrain_convae_train.py:111-121); epoch_parity_schedule with epochs=[...] (one entry per
    step) the odd-epoch "classifier only" / even-epoch "classifier frozen" halves (:212-235)."""
    from oracle.convae import numpy_params
    from oracle.train_step import OracleTrainer
    from speech_anonymization_amd.brain import Batch
    dev = torch.device("cuda:0")
    params = numpy_params(8886)
    wav = make_wave(B, N)
    lens = torch.tensor([1.0, 0.83, 0.61, 1.0, 0.9, 0.75, 1.0, 0.66][:B])
    gender = torch.arange(B) % 2
    w = dict(recon=0.1, sex=0.9, utility=0.0, confusion=0.0)
    w.update(weights or {})
    ora = OracleTrainer(params=params, threads=8, recon_w=w["recon"], sex_w=w["sex"], util_w=w["utility"],
                        conf_w=w["confusion"], model_type=model_type,
                        epoch_parity_schedule=epoch_parity_schedule)
    br = build(dtype, dev, params)
    hp = br.hparams
    hp.model_type, hp.epoch_parity_schedule = model_type, epoch_parity_schedule
    hp.recon_loss_weight, hp.sex_loss_weight = w["recon"], w["sex"]
    hp.utility_loss_weight, hp.confusion_loss_weight = w["utility"], w["confusion"]
    batch = Batch(wav, lens, gender)
    exact = dtype != torch.bfloat16
    ltol = {torch.float32: 3e-5, 'bf16x3': 3e-4}.get(dtype, 5e-2)
    # both precisions are held to north_star's 1e-4 against the fp32 oracle here: the oracle's own
    # fp32 result carries ~1e-5 of rounding noise on the classifier-branch gradients (its
    # fp32-vs-fp64 distance, tests/test_convae_gpu.py), which test_against_oracle_full_tensors
    # subtracts by comparing the f32 mode with the fp64 oracle instead
    gtol = {torch.float32: 1e-4, 'bf16x3': 1e-4}.get(dtype, None)
    for s in range(steps):
        if epochs is not None:
            ora.epoch = hp.epoch_counter.current = epochs[s]
        o_loss, aux = ora.fit_batch(wav, lens, gender)
        loss, grads = hip_step(br, batch)
        if verbose:
            print(f"teacher-forced step {s}: loss hip {float(loss):.6f} oracle {float(o_loss):.6f}")
        assert abs(float(loss) - float(o_loss)) < ltol * max(1.0, abs(float(o_loss))), (s, loss, o_loss)
        if gtol is not None:
            # north_star's 1e-4 on everything that does not pass through the classifier; 2e-4 on
            # the classifier branch at this tiny shape (8 utterances x 72 frames; at the benchmark
            # shapes the same gradients are held to 1e-4: test_against_oracle_full_tensors)
            worst = compare_grads(grads, aux["grads"], gtol, exact, f"step {s}", tol_cls=2 * gtol)
            if verbose:
                print(f"   worst gradient rel-MSE {worst:.3e} (limit {gtol:.0e})")
        # the update itself: same Adam on (nearly) the same gradient from the same state
        hsd = br.modules["ConvAE"].state_dict()
        for k, v in ora.model.state_dict().items():
            if v.dtype.is_floating_point and "running" not in k:
                # |Adam update| <= lr on both sides, + one fp32 ulp of the parameter itself (the
                # Noam rates of the first steps, ~1e-8, are below the ulp of a weight of size 0.3)
                slack = 2.001 * max(ora_lr(ora, s), 1e-12) + 2.0 ** -22 * float(v.abs().max())
                assert float((hsd[k].cpu() - v).abs().max()) <= slack, (s, k)
        load_oracle_state(br, ora)
    return float(loss)


def ora_lr(ora, s):
    """|Adam update| <= lr of that step: 1e-3 on the first (before Noam), the Noam rate after"""
    from oracle.train_step import noam_lr
    return 1e-3 if s == 0 else noam_lr(s)


if __name__ == "__main__":
    run()
    run_teacher_forced()
