"""Whole-step parity on the GPU (Brain hooks -> HIP library) against the oracle's step."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_train_step_fp32_matches_oracle():
    from tests import smoke_step
    smoke_step.run(torch.float32)


def test_train_step_bf16x3_matches_oracle():
    from tests import smoke_step
    smoke_step.run("bf16x3")


def test_train_step_bf16_runs_and_tracks_oracle():
    from tests import smoke_step
    smoke_step.run(torch.bfloat16)


@pytest.mark.parametrize("dtype", [torch.float32, "bf16x3"], ids=["f32", "bf16x3"])
def test_teacher_forced_four_steps(dtype):
    """steps 2-4 pinned at the step-1 tolerance (loss 3e-5 / 3e-4, every gradient 2e-5 / 1e-4)"""
    from tests import smoke_step
    smoke_step.run_teacher_forced(dtype, steps=4)


@pytest.mark.parametrize("dtype", [torch.float32, "bf16x3"], ids=["f32", "bf16x3"])
def test_endtoend_adversarial_sign_branch(dtype):
    """model_type endtoend: recon_w*recon - sex_w*sex + util_w*util - conf_w*confusion
    (speechbrain_convae_train.py:111-121), including a non-zero confusion weight"""
    from tests import smoke_step
    smoke_step.run_teacher_forced(dtype, steps=2, model_type="endtoend",
                                  weights=dict(recon=0.3, sex=0.6, utility=0.0, confusion=0.2))
    # and its "sex only" sub-branch (:112-113)
    smoke_step.run_teacher_forced(dtype, steps=1, model_type="endtoend",
                                  weights=dict(recon=0.0, sex=0.7, utility=0.0, confusion=0.2))


@pytest.mark.parametrize("dtype", [torch.float32, "bf16x3"], ids=["f32", "bf16x3"])
def test_epoch_parity_schedule_both_halves(dtype):
    """HEAD's schedule (speechbrain_convae_train.py:212-235): odd epoch = only the sex classifier
    trains (sex 0.5; encoder + decoder frozen: the backward stops at the classifier input), even
    epoch = classifier frozen, encoder/decoder train against it (sex 0.8).  The sequence
    odd, odd, even, odd also exercises torch 1.10's zero-filled gradients of frozen parameters
    (Adam keeps moving them on their moments)."""
    from tests import smoke_step
    smoke_step.run_teacher_forced(dtype, steps=4, epoch_parity_schedule=True, epochs=[1, 1, 2, 3])


def test_hip_graph_step_equals_eager_steps():
    """run_opts hip_graph=True: three eager steps, capture, then replays -- against the same steps
    launched eagerly (same kernels in the same order; Adam differs only in where its bias
    corrections are evaluated: on the device in the capturable form)."""
    from oracle.convae import numpy_params
    from tests import smoke_step
    from speech_anonymization_amd.brain import Batch
    dev = torch.device("cuda:0")
    wav = smoke_step.make_wave(4, 11360)
    batches = [Batch(wav * s, torch.tensor([1.0, 0.83, 0.61, 1.0]), torch.arange(4) % 2) for s in (1.0, 0.9, 0.8, 1.1, 0.7, 1.05, 0.95)]
    runs = []
    for graph in (False, True):
        br = smoke_step.build("bf16x3", dev, numpy_params(8886))
        if graph:
            import functools
            br.hip_graph, br.optimizer = True, None
            br.init_optimizers()
            assert torch.is_tensor(br.optimizer.param_groups[0]["lr"])
        losses = []
        for b in batches:
            br.step += 1
            losses.append(float(br.fit_batch(b)))
        torch.cuda.synchronize()
        if graph:
            assert len(br._graphs) == 1 and "graph" in next(iter(br._graphs.values()))
        runs.append((losses, {k: v.detach().clone() for k, v in br.modules["ConvAE"].state_dict().items()},
                     br.hparams.noam_annealing.n_steps, br.modules["normalize"].count))
    (l0, p0, n0, c0), (l1, p1, n1, c1) = runs
    assert n0 == n1 == len(batches) and c0 == c1
    for a, b in zip(l0, l1):
        assert abs(a - b) <= 1e-5 * max(1.0, abs(a)), (l0, l1)
    for k in p0:
        if p0[k].dtype.is_floating_point:
            assert float((p0[k] - p1[k]).abs().max()) <= 2e-6 + 1e-5 * float(p0[k].abs().max()), k


def test_hip_graph_follows_the_epoch_parity_schedule():
    """hipGraph mode over epochs 0, 1, 2 of HEAD's epoch-parity schedule
    (speechbrain_convae_train.py:212-235).  Epoch 0 freezes the classifier while it has no Adam
    moments (it stays out of the update); epoch 1 trains it; in epoch 2 it is frozen again but now
    HAS moments, and torch 1.10's zero-filled gradients keep moving it -- the same requires_grad
    pattern as epoch 0 with a different step.  The graph key carries the optimizer state of the
    frozen parameters, so epoch 2 captures its own graph; parameters equal the eager run's."""
    from oracle.convae import numpy_params
    from tests import smoke_step
    from speech_anonymization_amd.brain import Batch
    dev = torch.device("cuda:0")
    wav = smoke_step.make_wave(4, 11360)
    scales = (1.0, 0.9, 0.8, 1.1, 0.7)
    runs = []
    for graph in (False, True):
        br = smoke_step.build("bf16x3", dev, numpy_params(8886))
        br.hparams.epoch_parity_schedule = True
        if graph:
            br.hip_graph, br.optimizer = True, None
            br.init_optimizers()
        for epoch in (0, 1, 2):
            br.hparams.epoch_counter.current = epoch
            for s_ in scales:
                br.step += 1
                br.fit_batch(Batch(wav * s_, torch.tensor([1.0, 0.83, 0.61, 1.0]), torch.arange(4) % 2))
        torch.cuda.synchronize()
        if graph:
            assert len(br._graphs) == 3, list(br._graphs)          # one per epoch: 0 and 2 differ by the moments
        runs.append({k: v.detach().clone() for k, v in br.modules["ConvAE"].state_dict().items()})
    p0, p1 = runs
    moved = 0
    for k in p0:
        if p0[k].dtype.is_floating_point:
            assert float((p0[k] - p1[k]).abs().max()) <= 5e-6 + 2e-5 * float(p0[k].abs().max()), k
    # the classifier did move in epoch 2 of the eager run (the semantics under test)
    ref = numpy_params(8886)
    assert float((p0["sex_classifier.classify.6.weight"].cpu() - ref["sex_classifier.classify.6.weight"]).abs().max()) > 0


def test_frozen_classifier_and_recon_only():
    """the reference's requires_grad toggling by name (speechbrain_convae_train.py:219-235) and
    config 1 (recon 1.0 only, MSE): frozen parameters get no gradient, the rest still match."""
    from oracle.convae import numpy_params
    from tests import smoke_step
    from speech_anonymization_amd.brain import Batch, Stage
    dev = torch.device("cuda:0")
    br = smoke_step.build(torch.float32, dev, numpy_params(8886))
    br.hparams.recon_loss_weight, br.hparams.sex_loss_weight = 1.0, 0.0
    for name, p in br.modules["ConvAE"].named_parameters():
        p.requires_grad = "sex_classifier" not in name
    wav = smoke_step.make_wave(3, 11360)
    batch = Batch(wav, torch.ones(3), torch.arange(3) % 2)
    out = br.compute_forward(batch, Stage.TRAIN)
    loss = br.compute_objectives(out, batch, Stage.TRAIN)
    loss.backward()
    torch.cuda.synchronize()
    for name, p in br.modules["ConvAE"].named_parameters():
        if "sex_classifier" in name:
            assert p.grad is None, name
        else:
            assert p.grad is not None and torch.isfinite(p.grad).all(), name


def test_entry_script_config1_plumbing(tmp_path):
    """BASELINE config 1 on the GPU path: speechbrain_convae_train.py + convae.yaml, recon 1.0 only
    (MSE), gradient accumulation 3, synthetic utterances; writes train_log.txt and a checkpoint
    directory with the reference's layout."""
    import os
    import speechbrain_convae_train as entry
    cfg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                       "speechbrain_configs", "convae.yaml")
    entry.main([cfg, "--device", "cuda:0", "--model_type", "convae", "--folder", str(tmp_path),
                "--number_of_epochs", "2", "--batch_size", "2", "--synthetic", "12",
                "--synthetic_samples", "11360"])
    out = tmp_path / "8886"
    lines = open(out / "train_log.txt").read().strip().splitlines()
    assert len(lines) == 2 and lines[0].startswith("epoch: 1, lr: ") and "train loss" in lines[0]
    assert "steps: 2" in lines[0]                       # 6 batches / gradient_accumulation 3
    l1 = float(lines[0].split("train loss: ")[1].split(" ")[0])
    l2 = float(lines[1].split("train loss: ")[1].split(" ")[0])
    assert l2 < l1                                      # recon-only MSE goes down
    ck = sorted(os.listdir(out / "save"))
    assert ck and ck[-1].startswith("CKPT+")
    assert {"model.ckpt", "normalizer.ckpt", "noam_scheduler.ckpt", "counter.ckpt", "CKPT.yaml"} <= set(
        os.listdir(out / "save" / ck[-1]))


def test_fused_clip_takes_the_buckets_only_when_every_gradient_lives_there():
    """Brain._grad_flats: after a plain backward every .grad is a view of the three stage buckets (the two-launch
    clip applies); with a frozen stage, a frozen single parameter or an accumulated second backward it does not
    (torch's clip runs), and both clips move the parameters the same way."""
    from oracle.convae import numpy_params
    from tests import smoke_step
    from speech_anonymization_amd.brain import Batch, Stage
    dev = torch.device("cuda:0")
    wav = smoke_step.make_wave(4, 11360)
    batch = Batch(wav, torch.ones(4), torch.arange(4) % 2)

    def backward(br):
        out = br.compute_forward(batch, Stage.TRAIN)
        br.compute_objectives(out, batch, Stage.TRAIN).backward()
        return list(br.modules.parameters())
    br = smoke_step.build("bf16x3", dev, numpy_params(8886))
    params = backward(br)
    flats = br._grad_flats(params)
    assert flats is not None and len(flats) == 3 and sum(f.numel() for f in flats) == sum(p.numel() for p in params)
    params = backward(br)                                           # accumulated: .grad still lives in the first buckets
    assert br._grad_flats(params) is None
    br.optimizer.zero_grad()
    name, p0 = next(iter(br.modules["ConvAE"].named_parameters()))
    p0.requires_grad = False                                        # one parameter of a stage frozen: a hole in its bucket
    params = backward(br)
    assert br._grad_flats(params) is None
    p0.requires_grad = True
    br.optimizer.zero_grad()
    for n_, p in br.modules["ConvAE"].named_parameters():
        p.requires_grad = "sex_classifier" not in n_                # a whole stage frozen: two buckets
    params = backward(br)
    flats = br._grad_flats(params)
    assert flats is not None and len(flats) == 2
    # same step with either clip (max_grad_norm small enough to bite)
    res = []
    for fused in (True, False):
        b2 = smoke_step.build("bf16x3", dev, numpy_params(8886))
        b2.fused_clip, b2.max_grad_norm = fused, 0.05
        b2.step += 1
        b2.fit_batch(batch)
        torch.cuda.synchronize()
        res.append({k: v.detach().clone() for k, v in b2.modules["ConvAE"].state_dict().items() if v.dtype.is_floating_point})
    for k in res[0]:
        assert float((res[0][k] - res[1][k]).abs().max()) <= 1e-6 + 1e-5 * float(res[1][k].abs().max()), k
