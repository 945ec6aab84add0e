"""CPU tests of the host side: YAML surface, data manifest, checkpoint layout, schedules, and
the data-parallel plumbing on world_size-2 gloo."""
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_yaml_surface_and_overrides():
    from speech_anonymization_amd.yaml_loader import load_hyperpyyaml, parse_arguments
    from speech_anonymization_amd import Fbank, InputNormalization, losses, brain
    f, run_opts, ov = parse_arguments([os.path.join(ROOT, "speechbrain_configs", "convae.yaml"), "--device",
                                       "cuda:0", "--model_type", "convae", "--folder", "/tmp/sa_out",
                                       "--batch_size", "2", "--recon_loss_weight=0.1"])
    assert run_opts == {"device": "cuda:0"} and ov["folder"] == "/tmp/sa_out" and ov["recon_loss_weight"] == 0.1
    h = load_hyperpyyaml(open(f), ov)
    assert h["output_folder"] == "/tmp/sa_out/8886" and h["batch_size"] == 2
    assert isinstance(h["compute_features"], Fbank) and isinstance(h["normalize"], InputNormalization)
    assert h["modules"]["normalize"] is h["normalize"]                       # !ref keeps identity
    assert isinstance(h["loss_reconstruction"], losses.MSELoss)
    assert h["Adam"].keywords == {"lr": 0.001, "betas": (0.9, 0.98), "eps": 1e-9}
    assert isinstance(h["noam_annealing"], brain.NoamScheduler) and h["noam_annealing"].n_warmup_steps == 25000
    assert h["checkpointer"].recoverables["normalizer"] is h["normalize"]
    assert h["normalize"].update_until_epoch == 4


@pytest.mark.skipif(not os.path.exists("/root/reference/speechbrain_configs/convae.yaml"),
                    reason="reference checkout not present (GPU box)")
def test_reference_yaml_loads_unmodified():
    from speech_anonymization_amd.yaml_loader import load_hyperpyyaml, Unavailable
    from speech_anonymization_amd import Fbank
    h = load_hyperpyyaml(open("/root/reference/speechbrain_configs/convae.yaml"))
    assert isinstance(h["compute_features"], Fbank)
    assert h["recon_loss_weight"] == 1.0 and h["gradient_accumulation"] == 3 and h["seed"] == 8886
    assert isinstance(h["Transformer"], Unavailable)                         # out of scope: placeholder
    with pytest.raises(RuntimeError):
        h["Transformer"].forward


def test_noam_scheduler_against_reference_log(golden_dir):
    from speech_anonymization_amd.brain import NoamScheduler
    rows = json.load(open(os.path.join(golden_dir, "reference_pins.json")))["noam_train_log"]["steps_lr"]
    opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=1e-3)
    sch = NoamScheduler(1.0, 25000, 768)
    want = {n: lr for n, lr in rows[:2]}
    for n in range(1, max(want) + 1):
        _, lr = sch(opt)
        if n in want:
            assert abs(lr - want[n]) / want[n] < 6e-3
    assert opt.param_groups[0]["lr"] == lr


def test_csv_manifest_wav_and_batches(tmp_path):
    from speech_anonymization_amd import data
    rows = ["ID,duration,wav,spk_id,wrd,gender"]
    for i, (n, g) in enumerate([(4000, "M"), (6400, "F"), (3200, "F")]):
        sig = torch.sin(torch.arange(n) * 0.01 * (i + 1)) * 0.5
        data.write_audio(str(tmp_path / f"u{i}.wav"), sig)
        rows.append(f"u{i},{n / 16000},$data_root/u{i}.wav,spk{i},HELLO WORLD,{g}")
    (tmp_path / "train.csv").write_text("\n".join(rows) + "\n")
    ds = data.CsvDataset(str(tmp_path / "train.csv"), {"data_root": str(tmp_path)}, sorting="ascending")
    assert [r["id"] for r in ds.items] == ["u2", "u0", "u1"] and [r["gender"] for r in ds.items] == [1, 0, 1]
    bs = list(data.batches(ds, 2))
    assert len(bs) == 2
    wav, lens = bs[0].sig
    assert wav.shape == (2, 4000) and torch.allclose(lens, torch.tensor([0.8, 1.0]))
    assert float(wav[0, 3200:].abs().max()) == 0.0 and bs[0].gender.tolist() == [1, 0]
    shard0 = list(data.batches(ds, 2, rank=0, world=2))
    shard1 = list(data.batches(ds, 2, rank=1, world=2))
    # DistributedSampler semantics: 3 utterances over 2 ranks -> 2 each (one wraps around)
    assert [sum(b.gender.numel() for b in sh) for sh in (shard0, shard1)] == [2, 2]


def test_checkpoint_layout_roundtrip(tmp_path):
    from speech_anonymization_amd.brain import EpochCounter, NoamScheduler
    from speech_anonymization_amd.checkpoint import Checkpointer
    from speech_anonymization_amd.convae import ConvAutoencoder
    model = torch.nn.ModuleList([ConvAutoencoder()])
    noam, counter = NoamScheduler(1.0, 25000, 768), EpochCounter(10)
    noam.n_steps, counter.current = 123, 4
    ck = Checkpointer(str(tmp_path / "save"), {"model": model, "noam_scheduler": noam, "counter": counter})
    path = ck.save(epoch=4, meta={"ACC_external": 0.6, "Utility_Retention": 0.61})
    assert os.path.basename(path).startswith("CKPT+") and path.endswith("+00")
    assert sorted(os.listdir(path)) == ["CKPT.yaml", "counter.ckpt", "model.ckpt", "noam_scheduler.ckpt"]
    sd = torch.load(os.path.join(path, "model.ckpt"), weights_only=True)
    assert "0.encoder.0.weight" in sd and sd["0.decoder.1.weight"].shape == (128, 64, 5)
    model2 = torch.nn.ModuleList([ConvAutoencoder()])
    noam2, counter2 = NoamScheduler(1.0, 25000, 768), EpochCounter(10)
    Checkpointer(str(tmp_path / "save"), {"model": model2, "noam_scheduler": noam2,
                                           "counter": counter2}).recover_if_possible()
    assert noam2.n_steps == 123 and counter2.current == 4
    assert torch.equal(model2[0].encoder[0].weight, model[0].encoder[0].weight)


def _dist_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    from speech_anonymization_amd import distributed as sdist
    from speech_anonymization_amd.convae import ConvAutoencoder
    r, lr, w = sdist.ddp_init_group({"distributed_backend": "gloo"})
    model = ConvAutoencoder()
    named = list(model.named_parameters())
    b = sdist.StageBuckets(named, torch.device("cpu"))
    for k, _ in named:
        b.view(k).fill_(float(rank + 1))
    for st in sdist.StageBuckets.STAGES:
        b.reduce_stage(st)
    b.join()
    ok = all(torch.allclose(b.view(k), torch.full_like(b.view(k), 1.5)) for k, _ in named)
    sizes = {st: b.flat[st].numel() for st in b.STAGES}
    sums = torch.tensor([[1.0 + rank, 2.0], [3.0, 4.0 * (rank + 1)]], dtype=torch.float64)
    wf = model._bn_allreduce(sums)
    # ragged ranks: rank 0 holds B=3 utterances of L4=500 rows, rank 1 B=2 of 700
    Bq, L4 = (3, 500) if rank == 0 else (2, 700)
    gc = model._bn_global_counts([Bq * L4, Bq * (L4 - 4), Bq], torch.device("cpu"))
    lo, hi = sdist.shard_batch(10, rank, world)
    q.put((rank, ok, sizes, wf, sums.tolist(), (lo, hi), sdist.if_main_process(), gc.tolist()))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_data_parallel_plumbing_gloo_world2():
    """stage buckets average across ranks, SyncBN statistic sums add up, utterances shard."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_dist_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, sizes, wf, sums, shard, main, gc in res:
        assert ok and wf == 2 and main == (rank == 0)
        assert gc == [3 * 500 + 2 * 700, 3 * 496 + 2 * 696, 5]          # element counts add up across ranks
        assert sizes == {"decoder": 154561, "sex_classifier": 223298, "encoder": 155264}
        assert sums == [[3.0, 4.0], [6.0, 12.0]]
    assert res[0][5] == (0, 5) and res[1][5] == (5, 10)


def test_syncbn_sum_combination_equals_full_batch_statistics():
    """sa_fin_bn_fwd consumes (sum, sumsq) and a count: adding the sums of two half batches and
    doubling the count is the full-batch mean / biased variance (SyncBatchNorm semantics)."""
    x = torch.randn(8, 16, 50, dtype=torch.float64)
    halves = [x[:4], x[4:]]
    s = sum(h.sum(dim=(0, 2)) for h in halves)
    q = sum((h * h).sum(dim=(0, 2)) for h in halves)
    n = 8 * 50
    mean, var = s / n, q / n - (s / n) ** 2
    assert torch.allclose(mean, x.mean(dim=(0, 2))) and torch.allclose(var, x.var(dim=(0, 2), unbiased=False))


def test_shard_indices_give_every_rank_the_same_number_of_batches():
    """DistributedSampler semantics (a rank with an extra batch would hang in the collectives of a
    step its peers never run) and a seed + epoch shuffle like ReproducibleRandomSampler."""
    from speech_anonymization_amd.data import shard_indices
    for n, world, bs in [(11, 2, 5), (7, 4, 2), (16, 8, 1), (3, 8, 2), (10, 1, 4)]:
        shards = [shard_indices(n, True, 8886, 1, r, world) for r in range(world)]
        assert len({len(s) for s in shards}) == 1, (n, world)
        assert len({-(-len(s) // bs) for s in shards}) == 1
        assert set(i for s in shards for i in s) == set(range(n))             # nothing dropped
        assert sum(len(s) for s in shards) == -(-n // world) * world
    a, b = shard_indices(20, True, 8886, 1, 0, 1), shard_indices(20, True, 8886, 2, 0, 1)
    assert a != b and sorted(a) == sorted(b) == list(range(20))              # reshuffled every epoch
    assert shard_indices(20, True, 8886, 1, 0, 1) == a                        # reproducibly


def test_checkpoint_names_are_unique_and_resume_restores_optimizer(tmp_path):
    """two saves within one second get +00 / +01 (speechbrain's suffix) instead of sharing a
    directory; Brain.on_fit_start resumes model, scheduler, counter, optimizer moments and step
    from the newest checkpoint (a rerun on an existing output folder continues, it does not restart)."""
    from speech_anonymization_amd.brain import Brain, EpochCounter, NoamScheduler
    from speech_anonymization_amd.checkpoint import Checkpointer
    import functools

    def make():
        torch.manual_seed(0)
        lin = torch.nn.Linear(4, 3)
        noam, counter = NoamScheduler(1.0, 25000, 768), EpochCounter(10)
        ck = Checkpointer(str(tmp_path / "save"), {"model": lin, "noam_scheduler": noam, "counter": counter})
        br = Brain(modules={"lin": lin}, opt_class=functools.partial(torch.optim.Adam, lr=1e-2),
                   run_opts={"device": "cpu"}, checkpointer=ck)
        return br, lin, noam, counter, ck

    br, lin, noam, counter, ck = make()
    br.on_fit_start()
    assert br.resumed_from is None
    for _ in range(3):
        lin(torch.ones(2, 4)).sum().backward()
        br.optimizer.step(); br.optimizer.zero_grad(); noam(br.optimizer)
    counter.current, br.step, br.avg_train_loss = 2, 3, 0.25
    p1 = ck.save(br, epoch=2, meta={"loss": 0.25})
    p2 = ck.save(br, epoch=2, meta={"loss": 0.25})
    assert p1 != p2 and os.path.isdir(p1) and os.path.isdir(p2)
    assert {os.path.basename(p1)[-3:], os.path.basename(p2)[-3:]} <= {"+00", "+01"}
    assert "optimizer.ckpt" in os.listdir(p2) and "brain.ckpt" in os.listdir(p2)
    want_w = lin.weight.detach().clone()
    want_m = br.optimizer.state[lin.weight]["exp_avg"].clone()

    br2, lin2, noam2, counter2, _ = make()
    assert not torch.equal(lin2.weight, want_w)
    br2.on_fit_start()
    assert br2.resumed_from == p2
    assert torch.equal(lin2.weight, want_w) and noam2.n_steps == 3 and counter2.current == 2
    assert torch.equal(br2.optimizer.state[lin2.weight]["exp_avg"], want_m)
    assert br2.step == 3 and abs(br2.avg_train_loss - 0.25) < 1e-12


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without a torchrun environment starts its two ranks itself (before
    anything touches a GPU), they rendezvous (gloo here), take the MAX over ranks and rank 0's JSON
    line is relayed: what the driver's plain `python3 bench.py --gpus 8` relies on."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
                          "--plumbing-only"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2 and rec["backend"] == "gloo"
    assert rec["plumbing_only"] is True and rec["value"] is None
    assert rec["ms_per_step"] >= 20.0            # rank 1 sleeps 2 x 10 ms per step: MAX over ranks, not rank 0's 10


def test_train_logger_number_format():
    """speechbrain's FileTrainLogger: '%.2f' only for 1 < v < 100 (no abs: the negative losses of
    the endtoend sign branch print in scientific notation, as in the reference's train_log.txt)"""
    from speech_anonymization_amd.brain import FileTrainLogger
    assert FileTrainLogger._fmt({"a": 2.5, "b": -2.5, "c": 0.5, "d": 150.0}) == "a: 2.50, b: -2.50e+00, c: 5.00e-01, d: 1.50e+02"


def test_similarity_and_accuracy_stats():
    """utils/utility_similarity_aggregator.py:4-53 and the AccuracyStats calls of
    speechbrain_convae_train.py:133-149"""
    from speech_anonymization_amd.metrics import AccuracyStats, SimilarityMetricsStats
    s = SimilarityMetricsStats()
    s.append(torch.tensor([0.5, 0.7])); s.append(torch.tensor([0.9]))
    assert abs(float(s.peek()) - 0.7) < 1e-6 and abs(float(s.summarize()) - 0.7) < 1e-6
    assert s.denom == 3 and float(s.summary["average"]) == float(s.summarize())
    a = AccuracyStats()
    logp = torch.log(torch.tensor([[0.9, 0.1], [0.2, 0.8], [0.6, 0.4]]))
    lab = torch.tensor([0, 1, 1])
    a.append(logp.unsqueeze(0), lab.unsqueeze(0), torch.tensor(3).unsqueeze(0))
    assert abs(a.summarize() - 2 / 3) < 1e-9


def test_checkpoint_keep_only_find_and_average(tmp_path):
    """save_and_keep_only(max_keys=[Utility_Retention], min_keys=[ACC_external], num_to_keep),
    find_checkpoints(max_key / min_key) and average_checkpoints (speechbrain_convae_train.py:338-343,
    404-415)"""
    import time
    from speech_anonymization_amd.checkpoint import Checkpointer
    lin = torch.nn.Linear(3, 2)
    bn = torch.nn.BatchNorm1d(2)
    model = torch.nn.ModuleList([lin, bn])
    ck = Checkpointer(str(tmp_path / "save"), {"model": model})
    vals = [(0.60, 0.50), (0.55, 0.70), (0.65, 0.40), (0.50, 0.60)]      # (ACC_external, Utility_Retention)
    weights = []
    for i, (acc, ur) in enumerate(vals):
        with torch.no_grad():
            lin.weight.fill_(float(i)); bn.num_batches_tracked.fill_(i)
        weights.append(float(i))
        ck.save_and_keep_only(epoch=i, meta={"ACC_external": acc, "Utility_Retention": ur},
                              max_keys=["Utility_Retention"], min_keys=["ACC_external"], num_to_keep=1)
        time.sleep(0.01)
    kept = ck._list()
    # best Utility_Retention = #1 (0.70), lowest ACC_external = #3 (0.50) which is also the most recent
    assert len(kept) == 2
    best_ur = ck.find_checkpoints(max_key="Utility_Retention")
    best_acc = ck.find_checkpoints(min_key="ACC_external")
    assert ck._meta(os.path.basename(best_ur[0]))["Utility_Retention"] == 0.70
    assert ck._meta(os.path.basename(best_acc[0]))["ACC_external"] == 0.50
    avg = Checkpointer.average_checkpoints(ck.find_checkpoints(), recoverable_name="model")
    assert torch.allclose(avg["0.weight"], torch.full((2, 3), (1.0 + 3.0) / 2))
    assert int(avg["1.num_batches_tracked"]) == 2 and avg["1.num_batches_tracked"].dtype == torch.long


def test_head_plan_places_the_ranks_rows(monkeypatch):
    """ConvAutoencoder._head_plan: where this rank's rows sit in the gathered global batch of the FC
    head (offset, global rows, world); sizes that do not describe this rank are refused."""
    from speech_anonymization_amd import distributed as sdist
    from speech_anonymization_amd.convae import ConvAutoencoder, SaHipError
    m = ConvAutoencoder()
    assert m._head_plan(4) is None                               # one process: nothing to gather
    monkeypatch.setattr(sdist, "dp_active", lambda: True)
    monkeypatch.setattr(sdist, "world_size", lambda: 3)
    monkeypatch.setattr(sdist, "rank", lambda: 1)
    assert m._head_plan(4) is None                               # sizes unknown: per-BatchNorm exchange
    m.dp_batch_sizes = "equal"
    assert m._head_plan(4) == (4, 12, 3)
    m.dp_batch_sizes = [5, 4, 2]
    assert m._head_plan(4) == (5, 11, 3)
    for bad in ([5, 3, 2], [5, 4], "ragged"):
        m.dp_batch_sizes = bad
        with pytest.raises(SaHipError):
            m._head_plan(4)
    m.sync_bn = False
    m.dp_batch_sizes = "equal"
    assert m._head_plan(4) is None
