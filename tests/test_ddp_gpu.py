"""Data-parallel path end to end on ONE GPU: two ranks share cuda:0 and exchange through gloo
(RCCL needs one GPU per rank; the collectives' call sites, SyncBN statistic exchange, stage
buckets on the side stream and the gradient average are the same code).  Checks the property
DESIGN.md section 7 promises: 2 ranks x B utterances == 1 rank x 2B utterances."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _inputs(B, T):
    from oracle.features import synthetic_feats
    feats = synthetic_feats(B, T, seed=77)
    rs = np.random.RandomState(5)
    target = feats + 0.1 * torch.from_numpy(rs.standard_normal(feats.shape).astype("float32"))
    return feats, target, torch.arange(B) % 2


def _run(model, feats, target, gender, parts=None):
    """parts: [(lo, hi), ...] = the per-rank shards a data-parallel run would use; the loss is then
    the MEAN OVER RANKS of the per-rank mean losses (what averaging the ranks' gradients computes;
    equal to the plain batch mean only for equal shards)."""
    from speech_anonymization_amd import ops
    recon, logp = model(feats.cuda())
    parts = parts or [(0, feats.shape[0])]
    grs, dns = [], []
    for lo, hi in parts:
        _, g_r = ops.recon_loss(recon.detach()[lo:hi].contiguous(), target[lo:hi].cuda().contiguous(), "l1")
        _, dn, _ = ops.cls_losses(logp.detach()[lo:hi].contiguous(), gender[lo:hi].cuda())
        grs.append(g_r.view(hi - lo, *recon.shape[1:]) / len(parts))
        dns.append(dn / len(parts))
    torch.autograd.backward([recon, logp], [0.1 * torch.cat(grs), 0.9 * torch.cat(dns)])
    torch.cuda.synchronize()
    return {k: p.grad.detach().cpu() for k, p in model.named_parameters()}, \
           {k: v.detach().cpu() for k, v in model.state_dict().items() if "running" in k}


def _worker(rank, world, port, q, cuts=(0, 3, 6), sizes=None, fused_head=True):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), SA_DIST_BACKEND="gloo", SA_SAME_DEVICE="1")
    sys.path.insert(0, ROOT)
    from oracle.convae import numpy_params
    from speech_anonymization_amd import distributed as sdist
    from speech_anonymization_amd.convae import ConvAutoencoder
    sdist.ddp_init_group()
    torch.cuda.set_device(0)
    m = ConvAutoencoder(precision="f32", pooling_noise=None)
    m.load_state_dict(numpy_params(8886))
    m.cuda().train()
    m.dp_batch_sizes = sizes
    m.fused_head = fused_head
    feats, target, gender = _inputs(6, 72)
    lo, hi = cuts[rank], cuts[rank + 1]
    calls, plain_now = [0], sdist.all_reduce_now

    def counted(t, *a, **kw):
        calls[0] += 1
        return plain_now(t, *a, **kw)
    sdist.all_reduce_now = counted
    grads, bufs = _run(m, feats[lo:hi], target[lo:hi], gender[lo:hi])
    q.put((rank, {k: v.numpy() for k, v in grads.items()}, {k: v.numpy() for k, v in bufs.items()}, calls[0]))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("cuts,sizes,fused_head",
                         [((0, 3, 6), "equal", True), ((0, 4, 6), [4, 2], True), ((0, 4, 6), [4, 2], False),
                          ((0, 3, 6), None, True), ((0, 4, 6), None, True), ((0, 2, 3, 5, 6), [2, 1, 2, 1], True)],
                         ids=["3+3 global head", "4+2 global head", "4+2 global head, separate launches", "3+3", "4+2",
                              "2+1+2+1 global head (four ranks)"])
def test_two_ranks_equal_one_rank_with_double_batch(cuts, sizes, fused_head):
    """equal shards (3 + 3) and ragged ones (4 + 2: the SyncBatchNorm element COUNTS differ per
    rank and are all-reduced beside the sums, like torch.nn.SyncBatchNorm).  "global head": the
    ranks' batch sizes are known (dp_batch_sizes), the pooled rows and d log p are exchanged once
    each and the FC head runs on the global batch -- two immediate all-reduces fewer per step than
    the per-BatchNorm exchange, same gradients.  "separate launches": the head as it runs when the
    global batch exceeds the one-workgroup kernel (8 ranks x 32 utterances)."""
    import torch.multiprocessing as mp
    from oracle.convae import numpy_params
    from speech_anonymization_amd.convae import ConvAutoencoder
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 1000) + cuts[1] + (2 if sizes is None else 0) + (4 if not fused_head else 0)
    world = len(cuts) - 1
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, cuts, sizes, fused_head)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    m = ConvAutoencoder(precision="f32", pooling_noise=None)
    m.load_state_dict(numpy_params(8886))
    m.cuda().train()
    feats, target, gender = _inputs(6, 72)
    ref, ref_bufs = _run(m, feats, target, gender, parts=[(cuts[i], cuts[i + 1]) for i in range(world)])

    def rel(a, b):
        a, b = torch.as_tensor(a).double(), b.double()
        return float(((a - b) ** 2).sum() / (b ** 2).sum().clamp_min(1e-30))
    from tests.test_convae_gpu import NULL_BIAS
    for k, g in ref.items():
        if k in NULL_BIAS:
            continue
        for r in range(1, world):
            assert np.array_equal(res[0][1][k], res[r][1][k]), k    # every rank holds the average
        assert rel(res[0][1][k], g) < 2e-5, (k, rel(res[0][1][k], g))
    for k, v in ref_bufs.items():                                    # SyncBN: global statistics
        assert rel(res[0][2][k], v) < 1e-8, k
    # immediate (main-stream) exchanges of one step: the counts + 6 BatchNorm sums forward and 6
    # backward; the global head replaces 2 + 2 of them by 1 + 1
    assert all(r[3] == (13 if sizes is None else 11) for r in res), [r[3] for r in res]


def _rccl_worker(port, q, carrier, sizes="equal"):
    """one rank on RCCL: the step first without a process group, then with the data-parallel
    exchanges forced on (SA_FORCE_DP=1).  carrier "torch": torch.distributed's nccl backend;
    "lib": the library's own communicator (sa_comm_*), torch.distributed (gloo) only as rendezvous"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
                      LOCAL_RANK="0", SA_FORCE_DP="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if carrier == "lib":
        os.environ["SA_COMM"] = "lib"
    else:
        os.environ["SA_DIST_BACKEND"] = "nccl"
    sys.path.insert(0, ROOT)
    from oracle.convae import numpy_params
    from speech_anonymization_amd import distributed as sdist
    from speech_anonymization_amd.convae import ConvAutoencoder
    torch.cuda.set_device(0)
    feats, target, gender = _inputs(6, 72)
    out = []
    for dp in (False, True):
        if dp:
            sdist.ddp_init_group()
            assert sdist.dp_active()
            assert torch.distributed.get_backend() == ("gloo" if carrier == "lib" else "nccl")
            assert sdist.lib_comm_active() == (carrier == "lib")
        m = ConvAutoencoder(precision="bf16x3", pooling_noise=None)
        m.load_state_dict(numpy_params(8886))
        m.cuda().train()
        m.dp_batch_sizes = sizes
        if sizes is None:                  # the per-BatchNorm exchange runs the head as separate launches
            m.fused_head = False
        assert m._bn_syncs() == dp
        for _ in range(2):                 # two steps: the second reuses allocator blocks of the first
            for p in m.parameters():
                p.grad = None
            grads, bufs = _run(m, feats, target, gender)
        out.append(({k: v.numpy() for k, v in grads.items()}, {k: v.numpy() for k, v in bufs.items()}))
    # the collective itself, on the side stream, with RCCL's own averaging
    side = torch.cuda.Stream()
    x = torch.arange(1 << 20, dtype=torch.float32, device="cuda")
    side.wait_stream(torch.cuda.current_stream())
    ncalls = -1
    if carrier == "lib":
        from speech_anonymization_amd import _lib as L
        ncalls = L._lib.sa_comm_ncalls()
        b = sdist.StageBuckets([("encoder.w", x)], x.device)
        b.flat["encoder"].copy_(x)
        b.reduce_stage("encoder")
        b.join()
        x = b.flat["encoder"]
    else:
        with torch.cuda.stream(side):
            sdist.StageBuckets._average(x, 1)
        torch.cuda.current_stream().wait_stream(side)
    ok = bool(torch.equal(x.cpu(), torch.arange(1 << 20, dtype=torch.float32)))
    q.put((out, ok, ncalls))
    torch.distributed.barrier()
    if carrier == "lib":
        sdist.lib_comm_destroy()
        assert not sdist.lib_comm_active()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("carrier,sizes", [("torch", "equal"), ("lib", "equal"), ("lib", None)],
                         ids=["torch", "lib", "lib per-BN exchange"])
def test_rccl_world1_is_the_identity(carrier, sizes):
    """RCCL executes: a one-rank communicator bound to cuda:0 (torch.distributed's nccl backend, or
    the library's own through sa_comm_init), the three stage buckets averaged by ncclAvg on the side
    stream, the SyncBatchNorm sums / counts all-reduced for the main stream.
    With one rank every collective is the identity, so gradients and BatchNorm buffers must be
    BIT-equal to the same two steps without a process group.  (Two ranks need two GPUs: RCCL
    refuses two ranks on one device; the two-rank arithmetic is covered over gloo above.)"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(29700 + os.getpid() % 1000 + (carrier == "lib") + 2 * (sizes is None),
                                               q, carrier, sizes))
    p.start()
    (plain, dp), ok, ncalls = q.get(timeout=300)
    if carrier == "lib":
        # two steps x (3 stage buckets + the immediate exchanges of one step: 11 with the FC head on
        # the gathered global batch, 13 with one exchange per BatchNorm1d)
        assert ncalls == 2 * (3 + (13 if sizes is None else 11)), ncalls
    p.join(timeout=120)
    assert p.exitcode == 0
    assert ok
    for k in plain[0]:
        assert np.array_equal(plain[0][k], dp[0][k]), k
    for k in plain[1]:
        assert np.array_equal(plain[1][k], dp[1][k]), k


def _graph_worker(port, q, carrier):
    """one rank on RCCL with the data-parallel exchanges forced on: five train steps of the Brain
    eagerly, then the same five in hipGraph mode (three eager, the capture, replays)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
                      LOCAL_RANK="0", SA_FORCE_DP="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if carrier == "lib":
        os.environ["SA_COMM"] = "lib"
    else:
        os.environ["SA_DIST_BACKEND"] = "nccl"
    sys.path.insert(0, ROOT)
    from oracle.convae import numpy_params
    from tests import smoke_step
    from speech_anonymization_amd import _lib as L, distributed as sdist
    from speech_anonymization_amd.brain import Batch
    torch.cuda.set_device(0)
    sdist.ddp_init_group()
    assert sdist.dp_active() and sdist.capturable() == (carrier == "lib")
    dev = torch.device("cuda:0")
    wav = smoke_step.make_wave(4, 11360)
    runs, calls = [], []
    for graph in (False, True):
        br = smoke_step.build("bf16x3", dev, numpy_params(8886))
        assert br.modules["ConvAE"].dp_batch_sizes == "equal" and br.modules["ConvAE"]._bn_syncs()
        if graph:
            br.hip_graph, br.optimizer = True, None
            br.init_optimizers()
        n0 = L._lib.sa_comm_ncalls() if carrier == "lib" else 0
        for s_ in (1.0, 0.9, 0.8, 1.1, 0.7, 1.2):
            br.step += 1
            br.fit_batch(Batch(wav * s_, torch.tensor([1.0, 0.83, 0.61, 1.0]), torch.arange(4) % 2))
        torch.cuda.synchronize()
        calls.append(L._lib.sa_comm_ncalls() - n0 if carrier == "lib" else -1)
        captured = graph and len(getattr(br, "_graphs", {})) == 1 and all("graph" in e for e in br._graphs.values())
        runs.append(({k: v.detach().cpu().numpy() for k, v in br.modules["ConvAE"].state_dict().items()
                      if v.dtype.is_floating_point}, captured))
    q.put((runs, calls))
    torch.distributed.barrier()
    if carrier == "lib":
        sdist.lib_comm_destroy()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("carrier", ["lib", "torch"])
def test_hip_graph_records_the_data_parallel_exchanges(carrier):
    """run_opts hip_graph under data parallelism, on the library communicator: the RCCL all-reduces of a step
    (three stage buckets on the side stream, the statistic sums / counts / gathered head rows in line) are
    captured with the kernels.  One rank (every collective the identity): parameters after six steps equal
    the eager data-parallel run's, and the enqueue counter shows that only the three eager steps and the
    capture issued collectives from the host -- the replays carry theirs.  On torch.distributed's nccl
    backend the same run_opts fall back to eager steps (distributed.capturable)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_graph_worker, args=(29800 + os.getpid() % 1000 + (carrier == "lib"), q, carrier))
    p.start()
    runs, calls = q.get(timeout=300)
    p.join(timeout=120)
    assert p.exitcode == 0
    (p0, _), (p1, captured) = runs
    assert captured == (carrier == "lib")
    for k in p0:
        assert float(np.abs(p0[k] - p1[k]).max()) <= 5e-6 + 2e-5 * float(np.abs(p0[k]).max()), k
    if carrier == "lib":
        per_step = 3 + 11
        assert calls[0] == 6 * per_step, calls
        assert calls[1] == 4 * per_step, calls          # 3 eager steps + the capture; 2 replays
