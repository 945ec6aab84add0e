"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL ("nccl" backend on
ROCm) on xGMI.  Replaces what speechbrain does for the reference at
speechbrain_convae_train.py:524 (ddp_init_group), :534 (run_on_main), :314 (if_main_process) and
the DistributedDataParallel / SyncBatchNorm wrap inside Brain.

The path shards by utterance: each rank runs the whole step on its own B utterances; the only
exchanges are (1) the tiny BatchNorm statistic sums of the classifier (SyncBatchNorm semantics,
convae.ConvAutoencoder._bn_allreduce) and (2) the gradient average.  Gradients are written by
the backward kernels straight into three flat fp32 stage buckets (classifier 0.89 MB, decoder
0.62 MB, encoder 0.62 MB, in completion order); each bucket is all-reduced on a SIDE stream as soon as its stage is
complete, so the collective (latency-bound at this size) overlaps the rest of backward, and the
main stream joins the side stream once, before the optimizer touches the gradients.

Two carriers for the same exchanges: torch.distributed's "nccl" backend (RCCL; the default) or the
library's own communicator (`SA_COMM=lib` / run_opts["comm"] = "lib": csrc/sa_comm.hip,
include/sa_hip.h "data-parallel exchange" -- SURVEY 8b's sa_comm_init / sa_comm_destroy), which
owns its side stream and events; torch.distributed is then only the rendezvous that carries the
128-byte RCCL id (gloo by default, so that the process holds ONE RCCL communicator).
"""
import os

import torch
import torch.distributed as dist


def is_distributed():
    return int(os.environ.get("WORLD_SIZE", "1")) > 1


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def dp_active():
    """True when the data-parallel exchanges run: an initialised process group with more than one
    rank -- or with ONE rank under SA_FORCE_DP=1, the rehearsal that puts the real RCCL calls
    (communicator bound to the device, ncclAvg on the side stream, SyncBatchNorm sums and counts)
    on a one-GPU box, where every collective is the identity and the result must be bit-equal to
    the plain single-process step (tests/test_ddp_gpu.py::test_rccl_world1_is_the_identity)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("SA_FORCE_DP") == "1"


def capturable():
    """the data-parallel exchanges can be recorded into a hipGraph: only on the library communicator
    (sa_comm_*: plain RCCL launches on streams the library owns, forked from and joined to the capturing
    stream by events).  torch.distributed's nccl backend captured too in most runs of the one-rank
    rehearsal, but one run of the test suite hung inside the capture (its watchdog thread polls events
    while the capture is global-mode): the step then runs eagerly on that carrier.  gloo reduces on the host."""
    if not dp_active():
        return True
    return lib_comm_active()


def lib_comm_active():
    """the library-owned RCCL communicator is up (sa_comm_init has run in this process)"""
    from . import _lib as L
    return L._lib is not None and L._lib.sa_comm_world() > 0


def lib_comm_init(device_index):
    """sa_comm_unique_id on rank 0 -> the 128 bytes travel through the torch.distributed group ->
    sa_comm_init on every rank (collective)."""
    import ctypes as C
    from . import _lib as L
    lib = L.load()
    ident = C.create_string_buffer(128)
    if dist.get_rank() == 0:
        L.check(lib.sa_comm_unique_id(ident), "sa_comm_unique_id")
    box = [ident.raw]
    if dist.get_world_size() > 1:
        dist.broadcast_object_list(box, src=0)
    L.check(lib.sa_comm_init(dist.get_rank(), dist.get_world_size(), box[0], int(device_index)),
            "sa_comm_init")


def lib_comm_destroy():
    from . import _lib as L
    if lib_comm_active():
        L.check(L._lib.sa_comm_destroy(), "sa_comm_destroy")


def _lib_allreduce(t, avg, inline=False):
    import ctypes as C
    from . import _lib as L
    code = {torch.float32: L.F32, torch.float64: L.F64}[t.dtype]
    fn = L._lib.sa_comm_allreduce_inline if inline else L._lib.sa_comm_allreduce
    L.check(fn(L.ptr(t), C.c_longlong(t.numel()), code, int(avg), L.stream()), "sa_comm_allreduce")


def all_reduce_now(t):
    """in-place SUM whose result the next kernel on the current stream consumes (the BatchNorm
    statistic sums and counts): on the library communicator the collective is enqueued in the
    current stream itself -- no side stream, no event hop"""
    if lib_comm_active() and t.is_cuda:
        _lib_allreduce(t, False, inline=True)
    else:
        dist.all_reduce(t)


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def if_main_process():
    return rank() == 0


def run_on_main(func, args=None, kwargs=None):
    if if_main_process():
        func(*(args or ()), **(kwargs or {}))
    if world_size() > 1:
        dist.barrier()


def ddp_init_group(run_opts=None):
    """init_process_group from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT).  backend: run_opts["distributed_backend"] or nccl on GPU, gloo on
    CPU.  Returns (rank, local_rank, world_size)."""
    run_opts = run_opts or {}
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    rk = int(os.environ.get("RANK", "0"))
    lr = int(os.environ.get("LOCAL_RANK", "0"))
    if (ws > 1 or os.environ.get("SA_FORCE_DP") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        use_lib = (run_opts.get("comm") or os.environ.get("SA_COMM", "torch")) == "lib"
        backend = (run_opts.get("distributed_backend") or os.environ.get("SA_DIST_BACKEND")
                   or ("nccl" if torch.cuda.is_available() and not use_lib else "gloo"))
        if os.environ.get("SA_SAME_DEVICE") == "1":   # rehearsal: several ranks share GPU 0 (gloo only)
            lr = 0
        if backend == "nccl":
            torch.cuda.set_device(lr)
            dist.init_process_group(backend, rank=rk, world_size=ws,
                                    device_id=torch.device("cuda", lr))
        else:
            dist.init_process_group(backend, rank=rk, world_size=ws)
        if use_lib:
            torch.cuda.set_device(lr)
            lib_comm_init(lr)
    return rk, lr, ws


def shard_batch(n_items, rank_, world):
    """contiguous utterance shard [lo, hi) of rank `rank_` (weak scaling: the caller normally
    builds B utterances per rank; this is for sharding a fixed global list)."""
    per = -(-n_items // world)
    lo = min(n_items, rank_ * per)
    return lo, min(n_items, lo + per)


class StageBuckets:
    """Flat fp32 gradient buckets, one per backward stage, with views handed to the kernels."""

    STAGES = ("decoder", "sex_classifier", "encoder")

    def __init__(self, named_params, device, side_stream=None):
        self.views, self.flat = {}, {}
        for st in self.STAGES:
            items = [(k, p) for k, p in named_params if k.startswith(st)]
            n = sum(p.numel() for _, p in items)
            flat = torch.empty(n, dtype=torch.float32, device=device)
            off = 0
            for k, p in items:
                self.views[k] = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
            self.flat[st] = flat
        self.side = side_stream
        self.pending = False

    def view(self, key):
        return self.views[key]

    @staticmethod
    def _average(flat, w):
        """one collective: RCCL averages in the reduction itself (ncclAvg); gloo has no AVG"""
        if dist.get_backend() == "nccl":
            dist.all_reduce(flat, op=dist.ReduceOp.AVG)
        else:
            dist.all_reduce(flat)
            flat.div_(w)

    def reduce_stage(self, stage):
        """average bucket `stage` across ranks, asynchronously on the side stream.  The buckets
        are allocated per backward on purpose: autograd adopts their views as .grad, and with
        gradient accumulation the previous micro-step's .grad must stay intact while this one is
        written (a persistent bucket would be overwritten under it)."""
        w = world_size()
        if not dp_active():
            return
        flat = self.flat[stage]
        if flat.is_cuda and lib_comm_active():
            # library communicator: its own side stream waits for this stream's work so far; the
            # bucket stays referenced (.grad views) until after join(), so no record_stream
            _lib_allreduce(flat, True)
            self.pending = "lib"
            return
        if self.side is None or not flat.is_cuda:  # CPU / gloo (tests)
            self._average(flat, w)
            return
        self.side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.side):
            self._average(flat, w)
        # allocated on the main stream, used on the side stream: the caching allocator must not
        # hand the block out again before the collective has finished with it
        flat.record_stream(self.side)
        self.pending = True

    def join(self):
        if self.pending == "lib":
            from . import _lib as L
            L.check(L._lib.sa_comm_join(L.stream()), "sa_comm_join")
        elif self.pending:
            torch.cuda.current_stream().wait_stream(self.side)
        self.pending = False
