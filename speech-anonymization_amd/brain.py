"""Host-side training loop: a minimal ``Brain`` with speechbrain's hook order, and
``SexAnonymizationTraining`` mirroring the reference's overrides one for one
(speechbrain_convae_train.py:41-416: compute_forward :42-69, compute_objectives :71-193,
fit_batch :195-259, evaluate_batch :271-276).  speechbrain is not a dependency: the loop, the
Noam schedule, the epoch counter, check_gradients (non-finite guard + clip_grad_norm_) and the
DDP initialisation are restated here.  All per-batch tensor work happens in libsa_hip.so.
"""
import enum
import os
import types

import torch

from . import distributed as sdist


class Stage(enum.Enum):
    TRAIN = 1
    VALID = 2
    TEST = 3


class EpochCounter:
    """speechbrain.utils.epoch_loop.EpochCounter"""

    def __init__(self, limit):
        self.current, self.limit = 0, int(limit)

    def __iter__(self):
        return self

    def __next__(self):
        if self.current < self.limit:
            self.current += 1
            return self.current
        raise StopIteration


class NoamScheduler:
    """speechbrain.nnet.schedulers.NoamScheduler: lr = lr_initial * model_size^-0.5 *
    min(n^-0.5, n * n_warmup^-1.5); pinned by the lr column of the reference's train_log.txt."""

    def __init__(self, lr_initial, n_warmup_steps, model_size=None):
        self.lr_initial, self.n_warmup_steps = lr_initial, n_warmup_steps
        self.current_lr, self.n_steps = lr_initial, 0
        self.normalize = n_warmup_steps ** 0.5 if model_size is None else model_size ** (-0.5)

    def __call__(self, opt):
        self.n_steps += 1
        current_lr = getattr(opt, "_sa_host_lr", None)
        if current_lr is None:
            current_lr = opt.param_groups[0]["lr"]
        lr = self.lr_initial * self.normalize * min(self.n_steps ** (-0.5),
                                                    self.n_steps * self.n_warmup_steps ** (-1.5))
        for g in opt.param_groups:
            if torch.is_tensor(g["lr"]):      # capturable optimizer (hipGraph mode): the rate lives on the device
                g["lr"].fill_(lr)
            else:
                g["lr"] = lr
        opt._sa_host_lr = lr
        self.current_lr = current_lr
        return current_lr, lr

    def state_dict(self):
        return {"losses": [], "n_steps": self.n_steps}

    def load_state_dict(self, sd):
        self.n_steps = sd["n_steps"]


class Batch:
    """what speechbrain's PaddedBatch gives the hooks: .sig = (wavs, rel. lengths), .gender,
    .tokens_bos, .to(device) (speechbrain_convae_train.py:45-47,77-80,508-510)."""

    def __init__(self, wavs, lens, gender, tokens_bos=None, ids=None):
        self.sig = (wavs, lens)
        self.gender = gender
        self.tokens_bos = (tokens_bos if tokens_bos is not None
                           else torch.ones(wavs.shape[0], 1, dtype=torch.long), lens)
        self.id = ids

    def to(self, device):
        w, l = self.sig
        tb, tl = self.tokens_bos
        out = Batch.__new__(Batch)
        out.sig = (w.to(device, non_blocking=True), l.to(device, non_blocking=True))
        out.gender = self.gender.to(device, non_blocking=True)
        out.tokens_bos = (tb.to(device, non_blocking=True), tl.to(device, non_blocking=True))
        out.id = self.id
        return out


def settle_python_heap():
    """gc.collect() + gc.freeze(): everything alive once the step has reached its steady state
    (torch, the modules, the kernel tables: a few million objects) moves to the permanent
    generation.  Without it the interpreter's first full collection arrives about sixty steps into
    the process and takes ~100 ms of host time (measured: `tools/hiccup_probe.py`, step 63, 102 ms)
    during which nothing is launched and the GPU drains its queue -- ten train steps lost at once;
    afterwards full collections only walk what the steps themselves allocate."""
    import gc
    gc.collect()
    gc.freeze()


class Brain:
    """The subset of speechbrain.core.Brain the reference relies on."""
    GC_FREEZE_STEP = 8            # fit(): settle the interpreter heap after this many steps

    def __init__(self, modules=None, opt_class=None, hparams=None, run_opts=None, checkpointer=None):
        run_opts = dict(run_opts or {})
        self.device = torch.device(run_opts.get("device", "cuda:0" if torch.cuda.is_available() else "cpu"))
        if self.device.type == "cuda":
            # the HIP library launches on torch's CURRENT device / stream: `--device cuda:1` in one
            # process (the reference's CLI) must make that device current, on this thread and -- via
            # the tensors' device -- on autograd's
            torch.cuda.set_device(self.device)
        self.max_grad_norm = float(run_opts.get("max_grad_norm", 5.0))
        self.nonfinite_patience = int(run_opts.get("nonfinite_patience", 3))
        self.lazy_finite_check = bool(run_opts.get("lazy_finite_check", True))
        # hip_graph: capture the whole train step (forward, losses, backward, clip, Adam) in a
        # hipGraph per batch shape and replay it (single process; see SexAnonymizationTraining)
        self.hip_graph = bool(run_opts.get("hip_graph", False)) and self.device.type == "cuda"
        # gc_freeze (default on): fit() settles the interpreter heap after its first steps
        # (settle_python_heap: gc.collect + gc.freeze).  gc.freeze() is process-global -- cyclic garbage
        # among the objects alive at that point is never collected afterwards -- so a host application
        # that embeds the Brain can switch it off
        self.gc_freeze = bool(run_opts.get("gc_freeze", True))
        # fused_clip (default on; SA_FUSED_CLIP=0): clip_grad_norm_ as two launches on the backward's flat gradient
        # buckets when every gradient lives there (check_gradients / _grad_flats); same arithmetic up to the
        # order of the sum of squares
        self.fused_clip = bool(run_opts.get("fused_clip", os.environ.get("SA_FUSED_CLIP", "1") == "1"))
        # calibrate_xcd (default OFF; SA_CALIBRATE_XCD=1): see on_fit_start.  Measured: the per-XCD medians of one
        # launch carry 2-3 % of noise, as much as the effect (odd XCDs ~4 % slower than even ones on every chip
        # seen) -- the calibrated step was 1 % SLOWER (8.85 vs 8.74 ms); the mechanism stays for experiments
        self.calibrate_xcd = bool(run_opts.get("calibrate_xcd", os.environ.get("SA_CALIBRATE_XCD", "0") == "1"))
        self.distributed_launch = bool(run_opts.get("distributed_launch", sdist.is_distributed()))
        self.modules = torch.nn.ModuleDict(modules or {})
        # dp_batch_sizes (default "equal"): how the data-parallel ranks' batches relate.  The data
        # layer deals every rank the same number of utterances per batch (data.shard_indices,
        # DistributedSampler semantics), which lets the FC head exchange its pooled rows once and
        # run on the global batch (ConvAutoencoder._head_plan).  A caller that shards differently
        # passes the per-rank sizes, or None for the per-BatchNorm exchange that needs no sizes.
        # Applied to the modules in on_fit_start.
        self.dp_batch_sizes = run_opts.get("dp_batch_sizes", "equal")
        self.opt_class = opt_class
        self.hparams = types.SimpleNamespace(**(hparams or {}))
        self.checkpointer = checkpointer
        self.step, self.nonfinite_count, self.optimizer = 0, 0, None
        self.avg_train_loss = 0.0

    # ---- hooks to override
    def compute_forward(self, batch, stage):
        raise NotImplementedError

    def compute_objectives(self, predictions, batch, stage):
        raise NotImplementedError

    def on_stage_start(self, stage, epoch=None):
        pass

    def on_stage_end(self, stage, stage_loss, epoch=None):
        pass

    def on_fit_start(self):
        """speechbrain order: modules to the device, optimizer, then resume from the newest
        checkpoint if the output folder holds one (all ranks; model, scheduler, normaliser, epoch
        counter, optimizer moments -- reference checkpoints carry optimizer.ckpt too)."""
        self.modules.to(self.device)
        for m in self.modules.values():
            if hasattr(m, "dp_batch_sizes"):
                m.dp_batch_sizes = self.dp_batch_sizes
        if self.calibrate_xcd and self.device.type == "cuda":
            # once per process: the XCDs of a chip run the persistent convolution kernels at different speeds
            # (up to 12 %); their workgroups get tile ranges in proportion (ops.calibrate_xcd, ~0.1 s)
            from . import ops
            if ops._xcd_weights is None:
                ops.calibrate_xcd(self.device)
        self.init_optimizers()
        if self.checkpointer is not None:
            if self.optimizer is not None:
                self.checkpointer.add_recoverable("optimizer", self.optimizer)
            self.resumed_from = self.checkpointer.recover_if_possible(device=self.device, brain=self)

    def init_optimizers(self):
        if self.opt_class is not None and self.optimizer is None:
            params = list(self.modules.parameters())
            base = getattr(self.opt_class, "func", self.opt_class)
            given = getattr(self.opt_class, "keywords", None) or {}
            if base is torch.optim.Adam and "fused" not in given and params and all(p.is_cuda for p in params):
                try:        # PyTorch's single-launch Adam: same update, ~8 fewer launches per step
                    if self.hip_graph:
                        # capturable: step counts and the learning rate are device tensors, so a
                        # captured optimizer.step() follows the Noam schedule on replay
                        self.optimizer = self.opt_class(params, fused=True, capturable=True)
                        for g in self.optimizer.param_groups:
                            self.optimizer._sa_host_lr = float(g["lr"])
                            g["lr"] = torch.tensor(float(g["lr"]), dtype=torch.float32, device=self.device)
                    else:
                        self.optimizer = self.opt_class(params, fused=True)
                    return
                except (TypeError, RuntimeError):
                    self.hip_graph = False
            self.optimizer = self.opt_class(params)

    def check_gradients(self, loss):
        """speechbrain semantics: count non-finite losses (raise after `nonfinite_patience`),
        otherwise clip the global gradient norm to max_grad_norm."""
        bad = self._count_nonfinite(loss)
        if bad:
            self.nonfinite_count += bad
            if self.nonfinite_count > self.nonfinite_patience:
                raise ValueError("Loss is not finite and patience is exhausted.")
            if not (self.lazy_finite_check and loss.is_cuda):
                return False
        params = getattr(self, "_clip_params", None)
        if params is None:          # walking the module tree for 56 parameters costs 0.3 ms per step
            params = self._clip_params = list(self.modules.parameters())
        flats = self._grad_flats(params) if self.fused_clip else None
        if flats:
            from . import ops       # the same clip on the backward's flat buckets: 2 launches instead of ~8
            ops.clip_flats(flats, self.max_grad_norm)
        else:
            torch.nn.utils.clip_grad_norm_(params, self.max_grad_norm)
        return True

    def _grad_flats(self, params):
        """the flat gradient buckets of the ConvAE's last backward, if every gradient the clip would see lives in
        one of them and they hold nothing else (no frozen slice, no accumulated or cloned gradient, no other
        module's parameters); else None (torch's clip)."""
        model = self.modules["ConvAE"] if "ConvAE" in self.modules else None
        flats = getattr(model, "_last_flats", None) if model is not None else None
        if not flats:
            return None
        spans = [(f.data_ptr(), f.data_ptr() + 4 * f.numel(), f.numel()) for f in flats]
        seen = [0] * len(flats)
        for p in params:
            g = p.grad
            if g is None:
                continue
            if g.dtype != torch.float32 or not g.is_contiguous():
                return None
            a = g.data_ptr()
            for i, (lo, hi, _) in enumerate(spans):
                if lo <= a < hi:
                    seen[i] += g.numel()
                    break
            else:
                return None
        if any(n != span[2] for n, span in zip(seen, spans)):
            return None
        return flats

    def _count_nonfinite(self, loss):
        """number of non-finite losses newly known to the host.  `torch.isfinite(loss)` read on the
        host would stall the CPU until the GPU has finished the whole backward of this step, every
        step.  With lazy_finite_check (run_opts, default True on GPU) the flag is accumulated in a
        device counter whose value travels to pinned host memory with an asynchronous copy; the
        host looks at the newest copy that has ALREADY arrived (event query, never a wait), so a
        non-finite loss is counted a step or two late and nothing synchronises.  False restores
        the synchronous check.  Capture-safe (hipGraph mode): the counter update and the copy are
        recorded in the graph, the event is recorded by the replaying host code."""
        if not (self.lazy_finite_check and loss.is_cuda):
            return 0 if bool(torch.isfinite(loss)) else 1
        st = getattr(self, "_finite_state", None)
        if st is None:
            st = self._finite_state = dict(dev=torch.zeros((), dtype=torch.int32, device=loss.device),
                                           host=torch.zeros((), dtype=torch.int32).pin_memory(),
                                           event=torch.cuda.Event(), seen=0, pending=False)
        new = self._poll_nonfinite()
        st["dev"].add_((~torch.isfinite(loss.detach())).to(torch.int32))
        st["host"].copy_(st["dev"], non_blocking=True)
        if not torch.cuda.is_current_stream_capturing():
            st["event"].record()
            st["pending"] = True
        return new

    def _poll_nonfinite(self, wait=False):
        st = getattr(self, "_finite_state", None)
        if st is None or not st["pending"]:
            return 0
        if wait:
            st["event"].synchronize()
        elif not st["event"].query():
            return 0
        st["pending"] = False
        total = int(st["host"])
        new, st["seen"] = total - st["seen"], total
        return max(new, 0)

    def fit_batch(self, batch):
        outputs = self.compute_forward(batch, Stage.TRAIN)
        loss = self.compute_objectives(outputs, batch, Stage.TRAIN)
        loss.backward()
        if self.check_gradients(loss):
            self.optimizer.step()
        self.optimizer.zero_grad()
        return loss.detach()

    def evaluate_batch(self, batch, stage):
        out = self.compute_forward(batch, stage=stage)
        return self.compute_objectives(out, batch, stage=stage).detach()

    def update_average(self, loss, avg_loss):
        """running mean of the finite losses (speechbrain's update_average), kept ON THE DEVICE:
        reading the loss on the host every step would stall the CPU until the whole backward has
        finished -- the stall lazy_finite_check avoids.  It is read once, at the end of the epoch."""
        if not loss.is_cuda:
            if torch.isfinite(loss):
                avg_loss -= avg_loss / self.step
                avg_loss += float(loss) / self.step
            return avg_loss
        if not torch.is_tensor(avg_loss):
            avg_loss = torch.full((), float(avg_loss), device=loss.device, dtype=torch.float32)
        new = avg_loss + (loss.detach().float() - avg_loss) / self.step
        return torch.where(torch.isfinite(loss), new, avg_loss)

    def fit(self, epoch_counter, train_set, valid_set=None, progressbar=False, **_):
        self.on_fit_start()
        for epoch in epoch_counter:
            self.on_stage_start(Stage.TRAIN, epoch)
            self.modules.train()
            self.nonfinite_count = 0
            self.avg_train_loss = 0.0
            self.step = 0
            for batch in train_set:
                self.step += 1
                loss = self.fit_batch(batch)
                self.avg_train_loss = self.update_average(loss, self.avg_train_loss)
                if self.gc_freeze and self.step == self.GC_FREEZE_STEP and not self.__dict__.get("_gc_frozen"):
                    settle_python_heap()
                    self._gc_frozen = True
            self.avg_train_loss = float(self.avg_train_loss)          # the epoch's one host read
            self.on_stage_end(Stage.TRAIN, self.avg_train_loss, epoch)
            if valid_set is not None:
                self.on_stage_start(Stage.VALID, epoch)
                self.modules.eval()
                avg, n = 0.0, 0
                with torch.no_grad():
                    for batch in valid_set:
                        n += 1
                        avg += (float(self.evaluate_batch(batch, Stage.VALID)) - avg) / n
                self.on_stage_end(Stage.VALID, avg, epoch)

    def on_evaluate_start(self, max_key=None, min_key=None):
        pass

    def evaluate(self, test_set, max_key=None, min_key=None, **_):
        self.on_evaluate_start(max_key=max_key, min_key=min_key)
        self.on_stage_start(Stage.TEST, None)
        self.modules.eval()
        avg, n = 0.0, 0
        with torch.no_grad():
            for batch in test_set:
                n += 1
                avg += (float(self.evaluate_batch(batch, Stage.TEST)) - avg) / n
        self.on_stage_end(Stage.TEST, avg, None)
        return avg


class SexAnonymizationTraining(Brain):
    """The reference's Brain subclass for model_type convae / endtoend.  `self.asr_brain` (an
    asr.ASR, attached like speechbrain_convae_train.py:560-574 attaches the pretrained recogniser)
    switches the frozen-ASR utility loss on (throughput-only, parity unpinned);
    without it the term is 0, whatever utility_loss_weight says."""

    def features(self, wavs, wav_lens):
        feats = self.hparams.compute_features(wavs)
        current_epoch = self.hparams.epoch_counter.current
        # normalisation, top-dB clamp and the pad-to-36 (speechbrain_convae_train.py:62-63) are
        # one fused pass
        pad = 36 if self.hparams.model_type != "fcae" else None
        return self.modules.normalize(feats, wav_lens, epoch=current_epoch, pad_multiple=pad)

    def compute_forward(self, batch, stage):
        batch = batch.to(self.device)
        wavs, wav_lens = batch.sig
        feats = self.features(wavs, wav_lens)
        return self.modules.ConvAE(feats)

    def compute_objectives(self, predictions, batch, stage):
        reconstructed_speech, sex_logits = predictions
        batch = batch.to(self.device)
        sex_label = batch.gender
        wavs, wav_lens = batch.sig
        feats = self.features(wavs, wav_lens)          # the reference recomputes the targets (:82-87)
        hp = self.hparams
        utility_loss = 0.0
        asr_brain = getattr(self, "asr_brain", None)
        if stage == Stage.TRAIN and hp.utility_loss_weight > 0 and asr_brain is not None:
            from . import asr                                   # reference :97-102
            tokens_bos, _ = batch.tokens_bos
            utility_loss = asr.utility_loss(asr_brain, hp.loss_utility, feats, reconstructed_speech,
                                            wav_lens, tokens_bos, batch)
        B = reconstructed_speech.shape[0]
        recon_loss = hp.loss_reconstruction(reconstructed_speech.view(B, -1), feats.view(B, -1))
        sex_loss = hp.loss_sex_classification(sex_logits, sex_label)
        if hp.model_type == "endtoend":
            if hp.recon_loss_weight == 0.0 and hp.utility_loss_weight == 0.0:
                loss = hp.sex_loss_weight * sex_loss
            else:
                confusion_loss = hp.loss_confusion(sex_logits, None)
                loss = (hp.recon_loss_weight * recon_loss - hp.sex_loss_weight * sex_loss
                        + hp.utility_loss_weight * utility_loss
                        - hp.confusion_loss_weight * confusion_loss)
        else:
            loss = (hp.recon_loss_weight * recon_loss + hp.sex_loss_weight * sex_loss
                    + hp.utility_loss_weight * utility_loss)
        self.last_losses = dict(recon=recon_loss.detach(), sex=sex_loss.detach())
        if stage != Stage.TRAIN:
            # reference :130-149: accuracy of the model's own classifier head, and of the externally
            # trained x-vector classifier on the original and on the reconstructed features
            n = torch.tensor(sex_label.shape[0], device=sex_logits.device).unsqueeze(0)
            self.sex_classification_acc.append(sex_logits.unsqueeze(0), sex_label.unsqueeze(0), n)
            ext = getattr(self, "external_classifier", None)
            if ext is not None:
                with torch.no_grad():
                    lo, _, _ = ext.classify_batch_feats(feats)
                    lr, _, _ = ext.classify_batch_feats(reconstructed_speech.detach())
                self.sex_classification_acc_extern_orig.append(lo.unsqueeze(0), sex_label.unsqueeze(0), n)
                self.sex_classification_acc_extern.append(lr.unsqueeze(0), sex_label.unsqueeze(0), n)
            if asr_brain is not None:
                # reference :156-163 (VALID) / :164-173 (TEST, there together with beam search and WER,
                # which stay out of scope): cosine similarity of the frozen recogniser's encoder outputs
                # on the reconstructed and on the original features, one value per utterance ->
                # "Utility_Retention", the max_key of save_and_keep_only
                from . import ops
                tokens_bos, _ = batch.tokens_bos
                with torch.no_grad():
                    recon_enc, _ = asr_brain.get_predictions(reconstructed_speech.detach(), wav_lens, tokens_bos,
                                                             batch, eval=True, do_ctc=False)
                    orig_enc, _ = asr_brain.get_predictions(feats, wav_lens, tokens_bos, batch, eval=True,
                                                            do_ctc=False)
                    nb = recon_enc.shape[0]
                    self.utility_similarity_aggregator.append(
                        ops.cosine_rows(recon_enc.reshape(nb, -1), orig_enc.reshape(nb, -1)))
        return loss

    def apply_epoch_schedule(self):
        """HEAD's epoch-parity schedule (speechbrain_convae_train.py:212-235), enabled with
        hparams.epoch_parity_schedule; the historical runs (results/*/hyperparams.yaml) trained
        every parameter with fixed weights, which is the default here."""
        hp = self.hparams
        if not getattr(hp, "epoch_parity_schedule", False):
            return
        joint = hp.epoch_counter.current % 2 == 0
        if joint:
            hp.recon_loss_weight, hp.sex_loss_weight = 0.0, 0.8
            hp.utility_loss_weight, hp.confusion_loss_weight = 0.2, 0.0
        else:
            hp.recon_loss_weight, hp.sex_loss_weight, hp.utility_loss_weight = 0.0, 0.5, 0.0
        for name, param in self.modules.ConvAE.named_parameters():
            param.requires_grad = ("sex_classifier" not in name) if joint else ("sex_classifier" in name)

    def _zero_grads_of_frozen(self):
        """torch 1.10's optimizer.zero_grad() (the reference's pin; set_to_none=False by default)
        leaves zero-filled gradients behind, so a parameter the epoch-parity schedule freezes AFTER
        it has had a gradient keeps being updated by Adam on its moments.  Gradients are dropped
        (set to None) here so that autograd adopts the kernels' bucket views without a copy; the
        zero gradient of a frozen parameter with optimizer state is put back just before step()."""
        st = self.optimizer.state
        for group in self.optimizer.param_groups:
            for p in group["params"]:
                if p.grad is None and not p.requires_grad and p in st and len(st[p]):
                    z = getattr(self, "_zero_grad_cache", None)
                    if z is None:
                        z = self._zero_grad_cache = {}
                    g = z.get(p)
                    if g is None:
                        g = z[p] = torch.zeros_like(p)
                    p.grad = g

    # ---- hipGraph mode (run_opts hip_graph=True, one process): the ~230 launches of a step are
    # captured once per (batch shape, schedule state) after a few eager steps and replayed; the
    # host then issues one graph launch per step instead of ~230 kernel launches (at B = 10 the
    # eager step is bound by the host's launch rate).  Noam's rate and Adam's step counts live in
    # device tensors (capturable Adam); the non-finite-loss flag is read one step late from pinned
    # memory.  Data-parallel: the RCCL all-reduces (stage buckets on the side stream, statistic
    # sums in line) are recorded with the step -- fork and join of the side stream are event edges
    # inside the capture -- so every rank replays its own graph and the collectives meet on the
    # wire as in the eager step (same sequence, same sizes, whichever ranks replay or run eagerly).
    # Falls back to the eager path for gradient accumulation > 1 and on any carrier but the library
    # communicator (distributed.capturable).
    GRAPH_WARMUP = 3

    def _graph_key(self, batch):
        hp = self.hparams
        wavs, lens = batch.sig
        nrm = self.modules["normalize"] if "normalize" in self.modules else None
        upd = nrm is not None and hp.epoch_counter.current < getattr(nrm, "update_until_epoch", 0)
        # what the captured step does also depends on host state: the target-token shape (utility
        # branch), and -- under the epoch-parity schedule -- on whether a frozen parameter already has
        # Adam moments (torch 1.10's zero-filled gradients keep such a parameter moving, a stateless one
        # stays out of the update: _zero_grads_of_frozen), so both are part of the key
        st = self.optimizer.state if self.optimizer is not None else {}
        params = list(self.modules.ConvAE.parameters())
        tok = batch.tokens_bos[0]
        return (tuple(wavs.shape), str(wavs.dtype), hp.model_type, float(hp.recon_loss_weight),
                float(hp.sex_loss_weight), float(hp.utility_loss_weight),
                float(getattr(hp, "confusion_loss_weight", 0.0)), bool(upd),
                tuple(p.requires_grad for p in params),
                tuple(bool(st.get(p)) for p in params if not p.requires_grad),
                tuple(tok.shape), getattr(self, "asr_brain", None) is not None)

    def _step_core(self, batch):
        predictions = self.compute_forward(batch, Stage.TRAIN)
        loss = self.compute_objectives(predictions, batch, Stage.TRAIN)
        loss.backward()
        if getattr(self.hparams, "epoch_parity_schedule", False):
            self._zero_grads_of_frozen()
        self.check_gradients(loss)
        self.optimizer.step()
        return loss.detach()

    def _fit_batch_graph(self, batch):
        batch = batch.to(self.device)
        key = self._graph_key(batch)
        graphs = self.__dict__.setdefault("_graphs", {})
        ent = graphs.get(key)
        if ent is None:
            ent = graphs[key] = {"seen": 0}
        if "graph" not in ent:
            ent["seen"] += 1
            if ent["seen"] <= self.GRAPH_WARMUP:            # eager (also builds every cache / pool)
                self.optimizer.zero_grad(set_to_none=True)  # (static gradients of another graph)
                loss = self._step_core(batch)
                self.optimizer.zero_grad()
                self.hparams.noam_annealing(self.optimizer)
                return loss
            wavs, lens = batch.sig
            ent["wav"], ent["lens"] = wavs.clone(), lens.clone()
            ent["gender"] = batch.gender.clone()
            # the target tokens are part of the static batch (the utility branch decodes them); the
            # utterance ids ride along for the hooks that log them
            tok, tok_lens = batch.tokens_bos
            ent["tok"], ent["tok_lens"] = tok.clone(), tok_lens.clone()
            ent["batch"] = Batch(ent["wav"], ent["lens"], ent["gender"], tokens_bos=ent["tok"], ids=batch.id)
            ent["batch"].tokens_bos = (ent["tok"], ent["tok_lens"])
            torch.cuda.synchronize()
            self.nonfinite_count += self._poll_nonfinite(wait=True)      # nothing pending across the capture
            g = torch.cuda.CUDAGraph()
            self.optimizer.zero_grad(set_to_none=True)
            with torch.cuda.graph(g):
                ent["loss"] = self._step_core(ent["batch"])
            ent["graph"] = g
            # the capture itself does not execute the step: replay it for this batch below
        wavs, lens = batch.sig
        ent["wav"].copy_(wavs, non_blocking=True)
        ent["lens"].copy_(lens, non_blocking=True)
        ent["gender"].copy_(batch.gender, non_blocking=True)
        ent["tok"].copy_(batch.tokens_bos[0], non_blocking=True)
        ent["tok_lens"].copy_(batch.tokens_bos[1], non_blocking=True)
        ent["batch"].id = batch.id
        bad = self._poll_nonfinite()                        # counter copies of earlier replays
        if bad:
            self.nonfinite_count += bad
            if self.nonfinite_count > self.nonfinite_patience:
                raise ValueError("Loss is not finite and patience is exhausted.")
        ent["graph"].replay()
        st = getattr(self, "_finite_state", None)
        if st is not None:
            st["event"].record()
            st["pending"] = True
        self.hparams.noam_annealing(self.optimizer)
        return ent["loss"]

    def fit_batch(self, batch):
        self.apply_epoch_schedule()
        if (self.hip_graph and self.hparams.gradient_accumulation == 1 and sdist.capturable()
                and self.optimizer is not None):
            return self._fit_batch_graph(batch)
        predictions = self.compute_forward(batch, Stage.TRAIN)
        loss = self.compute_objectives(predictions, batch, Stage.TRAIN)
        (loss / self.hparams.gradient_accumulation).backward()
        if self.step % self.hparams.gradient_accumulation == 0:
            if getattr(self.hparams, "epoch_parity_schedule", False):
                self._zero_grads_of_frozen()
            self.check_gradients(loss)          # return value ignored, like the reference (:249-251)
            self.optimizer.step()
            self.optimizer.zero_grad()
            self.hparams.noam_annealing(self.optimizer)
        return loss.detach()

    def evaluate_batch(self, batch, stage):
        with torch.no_grad():
            predictions = self.compute_forward(batch, stage=stage)
            loss = self.compute_objectives(predictions, batch, stage=stage)
        return loss.detach()

    def load_external_classifier(self):
        """reference :261-269 reloads the pretrained x-vector classifier from absolute paths at
        every stage start; here it is an object handed over once: hparams["external_classifier"]
        (xvector.EncoderClassifier) or None (no ACC_external columns)."""
        ext = getattr(self.hparams, "external_classifier", None)
        if ext is not None:
            ext.eval()
            ext.to(self.device)
        return ext

    def on_stage_start(self, stage, epoch=None):
        from .metrics import AccuracyStats, SimilarityMetricsStats
        self.external_classifier = self.load_external_classifier()
        if stage != Stage.TRAIN:
            self.sex_classification_acc = AccuracyStats()
            self.sex_classification_acc_extern = AccuracyStats()
            self.sex_classification_acc_extern_orig = AccuracyStats()
            self.utility_similarity_aggregator = SimilarityMetricsStats()

    def on_evaluate_start(self, max_key=None, min_key=None):
        """reference :404-416: average the checkpoints selected by max_key / min_key into the model"""
        if self.checkpointer is None:
            return
        ckpts = self.checkpointer.find_checkpoints(max_key=max_key, min_key=min_key)
        if not ckpts:
            return
        sd = self.checkpointer.average_checkpoints(ckpts, recoverable_name="model", device=self.device)
        model = getattr(self.hparams, "model", None)
        if model is not None:
            model.load_state_dict(sd, strict=False)
            model.eval()

    def on_stage_end(self, stage, stage_loss, epoch=None):
        stats = {"loss": stage_loss}
        if stage == Stage.TRAIN:
            self.train_stats = stats
            return
        stats["ACC"] = self.sex_classification_acc.summarize()
        if self.external_classifier is not None:
            stats["ACC_external"] = self.sex_classification_acc_extern.summarize()
            stats["ACC_external_orig"] = self.sex_classification_acc_extern_orig.summarize()
        if self.utility_similarity_aggregator.denom:          # only with a frozen ASR in the loop (8f-2)
            stats["Utility_Retention"] = float(self.utility_similarity_aggregator.summarize())
        self.valid_stats = stats
        logger = getattr(self.hparams, "train_logger", None)
        if stage == Stage.VALID and sdist.if_main_process() and logger is not None:
            na = self.hparams.noam_annealing
            logger.log_stats(stats_meta={"epoch": epoch, "lr": na.current_lr, "steps": na.n_steps,
                                         "optimizer": self.optimizer.__class__.__name__},
                             train_stats=self.train_stats, valid_stats=stats)
        if stage == Stage.VALID and sdist.if_main_process() and self.checkpointer is not None:
            # reference :338-343
            keys = {k: stats[k] for k in ("ACC_external", "Utility_Retention") if k in stats}
            self.checkpointer.save_and_keep_only(self, epoch, dict(stats, **keys),
                                                 max_keys=[k for k in ("Utility_Retention",) if k in stats],
                                                 min_keys=[k for k in ("ACC_external",) if k in stats],
                                                 num_to_keep=5)


class FileTrainLogger:
    """speechbrain.utils.train_logger.FileTrainLogger line format
    (results/*/train_log.txt: "epoch: 1, lr: 2.60e-05, steps: 2854, optimizer: Adam - train loss: ...")."""

    def __init__(self, save_file):
        self.save_file = save_file

    @staticmethod
    def _fmt(d):
        out = []
        for k, v in d.items():
            if isinstance(v, float) and 1.0 < v < 100.0:      # speechbrain: no abs(); negatives go to %.2e
                v = f"{v:.2f}"
            elif isinstance(v, float):
                v = f"{v:.2e}"
            out.append(f"{k}: {v}")
        return ", ".join(out)

    def log_stats(self, stats_meta, train_stats=None, valid_stats=None, test_stats=None):
        s = self._fmt(stats_meta)
        for name, st in (("train", train_stats), ("valid", valid_stats), ("test", test_stats)):
            if st is not None:
                s += " - " + self._fmt({f"{name} {k}": v for k, v in st.items()})
        os.makedirs(os.path.dirname(os.path.abspath(self.save_file)), exist_ok=True)
        with open(self.save_file, "a") as f:
            f.write(s + "\n")
