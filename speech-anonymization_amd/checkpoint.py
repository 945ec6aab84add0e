"""Checkpointer with the on-disk layout of speechbrain's (convae.yaml:258-264):
``<checkpoints_dir>/CKPT+<date>+<time>+00/`` holding ``model.ckpt`` (torch state_dict of the
``model`` ModuleList, keys ``0.encoder.0.weight`` ...), ``normalizer.ckpt`` (dict with count /
glob_mean / glob_std / spk_dict_*), ``noam_scheduler.ckpt``, ``counter.ckpt`` (text int),
``brain.ckpt`` (text) and ``CKPT.yaml`` (meta).  CKPT.yaml is written as PLAIN yaml (floats), not
the pickled-tensor form some reference checkpoints contain."""
import os
import shutil
import time

import torch
import yaml


def _to_cpu(o):
    if torch.is_tensor(o):
        return o.detach().cpu()
    if isinstance(o, dict):
        return {k: _to_cpu(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return type(o)(_to_cpu(v) for v in o)
    return o


class Checkpointer:
    def __init__(self, checkpoints_dir, recoverables=None):
        self.checkpoints_dir = str(checkpoints_dir)
        self.recoverables = dict(recoverables or {})

    def add_recoverable(self, name, obj):
        self.recoverables[name] = obj

    def _list(self):
        if not os.path.isdir(self.checkpoints_dir):
            return []
        return sorted(d for d in os.listdir(self.checkpoints_dir) if d.startswith("CKPT+"))

    def save(self, brain=None, epoch=None, meta=None, num_to_keep=5):
        stamp = time.strftime("CKPT+%Y-%m-%d+%H-%M-%S")
        for n in range(100):                   # speechbrain's "+NN" suffix: unique within one second
            path = os.path.join(self.checkpoints_dir, f"{stamp}+{n:02d}")
            try:
                os.makedirs(path, exist_ok=False)
                break
            except FileExistsError:
                continue
        else:
            raise RuntimeError(f"could not create a unique checkpoint directory under {self.checkpoints_dir}")
        for key, obj in self.recoverables.items():
            fn = os.path.join(path, f"{key}.ckpt")
            if key == "counter":
                open(fn, "w").write(str(obj.current))
            elif hasattr(obj, "state_dict"):
                torch.save(_to_cpu(obj.state_dict()), fn)
        if brain is not None:
            open(os.path.join(path, "brain.ckpt"), "w").write(
                f"avg_train_loss: {float(brain.avg_train_loss)}\nstep: {brain.step}\n")
        m = {"unixtime": time.time(), "end-of-epoch": True}
        m.update({k: float(v) if isinstance(v, (int, float)) else v for k, v in (meta or {}).items()})
        if epoch is not None:
            m["epoch"] = epoch
        yaml.safe_dump(m, open(os.path.join(path, "CKPT.yaml"), "w"))
        for old in self._list()[:-num_to_keep]:
            shutil.rmtree(os.path.join(self.checkpoints_dir, old), ignore_errors=True)
        return path

    # ---- speechbrain Checkpointer surface the reference calls (speechbrain_convae_train.py:338-343,
    # 404-415): keep-only by metric, lookup by metric, checkpoint averaging
    def _meta(self, name):
        """CKPT.yaml of checkpoint `name` through the SAFE yaml loader (reference checkpoints hold
        pickled tensors there: those files are skipped, never unpickled)."""
        try:
            m = yaml.safe_load(open(os.path.join(self.checkpoints_dir, name, "CKPT.yaml")))
            return m if isinstance(m, dict) else {}
        except (OSError, yaml.YAMLError):
            return {}

    def find_checkpoints(self, max_key=None, min_key=None, max_num_checkpoints=None):
        """checkpoint directories, best first by `max_key` (highest) or `min_key` (lowest); most
        recent first when neither is given."""
        names = self._list()
        metas = {n: self._meta(n) for n in names}
        if max_key is not None:
            names = sorted((n for n in names if max_key in metas[n]), key=lambda n: -float(metas[n][max_key]))
        elif min_key is not None:
            names = sorted((n for n in names if min_key in metas[n]), key=lambda n: float(metas[n][min_key]))
        else:
            names = sorted(names, key=lambda n: -float(metas[n].get("unixtime", 0.0)))
        if max_num_checkpoints is not None:
            names = names[:max_num_checkpoints]
        return [os.path.join(self.checkpoints_dir, n) for n in names]

    def save_and_keep_only(self, brain=None, epoch=None, meta=None, max_keys=(), min_keys=(), num_to_keep=1):
        """save, then keep the union of: the `num_to_keep` best per key of max_keys / min_keys and
        the `num_to_keep` most recent; delete the rest."""
        path = self.save(brain, epoch, meta, num_to_keep=10 ** 9)
        keep = set(self.find_checkpoints(max_num_checkpoints=num_to_keep))
        for k in max_keys:
            keep.update(self.find_checkpoints(max_key=k, max_num_checkpoints=num_to_keep))
        for k in min_keys:
            keep.update(self.find_checkpoints(min_key=k, max_num_checkpoints=num_to_keep))
        for n in self._list():
            full = os.path.join(self.checkpoints_dir, n)
            if full not in keep:
                shutil.rmtree(full, ignore_errors=True)
        return path

    @staticmethod
    def average_checkpoints(ckpt_dirs, recoverable_name="model", device=None):
        """element-wise mean of `<recoverable_name>.ckpt` over the given checkpoints (integer
        tensors such as num_batches_tracked: floor of the mean), like
        speechbrain.utils.checkpoints.average_checkpoints."""
        avg, n = None, 0
        for d in ckpt_dirs:
            sd = torch.load(os.path.join(d, f"{recoverable_name}.ckpt"), map_location=device or "cpu",
                            weights_only=True)
            n += 1
            if avg is None:
                avg = {k: (v.clone().double() if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
            else:
                for k, v in sd.items():
                    avg[k] += v.double() if v.dtype.is_floating_point else v
        if avg is None:
            raise ValueError("no checkpoints to average")
        ref = torch.load(os.path.join(ckpt_dirs[0], f"{recoverable_name}.ckpt"), map_location=device or "cpu",
                         weights_only=True)
        return {k: ((v / n).to(ref[k].dtype) if v.dtype.is_floating_point else torch.div(v, n, rounding_mode="floor"))
                for k, v in avg.items()}

    def recover_if_possible(self, device=None, brain=None):
        """load the newest checkpoint into every recoverable (and brain.ckpt into `brain`);
        returns its path, or None when there is nothing to resume from."""
        ck = self._list()
        if not ck:
            return None
        path = os.path.join(self.checkpoints_dir, ck[-1])
        bfn = os.path.join(path, "brain.ckpt")
        if brain is not None and os.path.exists(bfn):
            kv = dict(ln.split(": ", 1) for ln in open(bfn).read().strip().splitlines() if ": " in ln)
            brain.avg_train_loss = float(kv.get("avg_train_loss", 0.0))
            brain.step = int(kv.get("step", 0))
        for key, obj in self.recoverables.items():
            fn = os.path.join(path, f"{key}.ckpt")
            if not os.path.exists(fn):
                continue
            if key == "counter":
                obj.current = int(open(fn).read().strip())
            else:
                obj.load_state_dict(torch.load(fn, map_location=device or "cpu", weights_only=True))
        if brain is not None and "optimizer" in self.recoverables:
            self._rebind_optimizer(brain)
        return path

    @staticmethod
    def _rebind_optimizer(brain):
        """optimizer.load_state_dict replaces the param_groups wholesale: a checkpoint written by an
        eager run carries lr as a Python float and capturable=False, one written in hipGraph mode a
        device tensor and capturable=True.  Put the groups back into the form THIS run's mode needs
        (and Noam's host copy of the rate), whichever mode wrote the checkpoint."""
        opt = brain.optimizer
        if opt is None:
            return
        for g in opt.param_groups:
            lr = float(g["lr"])
            opt._sa_host_lr = lr
            if getattr(brain, "hip_graph", False):
                g["lr"] = torch.tensor(lr, dtype=torch.float32, device=brain.device)
                if "capturable" in g:
                    g["capturable"] = True
            else:
                g["lr"] = lr
                if "capturable" in g:
                    g["capturable"] = False
        for p_, st in opt.state.items():                   # Adam's step counters follow the mode too
            if "step" in st and torch.is_tensor(st["step"]):
                want = brain.device if getattr(brain, "hip_graph", False) else torch.device("cpu")
                if st["step"].device != want:
                    st["step"] = st["step"].to(want)
