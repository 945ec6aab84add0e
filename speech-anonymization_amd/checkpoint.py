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
        return path
