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
        name = time.strftime("CKPT+%Y-%m-%d+%H-%M-%S+00")
        path = os.path.join(self.checkpoints_dir, name)
        os.makedirs(path, exist_ok=True)
        for key, obj in self.recoverables.items():
            fn = os.path.join(path, f"{key}.ckpt")
            if key == "counter":
                open(fn, "w").write(str(obj.current))
            elif hasattr(obj, "state_dict"):
                sd = obj.state_dict()
                sd = {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in sd.items()}
                torch.save(sd, fn)
        if brain is not None:
            open(os.path.join(path, "brain.ckpt"), "w").write(
                f"avg_train_loss: {brain.avg_train_loss}\nstep: {brain.step}\n")
        m = {"unixtime": time.time(), "end-of-epoch": True}
        m.update({k: float(v) if isinstance(v, (int, float)) else v for k, v in (meta or {}).items()})
        if epoch is not None:
            m["epoch"] = epoch
        yaml.safe_dump(m, open(os.path.join(path, "CKPT.yaml"), "w"))
        for old in self._list()[:-num_to_keep]:
            shutil.rmtree(os.path.join(self.checkpoints_dir, old), ignore_errors=True)
        return path

    def recover_if_possible(self, device=None):
        ck = self._list()
        if not ck:
            return None
        path = os.path.join(self.checkpoints_dir, ck[-1])
        for key, obj in self.recoverables.items():
            fn = os.path.join(path, f"{key}.ckpt")
            if not os.path.exists(fn):
                continue
            if key == "counter":
                obj.current = int(open(fn).read().strip())
            else:
                obj.load_state_dict(torch.load(fn, map_location=device or "cpu", weights_only=True))
        return path
