"""Loader for the HyperPyYAML subset the reference's configs use (hyperpyyaml is not a
dependency): ``!ref <key>`` (also embedded: ``!ref <folder>/<seed>``), ``!new:pkg.Class`` with a
mapping / sequence / no body, ``!name:pkg.func`` (functools.partial), ``!apply:pkg.func``,
tuple-like strings ``(8, 10, 80)``, and ``--key value`` overrides
(speechbrain_configs/convae.yaml:11-14,78-85,140,203-206,253-295; loaded at
speechbrain_convae_train.py:516-518).

Class paths of the speechbrain / reference objects on the hot path resolve to this package's HIP
implementations (CLASS_MAP); objects of subsystems that are out of scope (ASR transformer, beam
search, SpecAugment, pretrainer ...) become ``Unavailable`` placeholders that raise on use, so a
reference YAML loads unmodified and the ConvAE path runs.
"""
import ast
import functools
import importlib
import re

import yaml

CLASS_MAP = {
    "speechbrain.lobes.features.Fbank": "speech_anonymization_amd.features.Fbank",
    "speechbrain.processing.features.InputNormalization": "speech_anonymization_amd.features.InputNormalization",
    "speechbrain.nnet.schedulers.NoamScheduler": "speech_anonymization_amd.brain.NoamScheduler",
    "speechbrain.utils.epoch_loop.EpochCounter": "speech_anonymization_amd.brain.EpochCounter",
    "speechbrain.utils.train_logger.FileTrainLogger": "speech_anonymization_amd.brain.FileTrainLogger",
    "speechbrain.utils.checkpoints.Checkpointer": "speech_anonymization_amd.checkpoint.Checkpointer",
    "torch.nn.MSELoss": "speech_anonymization_amd.losses.MSELoss",
    "torch.nn.L1Loss": "speech_anonymization_amd.losses.L1Loss",
    "torch.nn.NLLLoss": "speech_anonymization_amd.losses.NLLLoss",
    "utils.mi_loss.MILoss": "speech_anonymization_amd.losses.MILoss",
    "utils.cosine_similarity_loss.CosineSimilarityLoss": "speech_anonymization_amd.losses.CosineSimilarityLoss",
    "models.ConvAutoEncoder.ConvAutoencoder": "speech_anonymization_amd.convae.ConvAutoencoder",
}


class Unavailable:
    """Placeholder for an object of an out-of-scope subsystem."""

    def __init__(self, path, args=None):
        self._path, self._args = path, args

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        raise RuntimeError(f"{self._path} is outside the ConvAE hot path and is not provided "
                           f"(SURVEY.md 8f); attribute {name!r} was requested")

    def __call__(self, *a, **k):
        raise RuntimeError(f"{self._path} is outside the ConvAE hot path and is not provided")

    def __repr__(self):
        return f"<Unavailable {self._path}>"


class _Tagged:
    def __init__(self, kind, path, body):
        self.kind, self.path, self.body = kind, path, body


class _Ref:
    def __init__(self, expr):
        self.expr = expr


class _Loader(yaml.SafeLoader):
    pass


def _multi(kind):
    def construct(loader, suffix, node):
        if isinstance(node, yaml.MappingNode):
            body = loader.construct_mapping(node, deep=True)
        elif isinstance(node, yaml.SequenceNode):
            body = loader.construct_sequence(node, deep=True)
        else:
            body = loader.construct_scalar(node)
            body = None if body in ("", None) else body
            if isinstance(body, str) and body.startswith("-"):     # "-[]": a one-item sequence
                try:
                    body = yaml.safe_load("- " + body[1:])
                except Exception:
                    pass
        return _Tagged(kind, suffix, body)
    return construct


for _k in ("new", "name", "apply"):
    _Loader.add_multi_constructor(f"!{_k}:", _multi(_k))
_Loader.add_constructor("!ref", lambda l, n: _Ref(l.construct_scalar(n)))


def _resolve_class(path):
    path = CLASS_MAP.get(path, path)
    mod, _, name = path.rpartition(".")
    try:
        return getattr(importlib.import_module(mod), name)
    except Exception:
        return None


def _has_unavailable(v):
    if isinstance(v, Unavailable):
        return True
    if isinstance(v, dict):
        return any(_has_unavailable(x) for x in v.values())
    if isinstance(v, (list, tuple)):
        return any(_has_unavailable(x) for x in v)
    return False


def _literal(v):
    if isinstance(v, str) and re.fullmatch(r"\([^()]*\)", v.strip()):
        try:
            return ast.literal_eval(v)
        except Exception:
            return v
    return v


def load_hyperpyyaml(stream, overrides=None):
    text = stream.read() if hasattr(stream, "read") else stream
    raw = yaml.load(text, Loader=_Loader)
    if overrides:
        if isinstance(overrides, str):
            overrides = yaml.safe_load(overrides) or {}
        raw.update(overrides)
    done = {}

    def get(key):
        if key not in done:
            if key not in raw:
                raise KeyError(f"!ref <{key}> is not defined")
            done[key] = build(raw[key])
        return done[key]

    def ref(expr):
        m = re.fullmatch(r"<([^<>]+)>", expr.strip())
        if m:                                   # whole-value reference: keep the object
            return get(m.group(1))
        return re.sub(r"<([^<>]+)>", lambda mm: str(get(mm.group(1))), expr)

    def build(v):
        if isinstance(v, _Ref):
            return ref(v.expr)
        if isinstance(v, _Tagged):
            body = build(v.body)
            obj = _resolve_class(v.path)
            args, kwargs = [], {}
            if isinstance(body, dict):
                kwargs = body
            elif isinstance(body, list):
                args = body
            elif body is not None:
                args = [body]
            if obj is None:
                return Unavailable(v.path, (args, kwargs))
            if v.kind == "name":
                return functools.partial(obj, *args, **kwargs) if (args or kwargs) else obj
            try:
                return obj(*args, **kwargs)     # !new: and !apply:
            except Exception as e:              # e.g. an out-of-scope class that needs real data
                if _has_unavailable(body):
                    return Unavailable(v.path, e)
                if v.path in CLASS_MAP or v.path.startswith("torch."):
                    raise
                return Unavailable(v.path, e)
        if isinstance(v, dict):
            return {k: build(x) for k, x in v.items()}
        if isinstance(v, list):
            return [build(x) for x in v]
        return _literal(v)

    for key in list(raw.keys()):                # file order (seed first: convae.yaml:11-12)
        get(key)
    return {k: done[k] for k in raw}


def parse_arguments(argv):
    """sb.parse_arguments: positional yaml file, run options, and arbitrary --key value overrides."""
    run_keys = {"device", "distributed_launch", "distributed_backend", "max_grad_norm",
                "nonfinite_patience", "debug", "local_rank"}
    hparams_file, run_opts, overrides = None, {}, {}
    i = 0
    while i < len(argv):
        a = argv[i]
        if a.startswith("--"):
            key = a[2:]
            if "=" in key:
                key, val = key.split("=", 1)
            elif i + 1 < len(argv) and not argv[i + 1].startswith("--"):
                i += 1
                val = argv[i]
            else:
                val = "True"
            val = yaml.safe_load(val)
            (run_opts if key in run_keys else overrides)[key] = val
        elif hparams_file is None:
            hparams_file = a
        i += 1
    return hparams_file, run_opts, overrides
