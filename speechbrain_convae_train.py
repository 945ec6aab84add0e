#!/usr/bin/env python3
"""Entry point with the reference's command line (speechbrain_convae_train.py:1-8,514-615):

    python speechbrain_convae_train.py speechbrain_configs/convae.yaml \
        --device cuda:0 --model_type convae --folder <output_dir> [--key value overrides]

    # data-parallel on one node (one process per GPU, RCCL):
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \
        speechbrain_convae_train.py speechbrain_configs/convae.yaml --distributed_launch

Extra options of this build: ``--synthetic N`` trains on N synthetic utterances per epoch instead
of CSV manifests (there is no dataset on the GPU box).  The frozen-ASR utility loss, the external
x-vector evaluation and WER are not part of this path."""
import os
import sys

import torch

import speech_anonymization_amd as pkg
from speech_anonymization_amd import brain as B, convae, data, distributed as sdist
from speech_anonymization_amd.yaml_loader import load_hyperpyyaml, parse_arguments


def main(argv):
    hparams_file, run_opts, overrides = parse_arguments(argv)
    synthetic = overrides.pop("synthetic", None)
    n_samples = int(overrides.pop("synthetic_samples", 161120))
    with open(hparams_file) as fin:
        hparams = load_hyperpyyaml(fin, overrides)
    rank, local_rank, world = sdist.ddp_init_group(run_opts)
    run_opts.setdefault("device", f"cuda:{local_rank}")
    if sdist.if_main_process():
        os.makedirs(hparams["output_folder"], exist_ok=True)

    if hparams["model_type"] == "convae":
        model = convae.ConvAutoencoder(precision=hparams.get("precision", "bf16x3"))
    elif hparams["model_type"] == "endtoend":              # reference :551-558 (EndToEnd.ConvReconstruction)
        from speech_anonymization_amd import endtoend, xvector
        clf = xvector.EncoderClassifier()
        ck = hparams.get("external_classifier_ckpt")       # directory with embedding_model.ckpt / classifier.ckpt
        if ck:
            clf.embedding_model.load_state_dict(torch.load(os.path.join(ck, "embedding_model.ckpt"),
                                                           map_location="cpu", weights_only=True))
            clf.classifier.load_state_dict(torch.load(os.path.join(ck, "classifier.ckpt"),
                                                      map_location="cpu", weights_only=True))
        model = endtoend.ConvReconstruction(clf, precision=hparams.get("precision", "bf16x3"))
    else:
        raise SystemExit("this path implements model_type convae and endtoend (SURVEY.md 8)")

    sa_brain = B.SexAnonymizationTraining(modules=hparams["modules"], opt_class=hparams["Adam"],
                                          hparams=hparams, run_opts=run_opts,
                                          checkpointer=hparams.get("checkpointer"))
    model = model.to(sa_brain.device)
    sa_brain.modules["ConvAE"] = model
    hparams["model"].append(sa_brain.modules["ConvAE"])      # ModuleList index 0 (reference :580)
    if hparams.get("asr_utility"):
        # reference :560-574 attaches the pretrained recogniser as sa_brain.asr_brain; its weights
        # are a hub fetch, so here `--asr_utility random` builds the same architecture with random
        # frozen weights (throughput-only) and a path loads this module's own state_dict
        from speech_anonymization_amd import asr
        rec = asr.ASR(dtype=torch.bfloat16 if sa_brain.device.type == "cuda" else None)
        if hparams["asr_utility"] != "random":
            rec.load_state_dict(torch.load(hparams["asr_utility"], map_location="cpu", weights_only=True))
        sa_brain.asr_brain = rec.to(sa_brain.device)
        if not hasattr(sa_brain.hparams, "loss_utility"):
            from speech_anonymization_amd import losses
            sa_brain.hparams.loss_utility = losses.CosineSimilarityLoss()

    bs = hparams["batch_size"]
    if synthetic:
        train = lambda: data.synthetic_dataset(int(synthetic), bs, n_samples, rank=rank, world=world)
        valid = lambda: data.synthetic_dataset(bs * world, bs, n_samples, seed=1, rank=rank, world=world)
    else:
        rep = {"data_root": hparams["data_folder"]}
        tr = data.CsvDataset(hparams["train_csv"], rep, hparams.get("sorting", "random"))
        va = data.CsvDataset(hparams["valid_csv"], rep, "ascending")
        shuffle = hparams["train_dataloader_opts"].get("shuffle", False) and hparams.get("sorting") == "random"
        train = lambda epoch: data.batches(tr, bs, shuffle, hparams["seed"], rank, world, epoch)
        valid = lambda epoch: data.batches(va, bs, False, 0, rank, world)
    if synthetic:
        train_, valid_ = train, valid
        train, valid = (lambda epoch: train_()), (lambda epoch: valid_())

    class Loader:
        """re-iterable; the epoch number seeds the shuffle (seed + epoch).  It is the epoch counter's
        own number, so a resumed run continues with the shuffle orders of the epochs it has not
        seen yet instead of replaying those of epochs 1..k."""

        def __init__(self, f, counter):
            self.f, self.counter = f, counter

        def __iter__(self):
            return iter(self.f(max(1, int(self.counter.current))))

    counter = hparams["epoch_counter"]
    sa_brain.fit(counter, Loader(train, counter), Loader(valid, counter))
    # the library-owned communicator (SA_COMM=lib) and its side stream go before the process group
    # that carried its id, and before interpreter teardown
    sdist.lib_comm_destroy()
    if sdist.world_size() > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1:])
