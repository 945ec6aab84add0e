"""Oracle: the whole reference train step on CPU.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates
speechbrain_convae_train.py:195-259 (fit_batch) -> :42-69 (compute_forward) -> :71-128
(compute_objectives) with the speechbrain pieces restated in oracle/features.py:
Fbank + InputNormalization are evaluated TWICE per step (the reference recomputes the
targets, :82-87, which also double-counts the batch in the normaliser's running stats),
``check_gradients`` = clip_grad_norm_(5.0), Adam(lr 1e-3, betas (0.9, 0.98), eps 1e-9),
then NoamScheduler(lr_initial 1.0, n_warmup 25000, model_size 768) AFTER optimizer.step().
"""
import torch
from . import features, losses
from .convae import ConvAutoencoder, numpy_params


def noam_lr(n_steps, lr_initial=1.0, n_warmup=25000, model_size=768):
    """speechbrain NoamScheduler._get_lr_scale; pinned by the lr column of the
    reference's train_log.txt (n=2854 -> 2.60e-05, n=25686 -> 2.25e-04)."""
    return lr_initial * model_size ** (-0.5) * min(n_steps ** (-0.5),
                                                   n_steps * n_warmup ** (-1.5))


class OracleTrainer:
    def __init__(self, params=None, recon_w=0.1, sex_w=0.9, util_w=0.0, conf_w=0.0,
                 recon_kind="l1", model_type="convae", max_grad_norm=5.0,
                 top_db_mode="utterance", threads=None, epoch_parity_schedule=False):
        if threads:
            torch.set_num_threads(threads)
        self.model = ConvAutoencoder()
        self.model.load_state_dict(params if params is not None else numpy_params())
        self.model.train()
        self.fbank = features.Fbank(top_db_mode=top_db_mode)
        self.normalize = features.InputNormalization()
        self.w = dict(recon=recon_w, sex=sex_w, utility=util_w, confusion=conf_w)
        self.recon_kind, self.model_type = recon_kind, model_type
        self.max_grad_norm = max_grad_norm
        self.opt = torch.optim.Adam(self.model.parameters(), lr=1e-3, betas=(0.9, 0.98),
                                    eps=1e-9)
        self.n_steps = 0
        self.epoch = 1
        self.epoch_parity_schedule = epoch_parity_schedule

    def apply_epoch_schedule(self):
        """HEAD's fit_batch preamble (speechbrain_convae_train.py:212-235): even epoch -> recon 0 /
        sex 0.8 / utility 0.2 / confusion 0, classifier frozen; odd epoch -> sex 0.5 only,
        everything but the classifier frozen.  (The utility term needs the frozen ASR, which is
        outside this path: its loss value is 0 here, the weight is still set.)"""
        joint = self.epoch % 2 == 0
        if joint:
            self.w.update(recon=0.0, sex=0.8, utility=0.2, confusion=0.0)
        else:
            self.w.update(recon=0.0, sex=0.5, utility=0.0)
        for name, p in self.model.named_parameters():
            p.requires_grad = ("sex_classifier" not in name) if joint else ("sex_classifier" in name)

    def feats(self, wavs, lens):
        f = self.fbank(wavs)
        f = self.normalize(f, lens, epoch=self.epoch)
        return features.pad_to_multiple(f, 36)

    def forward_loss(self, wavs, lens, gender):
        feats = self.feats(wavs, lens)                       # compute_forward
        recon, logp = self.model(feats)
        target = self.feats(wavs, lens)                      # compute_objectives (again)
        rl = losses.recon_loss(recon, target, self.recon_kind)
        sl = losses.sex_loss(logp, gender)
        cl = losses.confusion_loss(logp)
        loss = losses.total_loss(rl, sl, 0.0, cl, self.w, self.model_type)
        return loss, dict(recon=recon, logp=logp, feats=feats, target=target,
                          recon_loss=rl, sex_loss=sl, confusion_loss=cl)

    def fit_batch(self, wavs, lens, gender):
        if self.epoch_parity_schedule:
            self.apply_epoch_schedule()
        loss, aux = self.forward_loss(wavs, lens, gender)
        loss.backward()
        aux["grads"] = {k: (p.grad.detach().clone() if p.grad is not None else None)
                        for k, p in self.model.named_parameters()}
        aux["grad_norm"] = torch.nn.utils.clip_grad_norm_(
            [p for p in self.model.parameters() if p.grad is not None], self.max_grad_norm)
        self.opt.step()
        # torch 1.10 (the reference's pin, results/*/env.log): optimizer.zero_grad() defaults to
        # set_to_none=False, i.e. gradients stay allocated as zeros -- so a parameter frozen by the
        # epoch-parity schedule AFTER it has had a gradient is still updated by Adam (zero
        # gradient, non-zero moments).  Spelled out here because torch >= 2.0 changed the default.
        self.opt.zero_grad(set_to_none=False)
        self.n_steps += 1
        lr = noam_lr(self.n_steps)
        for g in self.opt.param_groups:
            g["lr"] = lr
        return loss.detach(), aux
