"""Throughput of the frozen-ASR utility branch (parity unpinned, random-init weights of
the reference architecture): the branch alone (original branch under no_grad + reconstruction branch
forward + data gradient back to the features) and the ConvAE train step with the branch attached
(loss = 0.1 recon + 0.9 sex + 0.2 utility).  Shape M, B utterances, U target tokens.

    python tools/asr_utility_bench.py [--batch 32] [--tokens 32] [--steps 10]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, steps, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--tokens", type=int, default=32)
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    import bench
    from speech_anonymization_amd import asr as A, losses
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    B, T, U = args.batch, 1008, args.tokens
    m = A.ASR(dtype=torch.bfloat16).to(dev)
    nparam = sum(p.numel() for p in m.parameters())
    feats = torch.randn(B, T, 80, device=dev)
    tok = torch.randint(3, 5000, (B, U), device=dev)
    tok[:, 0] = 1
    lens = torch.ones(B, device=dev)
    cos = losses.CosineSimilarityLoss()

    def branch():
        recon = (feats * 1.01).requires_grad_()
        A.utility_loss(m, cos, feats, recon, lens, tok).backward()
    ms_branch = timed(branch, args.steps)
    # flops of the branch: GEMM flops per token (2 x weights) x tokens, forward twice + data gradient once
    d, f, Te = 768, 3072, T // 4
    enc_tok, dec_tok = B * Te, B * U
    enc = 12 * (4 * d * d + 2 * d * f) * 2 * enc_tok + 12 * 4 * Te * d * enc_tok
    dec = 6 * (8 * d * d + 2 * d * f) * 2 * dec_tok + 6 * 2 * d * d * 2 * enc_tok
    src = 10240 * d * 2 * enc_tok
    cnn = (9 * 128 * (T // 2) * 40 + 9 * 128 * 256 * Te * 20 + 256 * 512 * Te * 20) * 2 * B
    flops = 3 * (enc + dec + src + cnn)

    brain = bench.build_brain(dev, "bf16x3", B)
    batch = bench.synthetic_batch(B, 0, dev)
    batch.tokens_bos = (tok, lens)
    brain.hparams.loss_utility = cos
    ms_plain = timed(lambda: brain.fit_batch(batch), args.steps)
    brain.asr_brain = m
    brain.hparams.utility_loss_weight = 0.2
    ms_with = timed(lambda: brain.fit_batch(batch), args.steps)
    # the same step as a hipGraph replay (run_opts hip_graph): the frozen branch is ~1 500 launches
    # of <= 30 us, i.e. host-issue-bound when launched eagerly
    ms_graph = None
    try:
        gb = bench.build_brain(dev, "bf16x3", B, hip_graph=True)
        gb.hparams.loss_utility = cos
        gb.asr_brain = m
        gb.hparams.utility_loss_weight = 0.2

        def gstep():
            gb.step += 1
            gb.fit_batch(batch)
        ms_graph = timed(gstep, args.steps, warmup=6)
    except Exception as e:                                      # capture is best-effort here
        print(f"graph replay failed: {type(e).__name__}: {e}", file=sys.stderr)
    print(json.dumps({
        "what": "frozen-ASR utility branch, random-init reference architecture (parity unpinned)",
        "params": nparam, "batch": B, "frames_per_utt": T, "target_tokens": U, "dtype": "bf16 (fp32 accumulate)",
        "branch_ms": ms_branch, "branch_gemm_tflops": flops / ms_branch / 1e9,
        "branch_frames_per_s": B * T / ms_branch * 1e3,
        "convae_step_ms": ms_plain, "convae_step_with_utility_ms": ms_with,
        "frames_per_s_with_utility": B * T / ms_with * 1e3,
        "convae_step_with_utility_graph_replay_ms": ms_graph,
        "frames_per_s_with_utility_graph_replay": (B * T / ms_graph * 1e3) if ms_graph else None,
        "hbm_allocated_gb": torch.cuda.max_memory_allocated() / 1e9}))


if __name__ == "__main__":
    main()
