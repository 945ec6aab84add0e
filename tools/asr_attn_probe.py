"""Attention of the frozen recogniser (B x 8 heads x 252 frames x 96): the explicit GEMM + softmax form of
asr._Attention against torch's fused scaled_dot_product_attention on strided head views (no head-major copies).
Forward + input gradient, bf16.   python tools/asr_attn_probe.py [B]"""
import sys, time
import torch, torch.nn.functional as F
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
h, T, dh, d = 8, 252, 96, 768
x = torch.randn(B, T, 3 * d, device=dev, dtype=torch.bfloat16)
bias = torch.zeros(B, 1, 1, T, device=dev)
bias[1, :, :, 200:] = float("-inf")


def manual(qkv):
    q, k, v = qkv.view(B, T, 3, h, dh).permute(2, 0, 3, 1, 4).contiguous()
    s = torch.matmul(q, k.transpose(-1, -2)) + bias.to(q.dtype)
    return torch.matmul(torch.softmax(s, -1), v).transpose(1, 2).reshape(B, T, d)


def sdpa(qkv, mask=True):
    v5 = qkv.view(B, T, 3, h, dh)
    q, k, v = (v5[:, :, i].transpose(1, 2) for i in range(3))
    m = bias.to(q.dtype).expand(B, h, T, T) if mask else None
    return F.scaled_dot_product_attention(q, k, v, attn_mask=m, scale=1.0).transpose(1, 2).reshape(B, T, d)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000


def fb(f):
    def run():
        xx = x.clone().requires_grad_()
        f(xx).sum().backward()
    return run


print("flash", torch.backends.cuda.flash_sdp_enabled(), "mem_efficient", torch.backends.cuda.mem_efficient_sdp_enabled(),
      "math", torch.backends.cuda.math_sdp_enabled())
ref = manual(x)
for name, f in (("manual", manual), ("sdpa+mask", sdpa), ("sdpa no mask", lambda q: sdpa(q, False))):
    try:
        out = f(x)
        err = (out.float() - ref.float()).abs().max().item() if name != "sdpa no mask" else float("nan")
        with torch.no_grad():
            t_f = timed(lambda: f(x))
        t_fb = timed(fb(f))
        print(f"{name:14s} fwd {t_f:7.1f} us   fwd+bwd {t_fb:7.1f} us   max |diff| vs manual {err:.3e}", flush=True)
    except Exception as e:
        print(f"{name:14s} failed: {type(e).__name__}: {str(e)[:300]}", flush=True)
for be in ("FLASH_ATTENTION", "EFFICIENT_ATTENTION"):
    try:
        from torch.nn.attention import sdpa_kernel, SDPBackend
        with sdpa_kernel(getattr(SDPBackend, be)):
            f = (lambda q: sdpa(q, False)) if be == "FLASH_ATTENTION" else sdpa
            with torch.no_grad():
                t_f = timed(lambda: f(x))
            t_fb = timed(fb(f))
        print(f"{be:20s} fwd {t_f:7.1f} us   fwd+bwd {t_fb:7.1f} us", flush=True)
    except Exception as e:
        print(f"{be:20s} failed: {type(e).__name__}: {str(e)[:300]}", flush=True)
