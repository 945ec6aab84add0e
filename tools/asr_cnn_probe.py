"""Where the frozen recogniser's convolutional front end spends its time: per block, forward (no_grad) and
forward + input gradient, bf16, B x 1008 x 80.   python tools/asr_cnn_probe.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from speech_anonymization_amd import asr as A
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
cnn = A.ConvolutionFrontEnd().to(dev).bfloat16()
x = torch.randn(B, 1008, 80, device=dev).bfloat16()


def timed(fn, n=10):
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


def block(i, xin):
    k, s = cnn.kernel_sizes[i], cnn.strides[i]
    h = A._reflect_pad1(xin) if k > 1 else xin
    w = cnn.w[i].to(h.dtype)
    if h.shape[-1] == 1:
        hin = h.reshape(h.shape[0], 1, h.shape[1], h.shape[2])
    else:
        hin, w = h.permute(0, 3, 1, 2), w.contiguous(memory_format=torch.channels_last)
    y = F.conv2d(hin, w, cnn.b[i].to(h.dtype), stride=s).permute(0, 2, 3, 1).contiguous()
    return A._LNLeaky.apply(y, cnn.ln_w[i], cnn.ln_b[i], 1e-5, 0.01)


with torch.no_grad():
    print(f"whole front end forward (no_grad): {timed(lambda: cnn(x)):8.1f} us")
xin = x.unsqueeze(-1)
for i in range(3):
    with torch.no_grad():
        t_f = timed(lambda: block(i, xin))
        out = block(i, xin)

    def fb():
        xi = xin.clone().requires_grad_()
        block(i, xi).sum().backward()
    print(f"block {i}: in {tuple(xin.shape)} -> out {tuple(out.shape)}   forward {t_f:8.1f} us   forward + input gradient {timed(fb):8.1f} us")
    xin = out


def fball():
    xi = x.clone().requires_grad_()
    cnn(xi).sum().backward()
print(f"whole front end forward + input gradient: {timed(fball):8.1f} us")
