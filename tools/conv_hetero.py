"""Experiment: does co-residence of workgroups with DIFFERENT periods break the lock-step of the
one-tile conv kernel?  The batch is split in two halves launched concurrently on two streams, one
with 64-row tiles, one with 128-row tiles (timing only; outputs go to disjoint halves of y).
python tools/conv_hetero.py"""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from speech_anonymization_amd import _lib as L, ops
dev = torch.device("cuda:0")
lib = L.load()
B, L4 = 32, 20160
x = torch.randn(B, L4, 128, device=dev)
w = torch.randn(128, 128, 5, device=dev) * 0.05
wf = ops.pack_weights(w, "conv_fwd", torch.float32, L.BF16X3)
s1 = torch.rand(B, 128, device=dev) + 0.5
y = torch.empty(B, L4, 128, device=dev)
a_out = torch.empty(B, L4, 128, device=dev, dtype=torch.bfloat16)
st = torch.empty(B, 315, 128, 2, device=dev)
s2 = torch.cuda.Stream()


def args(b0, nb, tm):
    a = L.SaConvArgs()
    a.x, a.wp, a.y = x[b0:b0 + nb].data_ptr(), wf.data_ptr(), y[b0:b0 + nb].data_ptr()
    a.s1, a.t1, a.swish = s1[b0:b0 + nb].data_ptr(), s1[b0:b0 + nb].data_ptr(), 1
    a.stats, a.a_out = st[b0:b0 + nb].data_ptr(), a_out[b0:b0 + nb].data_ptr()
    a.B, a.Lin, a.Lout = nb, L4, L4
    a.taps = L.make_taps(ops.taps_conv(5, 1, 2))
    a.tile_rows = tm
    return a


def launch(a, stream):
    L.check(lib.sa_conv_gemm(L.BF16X3, 128, 128, 1, 1, C.byref(a), C.c_void_p(stream.cuda_stream)), "conv")


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


main = torch.cuda.current_stream()
full64, full128 = args(0, B, 64), args(0, B, 128)
h = B // 2


def split(tm_a, tm_b, nb_a=h):
    aa, bb = args(0, nb_a, tm_a), args(nb_a, B - nb_a, tm_b)

    def f():
        s2.wait_stream(main)
        launch(aa, main)
        launch(bb, s2)
        main.wait_stream(s2)
    return f


for rnd in range(2):
    print(f"one launch, 64-row tiles : {timeit(lambda: launch(full64, main)):7.1f} us")
    print(f"one launch, 128-row tiles: {timeit(lambda: launch(full128, main)):7.1f} us")
    print(f"two streams 64 | 64      : {timeit(split(64, 64)):7.1f} us")
    print(f"two streams 64 | 128     : {timeit(split(64, 128)):7.1f} us")
    print(f"two streams 128 | 128    : {timeit(split(128, 128)):7.1f} us")
    print(f"two streams 64 (20) | 128 (12): {timeit(split(64, 128, 20)):7.1f} us")
