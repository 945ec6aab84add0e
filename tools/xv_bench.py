import sys, time, torch
sys.path.insert(0, "/root/repo")
from speech_anonymization_amd import xvector as HX
hx, hc = HX.Xvector(pooling_noise=None).cuda().eval(), HX.Classifier(input_shape=[None, None, 128]).cuda().eval()
B, T = 32, 1008
feats = torch.randn(B, T, 80, device="cuda")
lens = torch.ones(B, device="cuda")
def run():
    with torch.no_grad():
        e = hx(feats, lens)
        return hc(e)
for _ in range(3): run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): run()
torch.cuda.synchronize(); print("xvector fwd ms", (time.perf_counter() - t0) / 10 * 1e3)
