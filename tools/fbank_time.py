"""Time of the Fbank launch (sa_fbank_kernel + the per-utterance kernels) at the bench shape.
  python tools/fbank_time.py [B]      (SA_HIP_LIB selects another build of the library)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import speech_anonymization_amd as pkg
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
fb = pkg.Fbank(16000, 400, 80).to(dev)
wav = (0.1 * torch.randn(B, 161120)).to(dev)
for _ in range(5):
    y = fb(wav)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    y = fb(wav)
e1.record()
torch.cuda.synchronize()
print(f"Fbank B={B}: {e0.elapsed_time(e1) / 50 * 1000:.1f} us per call, checksum {float(y.raw.double().sum()):.6f}")
