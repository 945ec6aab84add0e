import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from tests.test_asr import _small
from speech_anonymization_amd.brain import Stage
dev = torch.device("cuda", 0)
brain = bench.build_brain(dev, "bf16x3", 3)
brain.asr_brain = _small(torch.bfloat16).to(dev)
brain.modules.ConvAE.pooling_noise = None
brain.hparams.epoch_counter.current = 10
brain.modules.eval()
brain.on_stage_start(Stage.VALID, 1)
def fin(name, t): print("   ", name, "finite", bool(torch.isfinite(t.float()).all()), "absmax", float(t.float().abs().max()))
for seed in (0, 1):
    batch = bench.synthetic_batch(3, seed, dev, n_samples=36 * 160 * 2 - 160)
    tok = torch.randint(3, 50, (3, 5), device=dev); tok[:, 0] = 1
    batch.tokens_bos = (tok, torch.ones(3, device=dev))
    print("seed", seed)
    with torch.no_grad():
        feats = brain.features(*batch.sig); fin("feats(before)", feats)
        recon, _ = brain.modules.ConvAE(feats); fin("recon(before)", recon)
        e_r, _ = brain.asr_brain.get_predictions(recon, batch.sig[1], tok, eval=True); fin("enc_r(before)", e_r)
    loss = brain.evaluate_batch(batch, Stage.VALID); fin("loss", loss)
    sc = brain.utility_similarity_aggregator.scores
    fin("scores", torch.stack(list(sc)))
    with torch.no_grad():
        feats = brain.features(*batch.sig); fin("feats", feats)
        recon, _ = brain.modules.ConvAE(feats); fin("recon", recon)
        e_r, _ = brain.asr_brain.get_predictions(recon, batch.sig[1], tok, eval=True); fin("enc_r", e_r)
        e_o, _ = brain.asr_brain.get_predictions(feats, batch.sig[1], tok, eval=True); fin("enc_o", e_o)
