"""CPU oracle for the ConvAE + gender-adversarial train step.

TEST INFRASTRUCTURE ONLY.  This package is a CPU restatement (plain PyTorch CPU ops /
numpy) of the reference algorithm on the hot path named by BASELINE.json:north_star.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it -- as the checker / the timed CPU baseline, never as the product path.
The product (``speech_anonymization_amd``) never imports anything from here and raises
when its HIP library is missing.

Parity pin status (see DESIGN.md "Oracle"):
  * convae.py, losses.py (ConvAutoencoder, TDNNSexClassifier, GradReverse, swish "GLU",
    CosineSimilarityLoss, ClusterMI/MILoss): PINNED -- checked bit-for-bit against the
    reference's own classes imported from /root/reference under a two-symbol speechbrain
    shim by oracle/gen_golden.py; the resulting vectors live in tests/golden/.
  * NoamScheduler: PINNED by the lr column of the reference's train_log.txt fixtures.
  * features.py (Fbank, InputNormalization), StatisticsPooling, xvector.py: the arithmetic
    lives in speechbrain (pinned at speechbrain/speechbrain@ff3bca4c, an EMPTY submodule in
    the reference checkout and not installed).  Restated from the published algorithm;
    "parity unpinned" except for the statistical pin of normalizer.ckpt and the state-dict
    layouts of normalizer.ckpt / classifier.ckpt.
"""
