"""speech_anonymization_amd -- MI355X-native ConvAE + gender-adversarial train step.

The per-batch compute of viswavi/speech-anonymization's ``speechbrain_convae_train.py`` path
(Fbank -> ConvAutoencoder fwd+bwd -> sex classifier -> losses) as hand-written HIP for gfx950
behind the reference's Python seams.  See DESIGN.md / INTEGRATION.md.
"""
from . import _lib                                    # noqa: F401
from ._lib import SaHipError, LIB_PATH                # noqa: F401
from .features import Fbank, InputNormalization      # noqa: F401
