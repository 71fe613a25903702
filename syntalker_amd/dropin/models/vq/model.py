from syntalker_amd.rvqvae import RVQVAE  # noqa: F401   (inference-side drop-in for models/vq/model.py:RVQVAE)
