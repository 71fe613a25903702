"""Text-prompt (HumanML3D-style) denoiser — drop-in for reference models/denoiser_h3d.py:12.

Differences from denoiser.MDM, all in the per-clip conditioning (the step kernels are identical):
``input_process3`` (768 -> 512) is always present and folded into the input matrix; a learned
``uncon_text_embeddings`` row replaces the style vector when ``y['uncond']`` is set
(denoiser_h3d.py:116-124); ``y['uncond_audio']`` zeroes the waveform and the word ids (:173-180).
"""
from .denoiser import MDM as _Base


class MDM(_Base):
    variant = "h3d"
