"""syntalker_amd — MI355X-native diffusion denoising hot path for SynTalker.

Host side (Python) mirrors the reference's two seams:
  * model factory      ``MDM(args)``                    (reference models/denoiser.py:12)
  * diffusion factory  ``create_gaussian_diffusion()``  (reference diffusion/model_util.py:8)
and calls hand-written HIP kernels for gfx950 through the C ABI declared in
``include/syn_hip.h`` (built into ``syntalker_amd/csrc/libsyn_hip.so``).
"""
__version__ = "0.1.0"
