"""Import-path shims: put ``syntalker_amd/dropin`` FIRST on sys.path (or call ``install()``) and the
reference's drivers (`train.py:85-94`, `diffusion_rvqvae_trainer.py:17-19,185`, `h3d_diffusion_new_trainer.py`)
pick up this implementation through their own import statements:

    from diffusion.model_util import create_gaussian_diffusion
    from diffusion.resample import create_named_schedule_sampler
    from diffusion.cfg_sampler import ClassifierFreeSampleModel, ...
    getattr(__import__("models.denoiser", fromlist=["something"]), "MDM")(args)

The reference's own ``models`` / ``diffusion`` packages contain much more than the hot path, so
``install()`` only overrides the five hot-path modules inside ``sys.modules`` and leaves the rest alone.
"""
import importlib
import sys

_MAP = {
    "diffusion.model_util": "syntalker_amd.dropin.diffusion.model_util",
    "diffusion.respace": "syntalker_amd.dropin.diffusion.respace",
    "diffusion.gaussian_diffusion": "syntalker_amd.dropin.diffusion.gaussian_diffusion",
    "diffusion.resample": "syntalker_amd.dropin.diffusion.resample",
    "diffusion.cfg_sampler": "syntalker_amd.dropin.diffusion.cfg_sampler",
    "models.denoiser": "syntalker_amd.dropin.models.denoiser",
    "models.denoiser_h3d": "syntalker_amd.dropin.models.denoiser_h3d",
}


# opt-in: the sampling / evaluation drivers only run the RVQ-VAEs in eval() (diffusion_rvqvae_trainer.py:28,105-161); the
# reference's RVQ-VAE *training* script needs the trainable original, so this alias is not installed by default
_MAP_RVQ = {"models.vq.model": "syntalker_amd.dropin.models.vq.model"}


def install(rvqvae: bool = False):
    """Alias the hot-path modules under the reference's module names (rvqvae=True: also `models.vq.model.RVQVAE`)."""
    for ref_name, ours in {**_MAP, **(_MAP_RVQ if rvqvae else {})}.items():
        mod = importlib.import_module(ours)
        sys.modules[ref_name] = mod
        parent, _, leaf = ref_name.rpartition(".")
        if parent in sys.modules:
            setattr(sys.modules[parent], leaf, mod)
