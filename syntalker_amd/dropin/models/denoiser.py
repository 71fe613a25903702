from syntalker_amd.denoiser import MDM, MDM_RVQ  # noqa: F401
