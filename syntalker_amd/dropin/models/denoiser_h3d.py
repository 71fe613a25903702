from syntalker_amd.denoiser_h3d import MDM  # noqa: F401
