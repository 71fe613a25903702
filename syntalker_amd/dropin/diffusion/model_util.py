from syntalker_amd.process import SpacedDiffusion, create_gaussian_diffusion, create_model_and_diffusion, space_timesteps  # noqa: F401
