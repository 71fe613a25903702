from syntalker_amd.process import SpacedDiffusion, _WrappedModel, space_timesteps  # noqa: F401
