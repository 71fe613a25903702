from syntalker_amd.resample import UniformSampler, create_named_schedule_sampler  # noqa: F401
