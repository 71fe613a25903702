from syntalker_amd.process import (GaussianDiffusion, LossType, ModelMeanType, ModelVarType,  # noqa: F401
                                   get_named_beta_schedule)
