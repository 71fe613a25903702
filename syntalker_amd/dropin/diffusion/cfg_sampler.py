from syntalker_amd.guidance import (ClassifierFreeSampleModel, ClassifierFreeSampleModel_Bodypart,  # noqa: F401
                                    TwoClassifierFreeSampleModel, TwoClassifierFreeSampleModel_Bodypart)
