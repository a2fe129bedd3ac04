from sudo_rm_rf_amd.dnn.models.improved_sudormrf import *  # noqa: F401,F403
from sudo_rm_rf_amd.dnn.models.improved_sudormrf import (_LayerNorm, GlobLN, ConvNormAct, NormAct,  # noqa: F401
                                                         DilatedConvNorm, UConvBlock, SuDORMRF)
