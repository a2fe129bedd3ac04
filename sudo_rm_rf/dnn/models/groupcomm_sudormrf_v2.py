from sudo_rm_rf_amd.dnn.models.groupcomm_sudormrf_v2 import (_LayerNorm, GlobLN, ConvNormAct, NormAct,  # noqa: F401
                                                             DilatedConvNorm, UConvBlock, TAC,
                                                             GC_UConvBlock, GroupCommSudoRmRf)
