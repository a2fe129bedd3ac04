from sudo_rm_rf_amd.dnn.losses.sisdr import *  # noqa: F401,F403
from sudo_rm_rf_amd.dnn.losses.sisdr import PairwiseNegSDR, PITLossWrapper  # noqa: F401
