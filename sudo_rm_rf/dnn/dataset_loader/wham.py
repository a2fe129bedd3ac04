from sudo_rm_rf_amd.feeder import *  # noqa: F401,F403
from sudo_rm_rf_amd.feeder import Dataset, WHAM_TASKS, EPS, normalize_tensor_wav  # noqa: F401
