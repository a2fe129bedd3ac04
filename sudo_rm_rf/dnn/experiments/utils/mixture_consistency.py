from sudo_rm_rf_amd.dnn.experiments.utils.mixture_consistency import apply  # noqa: F401
