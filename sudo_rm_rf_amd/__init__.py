"""sudo_rm_rf_amd -- MI355X-native (gfx950) SuDoRM-RF forward hot path.

Layout: ``csrc/`` hand-written HIP kernels + the C ABI (include/sudormrf_hip.h), ``_lib`` ctypes
binding, ``engine`` plan/workspace cache, ``ops`` per-kernel wrappers, ``dnn/`` host-side mirror of the
reference's module interface (same names as ``sudo_rm_rf.dnn``).
"""
__version__ = "0.1.0"
