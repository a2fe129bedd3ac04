"""CPU oracle for the SuDoRM-RF hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / the timed CPU baseline.  The product path
(``sudo_rm_rf_amd``) never imports this package and has no CPU fallback.

Contents
--------
schema.py        state_dict key/shape schema of the two reference models
                 (reference: improved_sudormrf.py:224-281,
                 groupcomm_sudormrf_v2.py:232-300, 343-418).
weights.py       deterministic numpy weight generator (every affine / slope /
                 bias perturbed so no parameter path is hidden by its default).
np_oracle.py     explicit-index numpy restatement (fp64 by default) of every
                 op on the path -- independent of ATen's conv semantics.
torch_oracle.py  functional torch-CPU restatement (same ATen kernels the
                 reference dispatches to); used for full-size parity and as the
                 ``cpu_baseline`` ("port") in bench.py.

Pinning: the reference holds no golden vectors for this path (SURVEY.md §4,
§8c), so parity is pinned against outputs of the reference itself, generated
in the build container by ``tools/make_golden.py`` (imports /root/reference)
and committed under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks
both restatements against those fixtures.
"""
