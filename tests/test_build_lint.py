"""The build's ISA lint (sudo_rm_rf_amd/build.py): objects containing the packed-fp32 operand form that is wrong on
gfx950 next to MFMAs (op_sel = 1 on src1; tools/probes/pk_opsel_probe.hip) are refused.  CPU-only: the lint is a text
scan of the device assembly hipcc leaves behind."""
import os

from sudo_rm_rf_amd import build

BAD = """
	v_pk_add_f32 v[46:47], v[46:47], s[40:41] op_sel:[0,1]
	v_pk_mul_f32 v[0:1], v[74:75], v[54:55] op_sel:[0,1] op_sel_hi:[0,0]
	v_pk_fma_f32 v[4:5], v[2:3], v[70:71], v[30:31] op_sel:[0,1,0] op_sel_hi:[1,0,1]
"""
GOOD = """
	v_pk_mul_f32 v[2:3], v[26:27], v[74:75] op_sel_hi:[1,0]
	v_pk_fma_f32 v[82:83], s[4:5], v[44:45], v[82:83] op_sel:[1,0,0]
	v_pk_fma_f32 v[40:41], v[70:71], v[0:1], v[110:111] op_sel:[0,0,1] op_sel_hi:[0,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]
	v_pk_add_f32 v[8:9], v[8:9], v[10:11] op_sel_hi:[1,0]
	v_cvt_pk_bf16_f32 v3, v0, v1
	; v_pk_add_f32 v[0:1], v[0:1], v[2:3] op_sel:[0,1]   (a comment is not an instruction)
"""


def test_lint_flags_src1_op_sel(tmp_path):
    p = tmp_path / "bad.s"
    p.write_text(BAD)
    hits = build.isa_lint(str(p))
    assert [h[0] for h in hits] == [2, 3, 4]


def test_lint_accepts_safe_forms(tmp_path):
    p = tmp_path / "good.s"
    p.write_text(GOOD)
    assert build.isa_lint(str(p)) == []


def test_every_source_is_listed_and_gemm_files_build_without_slp():
    listed = set(build.SOURCES)
    on_disk = {f for f in os.listdir(build.CSRC) if f.endswith(".hip")}
    assert listed == on_disk, (listed ^ on_disk)
    for f in ("srf_pwconv_bf16x3.hip", "srf_pwconv_wgrad.hip"):
        assert "-fno-slp-vectorize" in build.FILE_FLAGS[f]
