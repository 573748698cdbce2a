"""The shipped libmfpa.so contains no packed-fp32 instruction whose low lane selects a high half (`op_sel:[...]`).

That encoding (hipcc emits it freely for odd elements of a float pair) returned sporadically wrong low lanes next to MFMA
waves on this toolchain (profiles/r02_pk_fma_op_sel.md).  The check disassembles the gfx950 code objects of the binary that
travels to the GPU box -- not a re-compile -- so a future kernel that re-introduces the form fails here, on the CPU."""
import os

import pytest


@pytest.fixture(scope="module")
def lib_path():
    from musicfpaugment_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from musicfpaugment_amd.csrc.build import build
        build(verbose=False)
    return _lib.LIB_PATH


def test_no_packed_fp32_instruction_selects_a_high_half_into_its_low_lane(lib_path):
    from musicfpaugment_amd.csrc import isa_scan
    s = isa_scan.summary(lib_path)
    assert s["code_objects"] >= 11 and s["mfma"] > 1000 and s["packed_fp32"] > 100, s   # the scan really sees the kernels
    bad = isa_scan.packed_fp32_op_sel(lib_path)
    assert not bad, {k: (len(v), v[0]) for k, v in bad.items()}


def test_the_scanner_flags_the_form_when_it_is_there(tmp_path):
    """Self-test of the pattern on the two spellings llvm-objdump prints."""
    from musicfpaugment_amd.csrc import isa_scan
    hit = ["v_pk_fma_f32 v[0:1], v[120:121], v[68:69], v[0:1] op_sel:[0,1,0]",
           "v_pk_mul_f32 v[6:7], v[4:5], v[4:5] op_sel:[0,1] op_sel_hi:[1,0]"]
    ok = ["v_pk_fma_f32 v[2:3], s[38:39], v[16:17], v[2:3] op_sel_hi:[1,0,1]", "v_pk_add_f32 v[0:1], v[2:3], v[4:5]",
          "v_pk_mov_b32 v[0:1], v[2:3], v[4:5] op_sel:[1,0]", "v_mfma_f32_32x32x16_bf16 a[0:15], v[0:3], v[4:7], a[0:15]"]
    for ins in hit:
        assert isa_scan._PK_F32.search(ins) and isa_scan._OP_SEL.search(ins), ins
    for ins in ok:
        assert not (isa_scan._PK_F32.search(ins) and isa_scan._OP_SEL.search(ins)), ins
