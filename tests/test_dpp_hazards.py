"""CPU: the hand-scheduled inline asm of the sorting networks (bare v_min / v_max, v_pk_min/max_i16 under rewritten EXEC
masks, hand-counted s_nop — rank_stats.hpp, rank_hist.hpp, packed_sort_i16.hpp) is followed by compiler-generated DPP moves;
gfx950 needs 2 wait states between a VALU write of a VGPR and a DPP read of it and 5 after a VALU write of EXEC.  That the build keeps
them is a property of this toolchain's hazard handling around opaque asm text, so the final ISA of every K1 translation unit
is re-checked (tools/check_dpp_hazards.py), and the checker itself against a listing with known violations."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))

LISTING = '''
_Zfoo:
	v_min_f32 v3, v1, v2
	v_mov_b32_dpp v4, v3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1
	v_max_f32 v5, v1, v2
	s_nop 0
	v_mov_b32_dpp v6, v5 row_shr:1 row_mask:0xf bank_mask:0xf
	v_max_f32 v7, v1, v2
	s_nop 1
	v_mov_b32_dpp v8, v7 row_shr:1 row_mask:0xf bank_mask:0xf
	v_cmpx_lt_f32 exec, v1, v2
	s_nop 2
	v_mov_b32_dpp v9, v1 row_shr:1 row_mask:0xf bank_mask:0xf
	v_cmpx_lt_f32 exec, v1, v2
	s_nop 4
	v_add_f32_dpp v9, v1, v2 row_shr:1 row_mask:0xf bank_mask:0xf
	s_or_b64 exec, exec, s[0:1]
	v_mov_b32_dpp v13, v1 row_shr:1 row_mask:0xf bank_mask:0xf
	v_pk_min_i16 v[10:11], v1, v2
	v_mov_b32 v20, v21
	v_mov_b32_dpp v12, v11 row_mirror row_mask:0xf bank_mask:0xf
'''


def test_checker_finds_the_known_violations():
    import check_dpp_hazards as H
    bad = H.check(LISTING)
    lines = sorted(ln for _, ln, _ in bad)
    assert lines == [4, 7, 13, 21], bad          # 0 and 1 wait states after a VALU write, 3 after a VALU write of EXEC, a register
                                                 # pair; not: 2 / 5 wait states, EXEC restored by the scalar unit
    assert 'VALU write' in bad[0][2] and any('EXEC' in m for _, _, m in bad)


@pytest.mark.parametrize('dtype,all_tests', [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_k1_translation_units_keep_the_dpp_wait_states(dtype, all_tests):
    import check_dpp_hazards as H
    if not os.path.exists('/opt/rocm/bin/hipcc'):
        pytest.skip('no hipcc')
    text = open(H.compile_isa(dtype, all_tests)).read()
    n_dpp = sum(1 for ln in text.splitlines() if H.DPP_MOD.search(ln.split(';')[0]))
    assert n_dpp > 500                                # the listing really is the kernels' ISA
    assert H.check(text) == []
