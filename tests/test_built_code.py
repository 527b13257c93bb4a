"""The built HIP library's own machine code, checked without a GPU (tools/isa_hazards.py): round 4's seven-rays-per-chunk chase-tag failure was a live-range copy
that hipcc had placed AHEAD of the `s_or_b64 exec` re-converging the wavefront in a join block -- the lanes that had skipped the region (arenas that re-seed) never
got the flag position their observation reads (HISTORY.md).  The pattern is looked for in every kernel of the shipped code object."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import isa_hazards  # noqa: E402

# the faulty block as hipcc emitted it (commit 9b75351, -DLL_SEPMC_RAY_CHUNK=7, sepmc_step_kernel<1, false>), and the same block of the next commit's build
BAD = """
_Z4demov:
	s_and_saveexec_b64 s[2:3], s[6:7]
	s_cbranch_execz .LBB9_1160
; %bb.1147:
	v_mov_b32_e32 v0, 2
.LBB9_1160:                             ; %Flow3405
	v_mov_b64_e32 v[138:139], v[62:63]
	v_mov_b64_e32 v[136:137], v[60:61]
	s_or_b64 exec, exec, s[2:3]
	v_readlane_b32 s56, v252, 0
"""
GOOD = """
_Z4demov:
	s_and_saveexec_b64 s[2:3], s[6:7]
	s_cbranch_execz .LBB9_1160
; %bb.1147:
	v_mov_b32_e32 v0, 2
.LBB9_1160:                             ; %Flow3406
	s_or_b64 exec, exec, s[2:3]
	v_readlane_b32 s36, v252, 0
	v_mov_b64_e32 v[138:139], v[62:63]
	v_mov_b64_e32 v[136:137], v[60:61]
"""


def test_scanner_finds_a_copy_ahead_of_the_exec_restore():
    found = isa_hazards.exec_restore_scan(BAD)
    assert len(found) == 1 and found[0][1] == '.LBB9_1160' and len(found[0][3]) == 2, found
    assert isa_hazards.exec_restore_scan(GOOD) == []
    # the same two blocks as llvm-objdump --symbolize-operands prints them
    as_dump = lambda t: t.replace('_Z4demov:', '<_Z4demov>:').replace('.LBB9_1160:', '<L7>:').replace('.LBB9_1160', 'L7')
    assert len(isa_hazards.exec_restore_scan(as_dump(BAD))) == 1 and isa_hazards.exec_restore_scan(as_dump(GOOD)) == []


def test_shipped_code_object_has_no_valu_write_ahead_of_an_exec_restore():
    import __graft_entry__ as g
    lib = g.build_hip()
    assert isa_hazards.check_library(lib) == []
