"""Guards against the code-generation defect behind every "fragile kernel" failure of rounds 1 - 4 ("The code-generation
fragility: root cause"): the ROCm 7.2 compiler places the copies of a VGPR live-range split in front of the s_or_b64 that re-enables
the parked lanes of a join block, so lanes that sat out the region lose the value.  Two guards, both on the CPU:
  * the translation units with the big interior-point kernels are compiled with the basic VGPR allocator, which never splits a live
    range (csrc/Makefile: SAFE_RA) -- the defect cannot occur there by construction;
  * tools/check_exec_prologue.py scans the built gfx950 objects for the signature (two or more long-lived copies / reloads in front of
    an EXEC restore): none may be left, and the scanner must still recognise the defect in a recorded excerpt of a miscompiled build."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_big_kernels_are_built_with_the_allocator_that_does_not_split():
    mk = open(os.path.join(ROOT, "safe_control_amd", "csrc", "Makefile")).read()
    assert re.search(r"SAFE_RA\s*:=.*-vgpr-regalloc=basic", mk)
    for tu in ("mpc_cbf", "mpc_lin", "mpc_gn"):
        assert re.search(rf"FLAGS_{tu}\s*:=\s*\$\(SAFE_RA\)", mk), f"{tu}.hip must be compiled with SAFE_RA"
    assert "$(FLAGS_$*)" in mk


# the join block of the miscompiled KinematicBicycle2D kernel of round 4 (mpcgn_kernel<1, 10, false>, greedy allocator, continuation code
# added): loop-invariant per-lane registers are saved under the mask of the loop that just ended, EXEC is restored afterwards, and
# they are copied back under the full mask after the call
BAD = """
0000000000001000 <kernel>:
	s_cbranch_execz 3                                          // 000000001000: BF880003
	v_add_f64 v[0:1], v[0:1], v[2:3]                           // 000000001004: D2800000
	s_branch 65534                                             // 00000000100C: BF82FFFE
	s_nop 0                                                    // 000000001010: BF800000
	v_readlane_b32 s0, v253, 35                                // 000000001014: D2890000
	v_mov_b32_e32 v79, v151                                    // 00000000101C: 7E9E0397
	v_mov_b32_e32 v78, v150                                    // 000000001020: 7E9C0396
	v_readlane_b32 s0, v252, 54                                // 000000001024: D2890000
	s_or_b64 exec, exec, s[4:5]                                // 00000000102C: 87FE047E
	s_swappc_b64 s[30:31], s[0:1]                              // 000000001030: BE9E1E00
	v_mov_b32_e32 v151, v79                                    // 000000001034: 7F2E034F
	v_mov_b32_e32 v150, v78                                    // 000000001038: 7F2C034E
	s_endpgm                                                   // 00000000103C: BF810000
"""


def test_scanner_recognises_the_recorded_defect():
    import check_exec_prologue as C
    hits = C.scan(BAD)
    assert len(hits) == 1 and len(hits[0][2]) == 2, hits
    good = BAD.replace("	v_mov_b32_e32 v79, v151                                    // 00000000101C: 7E9E0397\n", "").replace(
        "	v_mov_b32_e32 v78, v150                                    // 000000001020: 7E9C0396\n", "")
    assert C.scan(good) == []


def test_built_objects_are_free_of_the_defect_signature():
    objs = [f for f in os.listdir(os.path.join(ROOT, "build", "csrc"))] if os.path.isdir(os.path.join(ROOT, "build", "csrc")) else []
    if not any(f.endswith(".o") for f in objs):
        pytest.skip("no built objects (run __graft_entry__.build() first)")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_exec_prologue.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
