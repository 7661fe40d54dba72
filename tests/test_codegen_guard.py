"""Guards against the code-generation defect behind every "fragile kernel" failure of rounds 1 - 4 ("The code-generation
fragility: root cause"): the ROCm 7.2 compiler places the copies of a VGPR live-range split in front of the s_or_b64 that re-enables
the parked lanes of a join block, so lanes that sat out the region lose the value.  Two guards, both on the CPU:
  * the translation units with the big interior-point kernels are compiled with the basic VGPR allocator, which never splits a live
    range (csrc/Makefile: SAFE_RA) -- the defect cannot occur there by construction;
  * tools/check_exec_prologue.py scans the built gfx950 objects for the signature (two or more long-lived copies / reloads in front of
    an EXEC restore; ONE in the translation units that keep the splitting allocator, minus the reviewed sites of
    tools/exec_prologue_allow.json): none may be left, and the scanner must still recognise the defect in a recorded excerpt of a
    miscompiled build.  The Makefile runs the scanner on every link (a hit fails the build) and records the compiler version the
    two -mllvm flags were validated on; the GPU half of the guard is tests/test_codegen_guard_gpu.py."""
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


SINGLE = BAD.replace("	v_mov_b32_e32 v78, v150                                    // 000000001020: 7E9C0396\n", "").replace(
    "	v_mov_b32_e32 v150, v78                                    // 000000001038: 7F2C034E\n", "")


def test_a_new_singleton_fails_in_a_unit_built_with_the_splitting_allocator(tmp_path, monkeypatch):
    """Threshold ONE where the allocator can split (csrc/Makefile: everything without SAFE_RA), with an allow-list keyed by kernel and
    blanked instruction text: the reviewed site passes, any other single copy in front of an EXEC restore is a hit."""
    import check_exec_prologue as C
    hits = C.scan(SINGLE)
    assert len(hits) == 1 and len(hits[0][2]) == 1
    assert C.norm("scratch_store_dword off, v5, off offset:728") == "scratch_store_dword off, v#, off offset:#"
    assert C.norm("v_mov_b32_e32 v79, v151") == "v_mov_b32_e32 v#, v#"
    greedy = C.unsafe_units()
    assert greedy("build/csrc/mpc_vtol_wave.o") and greedy("build/csrc/mpc_vtol_ms.o") and greedy("build/csrc/cbf_qp_f32.o") and greedy("x/od_cbf_qp.o")
    assert not greedy("build/csrc/mpc_gn.o") and not greedy("build/csrc/mpc_cbf.o")
    allow = C.load_allow()
    assert allow and all(a["review"] and a["kernel"] and a["instruction"] for a in allow)
    # end to end on fake objects: the scanner's main() over a "disassembly" that carries the singleton
    monkeypatch.setattr(C, "device_objects", lambda path: [SINGLE])
    monkeypatch.setattr(sys, "argv", ["check", "build/csrc/mpc_vtol_wave.o"])
    assert C.main() == 1                                                  # greedy unit, site not on the list: fails
    monkeypatch.setattr(sys, "argv", ["check", "build/csrc/mpc_gn.o"])
    assert C.main() == 0                                                  # basic allocator: singletons are ordinary region code
    monkeypatch.setattr(C, "load_allow", lambda: [dict(kernel="kernel", instruction="v_mov_b32_e32 v#, v#", review="test")])
    monkeypatch.setattr(sys, "argv", ["check", "build/csrc/mpc_vtol_wave.o"])
    assert C.main() == 0                                                  # the same site once reviewed and listed


def test_a_split_copy_followed_by_a_spill_of_the_copy_is_still_a_hit():
    """The exemption for scratch stores in front of an EXEC restore (a store whose data the block itself sets is region code) holds only
    when that data comes from an immediate or a scalar register: `v_mov vT, vLong; scratch_store vT` -- a live-range split copy that is
    spilled right away -- in front of the s_or_b64 is the defect's signature and must be reported; `v_mov vT, 0; scratch_store vT` must not."""
    import check_exec_prologue as C
    head = """
0000000000001000 <kernel>:
	s_cbranch_execz 3                                          // 000000001000: BF880003
	v_add_f64 v[0:1], v[0:1], v[2:3]                           // 000000001004: D2800000
	s_branch 65534                                             // 00000000100C: BF82FFFE
	s_nop 0                                                    // 000000001010: BF800000
	v_readlane_b32 s0, v253, 35                                // 000000001014: D2890000
"""
    tail = """	scratch_store_dword off, v5, off offset:728                // 000000001020: DC700000
	s_or_b64 exec, exec, s[4:5]                                // 00000000102C: 87FE047E
	s_endpgm                                                   // 00000000103C: BF810000
"""
    split_then_spill = head + "	v_mov_b32_e32 v5, v151                                     // 00000000101C: 7E9E0397\n" + tail
    from_constant = head + "	v_mov_b32_e32 v5, 0                                        // 00000000101C: 7E9E0280\n" + tail
    from_scalar = head + "	v_mov_b32_e32 v5, s7                                       // 00000000101C: 7E9E0207\n" + tail
    from_agpr = head + "	v_accvgpr_read_b32 v5, a12                                 // 00000000101C: D3D84005\n" + tail
    assert len(C.scan(split_then_spill)) == 1 and len(C.scan(from_agpr)) == 1
    assert C.scan(from_constant) == [] and C.scan(from_scalar) == []


# the compiler build the two -mllvm flags of SAFE_RA (undocumented switches of LLVM's AMDGPU back end) were validated on
VALIDATED_HIPCC = ("HIP version: 7.2.26015-fc0010cf6a", "roc-7.2.0 26014 7b800a19466229b8479a78de19143dc33c3ab9b5")


def test_compiler_is_the_one_the_allocator_flags_were_validated_on():
    mk = open(os.path.join(ROOT, "safe_control_amd", "csrc", "Makefile")).read()
    assert "hipcc_version.txt" in mk and "$(SCANNER) $(OBJS)" in mk       # every link records the compiler and runs the scanner
    rec = os.path.join(ROOT, "build", "csrc", "hipcc_version.txt")
    if os.path.exists(rec):
        txt = open(rec).read()
    else:
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True)
        if r.returncode != 0:
            pytest.skip("no hipcc here and no recorded version")
        txt = r.stdout
    assert all(v in txt for v in VALIDATED_HIPCC), (
        "the library was built with a compiler other than the one `-mllvm -vgpr-regalloc=basic -mllvm -disable-machine-licm` and the "
        "scanner were validated on (ROCm 7.2.0, clang roc-7.2.0 26014): re-run tools/build_variants.sh + tests/test_codegen_guard_gpu.py "
        "+ the bitwise-resume tests on the new compiler, then update VALIDATED_HIPCC.  Found:\n" + txt)


def test_built_objects_are_free_of_the_defect_signature():
    """(The Makefile runs the same scan on every link.)  Objects when there are any, else the shipped library itself."""
    bdir = os.path.join(ROOT, "build", "csrc")
    objs = [f for f in os.listdir(bdir)] if os.path.isdir(bdir) else []
    args = []
    if not any(f.endswith(".o") for f in objs):
        lib = os.path.join(ROOT, "safe_control_amd", "lib", "libsafe_control_hip.so")
        if not os.path.exists(lib):
            pytest.skip("nothing built yet (run __graft_entry__.build() first)")
        args = [lib]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_exec_prologue.py")] + args, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:]
