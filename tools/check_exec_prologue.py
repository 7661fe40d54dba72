#!/usr/bin/env python3
"""Scans the gfx950 code objects of the HIP library for a code-generation defect of the ROCm 7.2 compiler that produced every
"fragile kernel" failure of rounds 1 - 4 (the code-generation fragility, root cause).

The defect.  After a divergent region the compiler re-enables the parked lanes with  s_or_b64 exec, exec, s[a:b]  at the top of the
join block; copies and spills the register allocator adds to a block must go BEHIND that instruction.  In kernels that spill many
SGPRs the block starts with reloads of spilled SGPRs (v_readlane_b32 / scratch loads), and LLVM then stops recognising the rest as
the block's prologue: the copies of a live-range split ( v_mov_b32 v79, v151 ... ) are placed BEFORE the s_or_b64.  They execute
with the narrow mask of the region that just ended -- or with EXEC = 0 -- so the lanes that were parked keep stale data in the
copy; when the value is copied back under the full mask (after a call, a loop, ...) those lanes get garbage.  A per-lane value that
is loop-invariant (an LDS address, a lane-dependent weight) is then wrong from the second iteration on, for some lanes, in some
builds: any edit, inlining decision or scheduling change moves the split points.

This tool finds the pattern in the disassembly: a join block (it restores EXEC with s_or_b64 exec, exec, s[..] / s_mov_b64 exec)
in which a VGPR is WRITTEN by a plain copy (v_mov_b32 / v_mov_b64 / v_accvgpr_*) or touched by scratch_load / scratch_store
before that restore, with nothing but SGPR reloads and scalar instructions in between.
    python3 tools/check_exec_prologue.py [object or library ...]      (default: build/csrc/*.o)
Exit code 1 when a kernel has a hit.  tests/test_codegen_guard.py runs it on the built objects."""
import glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
ADDR = re.compile(r"//\s*([0-9A-Fa-f]+):")
COPY = re.compile(r"^(v_mov_b32_e32|v_mov_b64_e32|v_accvgpr_write_b32|v_accvgpr_read_b32|v_accvgpr_mov_b32)\s+([va]\[?\d+)")
SCR = re.compile(r"^scratch_(load|store)_")
PAIR = re.compile(r"^(v_mov_b32_e32|v_mov_b64_e32|v_accvgpr_write_b32|v_accvgpr_read_b32|v_accvgpr_mov_b32)\s+([va]\[?\d+)[^,]*,\s*([va]\[?\d+)")
RESTORE = re.compile(r"^s_or_b64\s+exec,\s*exec,")          # (s_mov_b64 exec, s[..] is also how a region is ENTERED: not counted)
NARROW = re.compile(r"^(s_and_saveexec_b64|s_andn2_b64\s+exec|s_and_b64\s+exec|s_mov_b64\s+exec)")


def device_objects(path):
    """gfx950 code objects inside a host object / shared library (or the file itself when it is one)."""
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        r = subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fat], capture_output=True)
        if r.returncode == 0 and os.path.exists(fat) and os.path.getsize(fat) > 0:
            co = os.path.join(tmp, "dev.co")
            r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--unbundle", f"--input={fat}", f"--output={co}",
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], capture_output=True)
            if r.returncode == 0 and os.path.exists(co):
                out.append(subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout)
        else:
            out.append(subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", path], capture_output=True, text=True).stdout)
    return out


NEAR = 300


def reg_numbers(tok):
    m = re.match(r"^([va])\[(\d+):(\d+)\]", tok)
    if m:
        return m.group(1), set(range(int(m.group(2)), int(m.group(3)) + 1))
    m = re.match(r"^([va])\[?(\d+)", tok)
    return (m.group(1), {int(m.group(2))}) if m else (None, set())


def written_recently(ins, at, src):
    """is the first register of `src` the destination of one of the NEAR instructions before index `at`?"""
    kind, want = reg_numbers(src)
    want = {min(want)} if want else set()
    for _, txt in ins[max(0, at - NEAR):at]:
        parts = txt.split(None, 1)
        if len(parts) < 2 or parts[0].startswith(("ds_write", "scratch_store", "global_store", "flat_store", "s_", "buffer_store")):
            continue
        k, regs = reg_numbers(parts[1].split(",")[0].strip())
        if k == kind and regs & want:
            return True
    return False


def scan(asm):
    """[(function, address, copies)] of the suspicious join blocks."""
    hits = []
    funcs, cur = [], None
    for line in asm.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            cur = (m.group(1), [])
            funcs.append(cur)
        elif cur is not None and "//" in line:
            a = ADDR.search(line)
            if a:
                cur[1].append((int(a.group(1), 16), line.split("//")[0].strip()))
    for name, ins in funcs:
        targets = set()
        for k, (addr, txt) in enumerate(ins):
            m = re.match(r"^s_(cbranch_\w+|branch)\s+(\d+)", txt)
            if m:
                off = int(m.group(2))
                if off >= 32768:
                    off -= 65536
                targets.add(addr + 4 + 4 * off)
        starts = sorted(i for i, (addr, _) in enumerate(ins) if addr in targets)
        # every plain VGPR-to-VGPR copy of the function, as (dst, src) of its first register: a live-range split shows as a copy
        # A <- B in one place and the reverse copy B <- A in another
        allc = set()
        for _, txt in ins:
            m = PAIR.match(txt)
            if m:
                allc.add((m.group(2), m.group(3)))
        for i in starts:
            copies = []
            for kk, (addr, txt) in enumerate(ins[i:i + 48]):
                if RESTORE.match(txt):
                    if copies:
                        hits.append((name, ins[i][0], copies))
                    break
                m = PAIR.match(txt)
                if m:
                    # the way back exists (a save / restore pair) and the source is a long-lived value: it was not computed by the
                    # region that just ended (no write to it in the NEAR instructions before the copy), so the parked lanes hold data
                    if (m.group(3), m.group(2)) in allc and not written_recently(ins, i + kk, m.group(3)):
                        copies.append(txt)
                    continue
                if COPY.match(txt) or SCR.match(txt):
                    if SCR.match(txt):
                        # a store whose data register was written in THIS block in front of it FROM A CONSTANT OR A SCALAR (a value the block
                        # sets for the lanes that run it: a phi of the region) is code of the region, not a split copy
                        ms = re.match(r"^scratch_store_\w+\s+off,\s*([va]\[?\d+(?::\d+\])?)", txt)
                        local = False
                        if ms:
                            kind, want = reg_numbers(ms.group(1))
                            for _, t2 in ins[i:i + kk]:
                                parts = t2.split(None, 1)
                                if len(parts) == 2 and parts[0].startswith("v_") and not parts[0].startswith(("v_readlane", "v_cmp")):
                                    k2, r2 = reg_numbers(parts[1].split(",")[0].strip())
                                    if k2 == kind and r2 & want:
                                        # ... and only when that write takes an immediate or an SGPR: a VGPR / AGPR -> VGPR copy followed by a
                                        # spill of the copy in front of the EXEC restore IS the defect's signature (split copy, then store)
                                        ops = [o.strip() for o in parts[1].split(",")]
                                        local = not (len(ops) >= 2 and re.match(r"^[va]\[?\d", ops[1]))
                        if not local:
                            copies.append(txt)
                    continue
                if NARROW.match(txt):
                    break
                if txt.startswith(("v_readlane_b32", "v_writelane_b32", "s_")) and not txt.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_endpgm", "s_barrier")):
                    continue
                break
    return hits


def norm(txt):
    """instruction text with register numbers and offsets blanked: the key of the allow-list."""
    return re.sub(r"offset:\d+", "offset:#", re.sub(r"\b([vas])\[?\d+(:\d+\])?", r"\1#", txt))


def load_allow():
    import json
    path = os.path.join(ROOT, "tools", "exec_prologue_allow.json")
    return json.load(open(path))["sites"] if os.path.exists(path) else []


def unsafe_units():
    """translation units NOT compiled with the allocator that cannot produce the defect (csrc/Makefile: SAFE_RA)."""
    mk = open(os.path.join(ROOT, "safe_control_amd", "csrc", "Makefile")).read()
    safe = set(re.findall(r"FLAGS_(\w+)\s*:=\s*\$\(SAFE_RA\)", mk))
    return lambda path: os.path.basename(path).split(".")[0] not in safe


# kernels of the translation units csrc/Makefile lists as GUARDED (big kernels on the splitting allocator)
KERNEL_OF_UNIT = {"mpc_vtol_wave": "mpcvtol_wave_kernel", "mpc_vtol_ms": "mpcvtol_ms_kernel", "mpc_du_ms": "mpcdu_ms_kernel"}


def guarded_kernels():
    mk = open(os.path.join(ROOT, "safe_control_amd", "csrc", "Makefile")).read()
    m = re.search(r"^GUARDED\s*:=\s*(.*)$", mk, re.M)
    return [KERNEL_OF_UNIT[u] for u in (m.group(1).split() if m else []) if u in KERNEL_OF_UNIT]


def main():
    """Threshold: two or more long-lived copies / reloads in front of an EXEC restore everywhere; ONE in the translation units built
    with the splitting (greedy) allocator, unless the site is in tools/exec_prologue_allow.json (kernel, blanked instruction text,
    the review that cleared it) -- a NEW singleton fails."""
    paths = [a for a in sys.argv[1:] if not a.startswith("--")] or sorted(glob.glob(os.path.join(ROOT, "build", "csrc", "*.o")))
    greedy, allow = unsafe_units(), load_allow()
    total, allowed = 0, 0
    for p in paths:
        strict = greedy(p)
        is_lib = p.endswith(".so")                                            # a linked library: which unit a kernel came from is no longer visible --
        for asm in device_objects(p):                                         # threshold one for the kernels of the GUARDED units only (by name)
            for name, addr, copies in scan(asm):
                if is_lib:
                    strict = any(k in name for k in guarded_kernels())
                if len(copies) < (1 if strict else 2):                    # single moves in front of an s_or_b64 are ordinary code of the region with the basic
                    continue                                              # allocator, which never splits (measured); with the greedy one they are reviewed one by one
                dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                dem = re.sub(r"\(.*", "", dem.replace("(anonymous namespace)::", "")).replace("void sc::", "")
                if len(copies) == 1 and any(dem.startswith(a["kernel"]) and norm(copies[0]) == a["instruction"] for a in allow):
                    allowed += 1
                    continue
                print(f"{os.path.basename(p)}: {dem[:70]} @ {addr:#x}: {len(copies)} VGPR copies before the EXEC restore: {'; '.join(copies[:4])}  [{norm(copies[0])}]")
                total += 1
    print(f"{total} join block(s) with long-lived copies / reloads before the EXEC restore in {len(paths)} file(s) ({allowed} reviewed singleton(s) allowed)")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
