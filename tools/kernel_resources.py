#!/usr/bin/env python3
"""Per-kernel register / spill / scratch / LDS figures from the gfx950 code objects of build/csrc/*.o
(llvm-readelf --notes of the unbundled device object).  Usage: python tools/kernel_resources.py [substring ...]"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def notes(obj):
    with tempfile.TemporaryDirectory() as tmp:
        out, fat = os.path.join(tmp, "dev.co"), os.path.join(tmp, "fat.bin")
        r = subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(fat) or os.path.getsize(fat) == 0:
            return ""
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--unbundle", f"--input={fat}", f"--output={out}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(out):
            return ""
        return subprocess.run([f"{LLVM}/llvm-readelf", "--notes", out], capture_output=True, text=True).stdout


def main():
    pats = sys.argv[1:]
    rows = []
    for f in sorted(os.listdir(os.path.join(ROOT, "build", "csrc"))):
        if not f.endswith(".o"):
            continue
        txt = notes(os.path.join(ROOT, "build", "csrc", f))
        for blk in txt.split("- .agpr_count:")[1:]:
            g = lambda k: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "?"])[1]
            name = g("name")
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            dem = dem.replace("(anonymous namespace)::", "")              # mpc_lin / mpc_gn keep their kernels in one
            dem = re.sub(r"\(.*", "", dem).replace("void sc::", "")
            if pats and not any(p in dem for p in pats):
                continue
            rows.append((f[:-2], dem[:70], g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_count"), g("sgpr_spill_count"),
                         g("private_segment_fixed_size"), g("group_segment_fixed_size")))
    print(f"{'tu':14s} {'kernel':70s} {'vgpr':>5s} {'vspill':>6s} {'sgpr':>5s} {'sspill':>6s} {'scratch':>7s} {'lds':>6s}")
    for r in rows:
        print(f"{r[0]:14s} {r[1]:70s} {r[2]:>5s} {r[3]:>6s} {r[4]:>5s} {r[5]:>6s} {r[6]:>7s} {r[7]:>6s}")


if __name__ == "__main__":
    main()
