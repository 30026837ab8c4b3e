#!/usr/bin/env python3
"""Register / LDS / scratch budget of every kernel in cogdl_amd/csrc, from the compiler's own report
(`hipcc -Rpass-analysis=kernel-resource-usage`, gfx950) -- needs no GPU.  A kernel with ScratchSize > 0 spills: the first
thing to look at when it streams memory.  Usage: python tools/resource_audit.py [--all]   (default: own kernels only, the
rocPRIM instantiations are listed with --all)"""
import glob
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cogdl_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-DCOGDL_HIP_BUILD",
         "-Rpass-analysis=kernel-resource-usage"]
BLOCK = re.compile(r"Function Name: (\S+).*?SGPRs: (\d+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?"
                   r"Occupancy \[waves/SIMD\]: (\d+).*?LDS Size \[bytes/block\]: (\d+)", re.S)


def audit(src, tmp):
    out = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-c", src, "-o", os.path.join(tmp, os.path.basename(src) + ".o")],
                         capture_output=True, text=True)
    if out.returncode:
        raise SystemExit("%s: %s" % (src, out.stderr[-400:]))
    return [(os.path.basename(src),) + m.groups() for m in BLOCK.finditer(out.stderr)]


def main():
    show_all = "--all" in sys.argv
    with tempfile.TemporaryDirectory() as tmp, ThreadPoolExecutor(max(1, (os.cpu_count() or 2) // 2)) as pool:
        rows = [r for rs in pool.map(lambda s: audit(s, tmp), sorted(glob.glob(os.path.join(CSRC, "*.hip")))) for r in rs]
    names = subprocess.run(["c++filt"], input="\n".join(r[1] for r in rows), capture_output=True, text=True).stdout.splitlines()
    print("%-24s %5s %5s %5s %8s %4s %7s  %s" % ("file", "SGPR", "VGPR", "AGPR", "scratch", "occ", "LDS", "kernel"))
    seen = set()
    for (f, _, sg, vg, ag, sc, occ, lds), name in zip(rows, names):
        if (not show_all and "rocprim" in name) or (f, name) in seen:
            continue
        seen.add((f, name))
        name = re.sub(r"^void ", "", name).replace("cogdl::", "")
        print("%-24s %5s %5s %5s %8s %4s %7s  %s" % (f, sg, vg, ag, sc + (" !" if int(sc) else ""), occ, lds, name[:150]))


if __name__ == "__main__":
    main()
