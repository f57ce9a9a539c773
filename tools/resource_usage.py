#!/usr/bin/env python3
"""Registers, scratch and LDS of every kernel of the library, as the compiler reports them
(-Rpass-analysis=kernel-resource-usage on the build's own flags).  build_hip() keeps the remarks of each source beside its object
(smarties_amd/csrc/_obj/<source>.remarks.txt); this prints one line per kernel and is what tests/test_kernel_resources.py reads.

  python tools/resource_usage.py [--all]        (default: kernels with scratch, or above 128 registers)
"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "smarties_amd", "csrc", "_obj")
PAT = re.compile(r"Function Name: (\S+).*?TotalSGPRs: (\d+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?"
                 r"Occupancy \[waves/SIMD\]: (\d+).*?SGPRs Spill: (\d+).*?VGPRs Spill: (\d+).*?LDS Size \[bytes/block\]: (\d+)", re.S)


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [o.strip() for o in out[:len(names)]]


def short(name):
    """hl::kernel<args>(params) -> kernel<args>"""
    n = name.replace("void ", "").replace("hl::", "")
    depth = 0
    for i, ch in enumerate(n):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return n[:i]
    return n


def kernels(obj_dir=OBJ):
    """{source: [{name, sgpr, vgpr, agpr, scratch, occupancy, sgpr_spill, vgpr_spill, lds}]}"""
    res = {}
    for f in sorted(glob.glob(os.path.join(obj_dir, "*.remarks.txt"))):
        rows = PAT.findall(open(f).read())
        names = demangle([r[0] for r in rows]) if rows else []
        res[os.path.basename(f)[:-len(".remarks.txt")]] = [
            dict(name=short(n), sgpr=int(r[1]), vgpr=int(r[2]), agpr=int(r[3]), scratch=int(r[4]), occupancy=int(r[5]),
                 sgpr_spill=int(r[6]), vgpr_spill=int(r[7]), lds=int(r[8])) for n, r in zip(names, rows)]
    return res


def scratch_instructions(source, obj_dir=OBJ):
    """{kernel (short name): number of instructions of its ISA that touch scratch memory} for one source: the device code object is
    taken out of the host object (llvm-objdump --offloading) and disassembled.  A frame slot the compiler reserved and never
    touches shows up as ScratchSize > 0 with a count of 0 here."""
    import shutil
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    with tempfile.TemporaryDirectory() as td:
        o = os.path.join(td, "x.o")
        shutil.copy(os.path.join(obj_dir, source + ".o"), o)
        subprocess.run([os.path.join(llvm, "llvm-objdump"), "--offloading", o], capture_output=True, text=True, cwd=td, check=True)
        co = [f for f in glob.glob(o + ".*") if "amdgcn" in f]
        dis = subprocess.run([os.path.join(llvm, "llvm-objdump"), "-d", co[0]], capture_output=True, text=True, check=True).stdout
    counts, cur = {}, None
    for line in dis.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1)
            counts[cur] = 0
        elif cur and re.search(r"\bscratch_(load|store)|\bbuffer_(load|store)\S* .*\boffen\b", line):
            counts[cur] += 1
    names = list(counts)
    return {short(d): counts[n] for n, d in zip(names, demangle(names))}


if __name__ == "__main__":
    everything = "--all" in sys.argv
    for src, ks in kernels().items():
        for k in ks:
            if everything or k["scratch"] > 0 or k["vgpr"] + k["agpr"] > 128:
                print("%-12s v=%3d a=%3d scratch=%4d occ=%d sgprSpill=%3d lds=%6d  %s" % (
                    src, k["vgpr"], k["agpr"], k["scratch"], k["occupancy"], k["sgpr_spill"], k["lds"], k["name"][:120]))
