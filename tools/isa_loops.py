#!/usr/bin/env python
"""Per-kernel summary of the compiled ISA (gfx950, no GPU needed): for every kernel of a .hip file, the straight-line stretch
between the first and the last MFMA (the K loop as the compiler laid it out) with counts of MFMA, LDS, vector-memory, VALU,
scalar branches and full waits (`s_waitcnt` with a zero count) - the two patterns that cost this round the most were a branch
per element / per tile inside a loop and `request, wait for it, use it` sequences the compiler could not pipeline.

    python tools/isa_loops.py wav2lip_amd/csrc/conv_tp2.hip [-D...]
"""
import collections
import re
import subprocess
import sys
import tempfile
import os

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def main():
    src = sys.argv[1]
    extra = sys.argv[2:]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-I", os.path.join(root, "include"),
                        "-o", out, src] + extra, check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    for m in re.finditer(r"\n(_Z\w+):\s*; @\1\n", text):
        name = m.group(1)
        end = text.find(".Lfunc_end", m.end())
        body = [l.strip() for l in text[m.end():end].split("\n")]
        ins = [l.split(";")[0].strip() for l in body if l and not l.startswith((";", ".")) and not l.endswith(":")]
        ins = [i for i in ins if i]
        idx = [k for k, i in enumerate(ins) if i.startswith("v_mfma")]
        demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
        if not idx:
            continue
        seg = ins[idx[0]:idx[-1] + 1]
        c = collections.Counter()
        for i in seg:
            op = i.split()[0]
            if op.startswith("v_mfma"): c["mfma"] += 1
            elif op.startswith("ds_read") or op.startswith("ds_load"): c["lds_rd"] += 1
            elif op.startswith("ds_"): c["lds_wr"] += 1
            elif op.startswith(("buffer_", "global_", "scratch_", "flat_")): c["vmem"] += 1
            elif op.startswith("s_cbranch") or op == "s_branch": c["branch"] += 1
            elif op == "s_barrier": c["barrier"] += 1
            elif op == "s_waitcnt":
                c["wait"] += 1
                if re.search(r"(vmcnt|lgkmcnt)\(0\)", i): c["wait0"] += 1
            elif op.startswith("v_"): c["valu"] += 1
        print("%-70s mfma %3d  valu %4d  lds rd/wr %3d/%3d  vmem %3d  branches %3d  barriers %2d  waits %3d (to zero: %3d)" %
              (demangled[:70], c["mfma"], c["valu"], c["lds_rd"], c["lds_wr"], c["vmem"], c["branch"], c["barrier"], c["wait"], c["wait0"]))


if __name__ == "__main__":
    main()
