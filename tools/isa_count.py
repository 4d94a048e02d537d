#!/usr/bin/env python
"""Static instruction accounting of a HIP kernel for gfx950 (no GPU needed): compile one .hip file to ISA and count, per
basic block, the instructions by class - MFMA, other VALU, vector memory, LDS reads / writes, scalar, waits / barriers -
and report the innermost loop (the K-step) against everything outside it (the per-work-item prologue + epilogue).

rocprofv3's thread trace (--att) needs librocprof-trace-decoder, which this image does not ship; on a part where the fp32
MFMA shares its issue / execution resources with the rest of the wave's vector instructions (profiles/r01/
i_mfma_overlap_microbench.txt) the instruction COUNT per MFMA is the quantity that sets the rate, and this gives it exactly.

    python tools/isa_count.py wav2lip_amd/csrc/conv_wino2.hip [kernel-name-substring ...]
"""
import os
import re
import subprocess
import sys
import tempfile

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("ds_read") or op.startswith("ds_load") or op.startswith("ds_bpermute") or op.startswith("ds_permute"):
        return "lds_rd"
    if op.startswith("ds_"):
        return "lds_wr"
    if op in ("s_waitcnt", "s_barrier", "s_nop", "s_sleep") or op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    src = sys.argv[1]
    want = sys.argv[2:]
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-o", out, src],
                       check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    # split into functions
    funcs = re.split(r"\n(?=_Z\w+:|\w+_kernel\w*:)", text)
    for f in funcs:
        m = re.match(r"(\w+):", f)
        if not m or ".amdhsa_kernel" in f[:200]:
            continue
        name = m.group(1)
        if "kernel" not in name or (want and not any(w in name for w in want)):
            continue
        blocks, cur, label, in_loop = [], {}, "entry", False
        depth_of = {}
        for line in f.splitlines()[1:]:
            line = line.strip()
            lm = re.match(r"(\.LBB\d+_\d+):\s*(;.*)?", line)
            if lm:
                blocks.append((label, cur, depth_of.get(label, 0)))
                label, cur = lm.group(1), {}
                d = re.search(r"Depth=(\d+)", line)
                depth_of[label] = int(d.group(1)) if d else (1 if "in Loop" in line else 0)
                if "Inner Loop Header" in line or "Parent Loop" in line:
                    depth_of[label] = max(depth_of[label], 2 if "Parent Loop" in line else depth_of[label])
                continue
            if not line or line.startswith((";", ".", "//")) or line.endswith(":"):
                continue
            op = line.split()[0]
            c = classify(op)
            cur[c] = cur.get(c, 0) + 1
        blocks.append((label, cur, depth_of.get(label, 0)))
        tot = {}
        for _, b, _ in blocks:
            for k, v in b.items():
                tot[k] = tot.get(k, 0) + v
        # the K-loop = the block(s) holding most MFMAs
        kblock = max(blocks, key=lambda b: b[1].get("mfma", 0))
        kl = kblock[1]
        rest = {k: tot.get(k, 0) - kl.get(k, 0) for k in tot}
        keys = ["mfma", "valu", "vmem", "lds_rd", "lds_wr", "salu", "wait", "other"]
        print("== %s" % name)
        print("   %-34s" % "" + " ".join("%7s" % k for k in keys) + "   non-MFMA per MFMA")
        for tag, d in (("K-step loop body (%s)" % kblock[0], kl), ("outside it (per work item / per launch)", rest), ("total", tot)):
            nm = sum(v for k, v in d.items() if k != "mfma")
            print("   %-34s" % tag + " ".join("%7d" % d.get(k, 0) for k in keys) +
                  ("   %.2f" % (nm / d["mfma"]) if d.get("mfma") else ""))


if __name__ == "__main__":
    main()
