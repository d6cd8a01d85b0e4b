#!/usr/bin/env python3
"""Compile mcorb_kernels.hip for gfx950 with --save-temps and print, for one kernel, the instruction mix per
basic block (VALU / LDS / SALU / VMEM counts) -- the static counterpart of the SQ_INSTS_VALU counter.

    python3 tools/isa_stats.py k_fast_cellsILi48ELi2 [--dump]      # substring of the mangled name
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = "/tmp/mcorb_isa"


def main():
    pat = sys.argv[1]
    dump = "--dump" in sys.argv
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(ROOT, "mc-slam_amd", "csrc", "mcorb_kernels.hip")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-mllvm", "-amdgpu-mfma-vgpr-form",
                           "-I" + os.path.join(ROOT, "mc-slam_amd", "csrc"), "-I" + os.path.join(ROOT, "include"),
                           "--save-temps", "-c", src, "-o", os.path.join(OUT, "k.o")], cwd=OUT)
    asm = open(os.path.join(OUT, "mcorb_kernels-hip-amdgcn-amd-amdhsa-gfx950.s")).read().split("\n")
    start = next(i for i, l in enumerate(asm) if re.match(r"^_Z\w*" + re.escape(pat) + r"\w*:", l))
    end = next(i for i in range(start, len(asm)) if "s_endpgm" in asm[i] and not any("s_endpgm" in x for x in asm[i + 1:i + 3]))
    end = max(i for i in range(start, len(asm)) if "s_endpgm" in asm[i] and i < start + 20000 and
              not any(re.match(r"^_Z\w+:", x) for x in asm[start + 1:i]))
    body = asm[start:end + 1]
    print(asm[start])
    blocks, cur = [], ["entry", 0, 0, 0, 0]
    for l in body[1:]:
        m = re.match(r"^(\.LBB\S+):\s*(;.*)?", l) or re.match(r"^; (%bb\.\d+):\s*(;.*)?", l)   # (fall-through blocks have no label, only a comment)
        if m:
            blocks.append(cur)
            cur = [m.group(1) + " " + (m.group(2) or ""), 0, 0, 0, 0]
            continue
        t = l.strip()
        if t.startswith("v_"):
            cur[1] += 1
        elif t.startswith("ds_"):
            cur[2] += 1
        elif t.startswith("s_"):
            cur[3] += 1
        elif t.startswith(("global_", "buffer_", "flat_", "scratch_")):
            cur[4] += 1
    blocks.append(cur)
    tot = [sum(b[k] for b in blocks) for k in range(1, 5)]
    for b in blocks:
        if b[1] + b[2] + b[4] >= 4:
            print("  %-70s VALU %4d  LDS %3d  SALU %4d  VMEM %3d" % (b[0][:70], b[1], b[2], b[3], b[4]))
    print("  total: VALU %d LDS %d SALU %d VMEM %d" % tuple(tot))
    for l in asm:
        if pat in l and (".vgpr_count" in l or ".sgpr_count" in l):
            print(l)
    if dump:
        open(os.path.join(OUT, "kernel.s"), "w").write("\n".join(body))
        print("written", os.path.join(OUT, "kernel.s"))


if __name__ == "__main__":
    main()
