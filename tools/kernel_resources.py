#!/usr/bin/env python3
"""Registers, static LDS and occupancy of every kernel instantiation, from the compiler's own metadata.

Compiles each csrc/*.hip for gfx950 to assembly (device side only, no GPU needed) and reads the `; NumVgprs / ; NumAgprs /
; ScratchSize / ; Occupancy / ; LDSByteSize` comment block LLVM emits per kernel. Dynamic LDS (the window kernels') is chosen
at launch and listed in DESIGN.md; `Occupancy` is waves per SIMD allowed by registers and static LDS alone.
usage: kernel_resources.py <tag>   → profiles/<tag>_kernel_resources.md
"""
import glob
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "piper-swift_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-w", "-S", "--cuda-device-only", "-x", "hip"]


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return p.stdout.splitlines()


def one(src):
    out = f"/tmp/kres_{os.path.basename(src)}.s"
    if not (os.path.exists(out) and os.path.getmtime(out) > max(os.path.getmtime(f) for f in glob.glob(os.path.join(CSRC, "*")))):
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, src, "-o", out], check=True, capture_output=True)
    rows, cur = [], None
    for ln in open(out):
        m = re.match(r"^(_Z\w+):\s*; @", ln)
        if m:
            cur = {"sym": m.group(1)}
        elif cur is not None:
            m = re.match(r"^; (NumVgprs|NumAgprs|TotalNumVgprs|ScratchSize|Occupancy|LDSByteSize|TotalNumSgprs|codeLenInByte):? =? ?(\d+)", ln)
            if m:
                cur[m.group(1)] = int(m.group(2))
                if m.group(1) == "Occupancy":
                    rows.append(cur)
                    cur = None
    return os.path.basename(src), rows


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r2"
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    with ThreadPoolExecutor(4) as ex:
        res = list(ex.map(one, srcs))
    lines = [f"# Kernel resources ({tag}) — hipcc -O3 --offload-arch=gfx950, from the compiler's per-kernel metadata", "",
             "`occ` = waves per SIMD allowed by registers and static LDS (gfx950: 512 VGPRs per SIMD lane, ≤ 8 waves). The window kernels",
             "(`conv_win`, `conv_pipe`, `rb_pair`, `conv_bf16`, `rel_attention_lds`) take DYNAMIC LDS at launch — their resident blocks",
             "per CU are bounded by that (DESIGN.md lists the sizes), not by this column.", ""]
    for src, rows in res:
        if not rows:
            continue
        names = demangle([r["sym"] for r in rows])
        kern = [(n, r) for n, r in zip(names, rows)]
        lines += [f"## {src} ({len(kern)} kernels)", "", "| kernel | VGPR | AGPR | SGPR | scratch B | static LDS B | occ | code B |", "|---|---:|---:|---:|---:|---:|---:|---:|"]
        if len(kern) > 40:  # the streamed conv templates: a summary row per kernel name is enough
            groups = {}
            for n, r in kern:
                groups.setdefault(re.sub(r"<.*", "", n.replace("void ", "").replace("ph::(anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("ph::detail::", "")), []).append(r)
            for g, rs in groups.items():
                f = lambda k: f"{min(r.get(k, 0) for r in rs)}–{max(r.get(k, 0) for r in rs)}"
                lines.append(f"| `{g}` × {len(rs)} instantiations | {f('NumVgprs')} | {f('NumAgprs')} | {f('TotalNumSgprs')} | {f('ScratchSize')} | {f('LDSByteSize')} | {f('Occupancy')} | {f('codeLenInByte')} |")
        else:
            for n, r in kern:
                n = n.replace("void ", "").replace("ph::(anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("ph::detail::", "")
                n = re.sub(r"\(.*", "", n)
                lines.append(f"| `{n[:90]}` | {r.get('NumVgprs', 0)} | {r.get('NumAgprs', 0)} | {r.get('TotalNumSgprs', 0)} | {r.get('ScratchSize', 0)} | {r.get('LDSByteSize', 0)} | "
                             f"{r.get('Occupancy', 0)} | {r.get('codeLenInByte', 0)} |")
        lines.append("")
    path = os.path.join(ROOT, "profiles", f"{tag}_kernel_resources.md")
    open(path, "w").write("\n".join(lines))
    print(path, sum(len(r) for _, r in res), "kernels")


if __name__ == "__main__":
    main()
