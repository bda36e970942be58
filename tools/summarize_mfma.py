#!/usr/bin/env python3
"""MFMA-busy / wait shares per kernel family from the PMC passes of tools/run_pmc_mfma.sh.

usage: summarize_mfma.py <tag> <mfma pmc dir> <wave pmc dir> [<label>]
busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs · 256 CUs · kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 (summed over the
8 XCDs; reads high on dispatches shorter than ≈ 0.3 ms, MI355X_MICROARCH.md) — so the table also gives busy cycles per SIMD;
wait share = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (both in quad-cycles).
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_prof import family  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def read(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            e = acc[family(r["Kernel_Name"])][r["Counter_Name"]]
            e[0] += float(r["Counter_Value"])
            e[1] += 1
    return acc


def main():
    tag, d1, d2 = sys.argv[1], sys.argv[2], sys.argv[3]
    label = sys.argv[4] if len(sys.argv) > 4 else ""
    a, b = read(d1), read(d2)
    rows = []
    for k in a:
        busy, gui = a[k].get("SQ_VALU_MFMA_BUSY_CYCLES"), a[k].get("GRBM_GUI_ACTIVE")
        if not busy or not gui or busy[0] == 0:
            continue
        n = busy[1]
        cyc = gui[0] / n / 8.0
        per_simd = busy[0] / n / 1024.0
        wc, wi = b.get(k, {}).get("SQ_WAVE_CYCLES"), b.get(k, {}).get("SQ_WAIT_INST_ANY")
        rows.append({"kernel": k, "launches": n, "mfma_busy_cycles_per_simd": round(per_simd, 1), "kernel_cycles": round(cyc, 1),
                     "mfma_busy": round(per_simd / cyc, 4), "wait_inst_share": round(wi[0] / wc[0], 4) if wc and wi and wc[0] else None})
    rows.sort(key=lambda r: -r["mfma_busy_cycles_per_simd"] * r["launches"])
    lines = [f"# Matrix-pipe busy per kernel family ({tag}{', ' + label if label else ''})", "",
             "| kernel family | launches | MFMA-busy cycles / SIMD / launch | kernel cycles (GRBM_GUI_ACTIVE/8) | busy | issue-wait share |", "|---|---:|---:|---:|---:|---:|"]
    for r in rows:
        w = f"{100 * r['wait_inst_share']:.1f}%" if r["wait_inst_share"] is not None else "—"
        lines.append(f"| `{r['kernel']}` | {r['launches']} | {r['mfma_busy_cycles_per_simd']} | {r['kernel_cycles']} | {100 * r['mfma_busy']:.1f}% | {w} |")
    open(os.path.join(ROOT, "profiles", f"{tag}_mfma_busy.md"), "w").write("\n".join(lines) + "\n")
    json.dump(rows, open(os.path.join(ROOT, "profiles", f"{tag}_mfma_busy.json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
