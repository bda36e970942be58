#!/usr/bin/env python3
"""Summarise rocprofv3 outputs into profiles/: kernel-trace stats and (optionally) the FETCH_SIZE / WRITE_SIZE PMC passes.

usage: summarize_prof.py <round tag> <stats dir> [<fetch pmc dir> <write pmc dir>] [--steps N]
Per-launch HBM traffic follows MI355X_MICROARCH.md §HBM: bytes = (2·FETCH_SIZE + WRITE_SIZE)·1024 — FETCH_SIZE counts
128-byte requests at 64 bytes on gfx950 (×2 correction), WRITE_SIZE is exact for streaming stores; both are in KiB.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find(d, suffix):
    m = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return m[0] if m else None


def short(name):
    name = name.replace("void ", "").replace("ph::(anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("ph::detail::", "")
    return name.split("(")[0]


def family(name):
    n = short(name)
    for f in ("conv_bf16_kernel", "pack_act_c8_kernel", "pack_mean3_c8_kernel", "rb_pair_bf16_kernel", "rb_pair_kernel", "flow_seam_kernel", "conv_cout1_wide_kernel", "conv_pipe_kernel", "conv_win_kernel", "rel_attention_lds_kernel", "rel_attention_mfma_kernel",
              "attention_block_kernel", "dds_layer_kernel", "dp_init_kernel", "dp_spline_kernel", "dp_final_kernel", "conv_k1_ln_kernel", "conv_k3_ln_kernel", "conv_k3_r8_kernel", "conv_k1_kernel", "conv_gate_kernel", "conv_cout1_split_kernel", "conv_short_kernel", "conv_stream_kernel", "conv_tile_kernel", "conv_small_cout_kernel", "conv_direct_kernel", "rel_attention_kernel",
              "add_layernorm_kernel", "mrf_mean_lrelu_kernel", "embed_kernel", "expand_noise_kernel", "pack_conv"):
        if n.startswith(f):
            return f
    return n


def pmc(dirname, counter):
    f = find(dirname, "counter_collection.csv")
    per = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            k = family(r["Kernel_Name"])
            per[k][0] += float(r["Counter_Value"])
            per[k][1] += 1
    return per


def main():
    tag, stats_dir = sys.argv[1], sys.argv[2]
    out = {"round": tag}
    rows = list(csv.DictReader(open(find(stats_dir, "kernel_stats.csv"))))
    fam = collections.defaultdict(lambda: [0, 0.0])
    rows = [r for r in rows if "spin_kernel" not in r["Name"]]  # timing scaffold of piper_hip_voice_profile, not the path
    for r in rows:
        k = family(r["Name"])
        fam[k][0] += int(r["Calls"])
        fam[k][1] += float(r["TotalDurationNs"])
    tot = sum(v[1] for v in fam.values())
    out["kernel_families"] = [{"kernel": k, "calls": v[0], "total_ms": round(v[1] / 1e6, 3), "avg_us": round(v[1] / v[0] / 1e3, 3),
                               "share": round(v[1] / tot, 4)} for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1])]
    lines = [f"# rocprofv3 --kernel-trace --stats summary ({tag})", "", "| kernel family | calls | total ms | avg µs | share |",
             "|---|---:|---:|---:|---:|"]
    for e in out["kernel_families"]:
        lines.append(f"| `{e['kernel']}` | {e['calls']} | {e['total_ms']} | {e['avg_us']} | {100 * e['share']:.1f}% |")
    lines += ["", "## per instantiation (top 25 by total time)", "", "| kernel | calls | avg µs | min µs | max µs | % |", "|---|---:|---:|---:|---:|---:|"]
    for r in rows[:25]:
        lines.append(f"| `{short(r['Name'])[:110]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.2f} | {float(r['MinNs']) / 1e3:.2f} | "
                     f"{float(r['MaxNs']) / 1e3:.2f} | {r['Percentage']} |")
    if len(sys.argv) >= 5 and not sys.argv[3].startswith("--"):
        fe, wr = pmc(sys.argv[3], "FETCH_SIZE"), pmc(sys.argv[4], "WRITE_SIZE")
        lines += ["", "## HBM traffic per launch (PMC, separate passes; bytes = (2·FETCH_SIZE + WRITE_SIZE)·1024)", "",
                  "| kernel family | launches | FETCH_SIZE KiB/launch | WRITE_SIZE KiB/launch | HBM MB/launch |", "|---|---:|---:|---:|---:|"]
        out["traffic"] = {}
        for k in sorted(set(fe) | set(wr), key=lambda k: -(fe[k][0] + wr[k][0])):
            nf, nw = max(fe[k][1], 1), max(wr[k][1], 1)
            f_kib, w_kib = fe[k][0] / nf, wr[k][0] / nw
            mb = (2 * f_kib + w_kib) * 1024 / 1e6
            out["traffic"][k] = {"launches": fe[k][1], "fetch_kib": round(f_kib, 1), "write_kib": round(w_kib, 1), "hbm_mb_per_launch": round(mb, 4)}
            lines.append(f"| `{k}` | {fe[k][1]} | {f_kib:.1f} | {w_kib:.1f} | {mb:.3f} |")
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    open(os.path.join(ROOT, "profiles", f"{tag}_rocprof_summary.md"), "w").write("\n".join(lines) + "\n")
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_rocprof_summary.json"), "w"), indent=1)
    print("\n".join(lines[:14]))


if __name__ == "__main__":
    main()
