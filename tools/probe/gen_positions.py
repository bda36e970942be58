#!/usr/bin/env python3
"""Per-position durations of the generator's launches in a rocprofv3 kernel trace of bench.py (positions counted from conv_pre, the k = 7
short-row conv in front of the generator): python tools/probe/gen_positions.py gpurun_out/short_ab/c1 [c2 …]"""
import collections, csv, glob, re, sys

for d in sys.argv[1:]:
    f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    seqs = collections.defaultdict(list)
    pos = None
    for r in rows:
        n = r["Kernel_Name"]
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if "conv_short_kernel<7" in n:
            pos = 0
            continue
        m = re.search(r"(conv_win_kernel<[^>]*>|conv_pipe_kernel<[^>]*>|rb_pair_kernel<\d>|conv_cout1_\w+)", n)
        if pos is not None and m:
            g = (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
            seqs[(pos, m.group(1), g)].append(us)
            pos += 1
            if "cout1" in n:
                pos = None
    print(d)
    tot = 0.0
    for k in sorted(seqs):
        v = seqs[k]
        if len(v) > 40:
            print("   pos %d %-28s grid %-14s n %4d avg %7.2f us" % (k[0], k[1], k[2], len(v), sum(v) / len(v)))
            tot += sum(v) / len(v)
    print("   sum of the frequent ones: %.1f us" % tot)
