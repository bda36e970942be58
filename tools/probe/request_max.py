#!/usr/bin/env python3
"""Which requests of bench.py's mixed-length stream are the slow ones, and what did their plan build cost?  gpurun -- python tools/probe/request_max.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "piper-swift_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import piper_hip as ph
import katdata as kd
cfg = ph.voice_config("medium")
backend = ph.HipBackend(0)
rt = ph.HipRuntime(backend, cfg, ph.synthetic_blob(cfg, 1234))
rs = np.random.RandomState(20240607)
rows = []
seen = set()
for i in range(200):
    Tn = int(np.clip(np.exp(rs.normal(np.log(90.0), 0.6)), 14, 400))
    rid = [kd.FIXTURE_IDS[j % 14] for j in range(Tn)]
    rdur = [int(x) for x in rs.randint(1, 6, size=Tn)]
    rnz = kd.sym(5000 + i, (cfg.inter, int(sum(rdur))), 1.7320508)
    a = time.perf_counter()
    rt.prepare(2, rid, rdur, rnz, 0.667)
    b = time.perf_counter()
    rt.launch(2)
    c = time.perf_counter()
    rt.collect(2)
    d = time.perf_counter()
    pi = rt.plan_info(2)
    key = (pi["bucket_t"], pi["bucket_f"])
    rows.append(((d - a) * 1e3, i, key, key not in seen, (b - a) * 1e3, (c - b) * 1e3, (d - c) * 1e3, rt.last_build_breakdown() if key not in seen else None))
    seen.add(key)
rows.sort(reverse=True)
for r in rows[:12]:
    print("%.2f ms  req %3d bucket %s new=%s prepare %.2f launch %.2f collect %.2f  %s" % r)
