#!/usr/bin/env python3
"""First (eager) run of one bucket in a fresh process, then a replay: rocprofv3 --kernel-trace --stats -- python3 tools/probe/first_run.py [ids] [frames_per_id]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "piper-swift_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import piper_hip as ph
import katdata as kd
T = int(sys.argv[1]) if len(sys.argv) > 1 else 400
fpi = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = ph.voice_config("medium")
rt = ph.HipRuntime(ph.HipBackend(0), cfg, ph.synthetic_blob(cfg, 1234))
ids = [kd.FIXTURE_IDS[j % 14] for j in range(T)]
dur = [fpi] * T
nz = kd.sym(7, (cfg.inter, sum(dur)), 1.7320508)
for rep in range(3):
    a = time.perf_counter(); rt.prepare(0, ids, dur, nz, 0.667); b = time.perf_counter(); rt.launch(0); c = time.perf_counter(); rt.collect(0); d = time.perf_counter()
    print("run %d: prepare %.2f launch %.2f collect %.2f ms  gpu %.3f ms" % (rep, (b - a) * 1e3, (c - b) * 1e3, (d - c) * 1e3, rt.last_gpu_ms(0)))
