#!/usr/bin/env python3
"""Does the sum of subset replays equal the whole-utterance replay? (piper_hip_voice_time_subset vs last_gpu_ms)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "piper-swift_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import katdata as kd
import piper_hip as ph
b = ph.HipBackend(0)
cfg = ph.voice_config("medium")
rt = ph.HipRuntime(b, cfg, ph.synthetic_blob(cfg, 1234))
ids = kd.FIXTURE_IDS * 8
dur = [3] * len(ids)
noise = kd.sym(1, (cfg.inter, 3 * len(ids)), 1.7)
rt.prepare(0, ids, dur, noise, 0.667)
for _ in range(5):
    rt.launch(0); rt.collect(0)
g = []
for _ in range(20):
    rt.launch(0); rt.collect(0); g.append(rt.last_gpu_ms(0))
print(f"whole utterance graph: {sum(g)/len(g)*1000:.1f} us")
tot = 0
for flt in ["", "enc", "flow", "dec.", "in_gate", "res_skip", "rel_attention", "qkv", "ffn1", "ffn2", "o_add", "conv_mfma", "embed", "expand"]:
    r = rt.time_subset(0, flt, 30)
    print(f"subset {flt!r:16}: {r}")
