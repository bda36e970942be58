#!/usr/bin/env python3
"""Per-launch timing table of one utterance's schedule (HIP events around every launch, eager replay)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "piper-swift_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import katdata as kd  # noqa: E402
import piper_hip as ph  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--factor", type=int, default=8)
ap.add_argument("--quality", default="medium")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--precision", default="f32")
ap.add_argument("--batch", type=int, default=1)
args = ap.parse_args()
b = ph.HipBackend(0)
cfg = ph.voice_config(args.quality)
rt = ph.HipRuntime(b, cfg, ph.synthetic_blob(cfg, 1234))
rt.set_precision(args.precision)
ids = kd.FIXTURE_IDS * args.factor
dur = [3] * len(ids)
noise = kd.sym(1, (cfg.inter, 3 * len(ids)), 1.7)
if args.batch > 1:
    rt.prepare_batch(0, [(ids, dur, noise)] * args.batch, 0.667)
else:
    rt.prepare(0, ids, dur, noise, 0.667)
for _ in range(3):
    rt.launch(0); rt.collect(0)
g = []
for _ in range(10):
    rt.launch(0); rt.collect(0); g.append(rt.last_gpu_ms(0))
st = rt.profile(0, args.iters)
tot = sum(s["avg_us"] for s in st)
print(f"# {args.quality} {args.precision} factor={args.factor} batch={args.batch}: graph gpu_ms={sum(g)/len(g):.4f}  eager sum={tot:.1f} us  launches={len(st)}")
print(f"{'launch':44s} {'us':>9s} {'GFLOP':>9s} {'TFLOP/s':>9s} {'GB/s':>9s}")
for s in st:
    tf = s["flops"] / (s["avg_us"] * 1e-6) / 1e12 if s["avg_us"] > 0 else 0
    gb = s["bytes"] / (s["avg_us"] * 1e-6) / 1e9 if s["avg_us"] > 0 else 0
    print(f"{s['name']:44s} {s['avg_us']:9.2f} {s['flops']/1e9:9.4f} {tf:9.2f} {gb:9.1f}")
