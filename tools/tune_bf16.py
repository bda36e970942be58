#!/usr/bin/env python3
"""Sweep the (MTW, NTW, WM) tile configuration of conv_bf16_kernel over the generator's conv groups (tuning aid).

For each configuration the voice schedule is rebuilt with PIPER_HIP_BF16_CFG set and each group of launches
(`dec.s<u>.rb<j>.`, the ConvTransposes, conv_pre) is replayed as its own HIP graph (piper_hip_voice_time_subset)."""
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
ap.add_argument("--quality", default="high")
ap.add_argument("--batch", type=int, default=1)
args = ap.parse_args()
b = ph.HipBackend(0)
cfg = ph.voice_config(args.quality)
blob = ph.synthetic_blob(cfg, 1234)
ids = kd.FIXTURE_IDS * args.factor
dur = [3] * len(ids)
noise = kd.sym(1, (cfg.inter, 3 * len(ids)), 1.7)
groups = ["dec.conv_pre", "lrelu_convT"] + [f"dec.s{u}.rb{j}." for u in range(cfg.n_ups) for j in range(3)]
cfgs = ["default"] + [f"{m},{nt},{wm}" for wm in (1, 2, 4) for m in (1, 2) for nt in (1, 2, 4) if not (m == 2 and wm == 1)]
res = {}
for c in cfgs:
    if c == "default":
        os.environ.pop("PIPER_HIP_BF16_CFG", None)
    else:
        os.environ["PIPER_HIP_TUNING"] = "1"  # switches are honoured only with this set
        os.environ["PIPER_HIP_BF16_CFG"] = c
    rt = ph.HipRuntime(b, cfg, blob)
    rt.set_precision("bf16")
    if args.batch > 1:
        rt.prepare_batch(0, [(ids, dur, noise)] * args.batch, 0.667)
    else:
        rt.prepare(0, ids, dur, noise, 0.667)
    rt.launch(0); rt.collect(0)
    for g in groups:
        us, n, fl, _ = rt.time_subset(0, g, iters=20)
        res[(c, g)] = (us, n, fl)
    rt.close()
print(f"# {args.quality} factor={args.factor} batch={args.batch}: avg us per launch of each group (graph replay), by MTW,NTW,WM")
print(f"{'group':16s}" + "".join(f"{c:>9s}" for c in cfgs) + "   best")
for g in groups:
    row = [res[(c, g)][0] for c in cfgs]
    best = min(range(1, len(cfgs)), key=lambda i: row[i])
    print(f"{g:16s}" + "".join(f"{v:9.2f}" for v in row) + f"   {cfgs[best]} ({res[(cfgs[best], g)][2] / (row[best] * res[(cfgs[best], g)][1] * 1e-6) / 1e12:.0f} TF)")
