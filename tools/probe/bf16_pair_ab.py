#!/usr/bin/env python3
"""A/B of the fused bf16 ResBlock1 pairs against the two-launch path (env PIPER_HIP_NO_RB_PAIR is read once per process):
   python tools/probe/bf16_pair_ab.py run out.npy   (in each environment), then   ... cmp a.npy b.npy"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "piper-swift_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

if sys.argv[1] == "run":
    import katdata as kd
    import piper_hip as ph
    b = ph.HipBackend(0)
    cfg = ph.voice_config("high")
    rt = ph.HipRuntime(b, cfg, ph.synthetic_blob(cfg, 1234))
    rt.set_precision("bf16")
    outs = []
    for factor, bucket_trick in ((8, 0), (3, 5)):
        ids = (kd.FIXTURE_IDS * factor)[:14 * factor - bucket_trick]
        dur = [3] * len(ids)
        noise = kd.sym(7 + factor, (cfg.inter, 3 * len(ids)), 1.7)
        rt.prepare(0, ids, dur, noise, 0.667)
        rt.launch(0)
        a = rt.collect(0)
        for _ in range(3):
            rt.launch(0); rt.collect(0)
        outs.append(a)
        print(f"factor {factor}: {a.size} samples, gpu_ms {rt.last_gpu_ms(0):.4f}, launches {len(rt.profile(0, 2))}")
    np.save(sys.argv[2], np.concatenate(outs))
else:
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    d = np.abs(a - b).max()
    snr = 10 * np.log10((a.astype(np.float64) ** 2).sum() / max(((a - b).astype(np.float64) ** 2).sum(), 1e-300))
    print(f"max|fused − unfused| = {d:.3e}, SNR {snr:.1f} dB, finite {np.isfinite(a).all() and np.isfinite(b).all()}")
    # both against the fp32 oracle on the second utterance (factor 3 minus 5 ids: a bucketed plan)
    import katdata as kd
    import oracle as orc
    import piper_hip as ph
    cfg = ph.voice_config("high")
    blob = ph.synthetic_blob(cfg, 1234)
    ids = (kd.FIXTURE_IDS * 3)[:14 * 3 - 5]
    dur = [3] * len(ids)
    noise = kd.sym(7 + 3, (cfg.inter, 3 * len(ids)), 1.7)
    ref = orc.synthesize(cfg, blob, ids, dur, noise, 0.667).astype(np.float64)
    for name, x in (("fused", a), ("unfused", b)):
        y = x[-ref.size:].astype(np.float64)
        print(f"  {name}: SNR vs fp32 oracle {10 * np.log10((ref ** 2).sum() / ((y - ref) ** 2).sum()):.1f} dB, max|Δ| {np.abs(y - ref).max():.3e}")
    sys.exit(0 if np.isfinite(a).all() and np.isfinite(b).all() else 1)
