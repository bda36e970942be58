#!/usr/bin/env python3
"""Generate tests/golden/*.npz — run ONLY in the build container (needs torch + transformers, CPU).

Golden vectors for the hot path.  The reference holds no numeric fixtures for this path and its
Swift/Metal sources cannot run here, so the vectors come from an independent implementation of the same ONNX
operator definitions: PyTorch-CPU fp32 functional ops, composed into the Piper VITS modules, and cross-checked at
generation time against the third-party VITS implementation in `transformers.models.vits.modeling_vits`
(attention incl. the relative-position skew, WaveNet / coupling block, HiFi-GAN, encoder layer).

Inputs are NOT stored: tests regenerate them from tests/katdata.py (SplitMix64, version-proof); only expected
outputs (sub-sampled when large) are committed.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as Fn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "piper-swift_amd", "python"))
import katdata as kd  # noqa: E402
import piper_hip as ph  # noqa: E402  (host-only helpers: config presets, blob layout, synthetic blob)

torch.set_num_threads(4)
torch.manual_seed(1234)
MAX_STORE = 20000


def subsample(a):
    a = np.ascontiguousarray(a, np.float32).reshape(-1)
    if a.size <= MAX_STORE:
        return a
    step = -(-a.size // MAX_STORE)
    return a[::step].copy()


from torch_ref import Ref, hf_crosscheck, t  # noqa: E402  (tests/torch_ref.py: the torch model + its HF cross-check)


def main():
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    ops = {}
    with torch.no_grad():
        for i, cs in enumerate(kd.CONV1D_CASES):
            name, cin, cout, k, d, pl, pr, s, g, bias, L, n = cs
            x, w, b = kd.conv1d_inputs(i)
            y = Fn.conv1d(Fn.pad(t(x), (pl, pr)), t(w), None if b is None else t(b), stride=s, dilation=d, groups=g) \
                if (L + pl + pr - d * (k - 1) - 1) >= 0 else torch.zeros(n, cout, 0)
            ops["conv1d." + name] = subsample(y.numpy())
        for i, cs in enumerate(kd.CONVT_CASES):
            name, cin, cout, k, s, pl, pr, op, d, g, bias, L, n = cs
            x, w, b = kd.convt_inputs(i)
            assert pl == pr
            y = Fn.conv_transpose1d(t(x), t(w), None if b is None else t(b), stride=s, padding=pl, output_padding=op, groups=g,
                                    dilation=d)
            ops["convt." + name] = subsample(y.numpy())
        for i, (name, sa, sb) in enumerate(kd.MATMUL_CASES):
            a, b = kd.matmul_inputs(i)
            ops["matmul." + name] = subsample(torch.matmul(t(a), t(b)).numpy())
        for i, (name, s) in enumerate(kd.SOFTMAX_CASES):
            ops["softmax." + name] = subsample(Fn.softmax(t(kd.softmax_input(i)), dim=-1).numpy())
        for i, (name, sa, sb) in enumerate(kd.BINARY_CASES):
            a, b = kd.binary_inputs(i)
            ta, tb = t(a), t(b)
            ops["binary." + name + ".add"] = subsample((ta + tb).numpy())
            ops["binary." + name + ".sub"] = subsample((ta - tb).numpy())
            ops["binary." + name + ".mul"] = subsample((ta * tb).numpy())
            ops["binary." + name + ".div"] = subsample((ta / (tb.abs() + 0.5)).numpy())
        x = t(kd.unary_input())
        ops["unary.relu"] = torch.relu(x).numpy()
        ops["unary.leakyrelu"] = Fn.leaky_relu(x, 0.1).numpy()
        ops["unary.tanh"] = torch.tanh(x).numpy()
        ops["unary.sigmoid"] = torch.sigmoid(x).numpy()
        ops["unary.exp"] = torch.exp(x.clamp(max=80)).numpy()
        ops["unary.erf"] = torch.erf(x).numpy()
        ops["unary.softplus"] = Fn.softplus(x, threshold=1e9).numpy()
    np.savez_compressed(os.path.join(out_dir, "ops_kat.npz"), **ops)

    cfg_m, cfg_h = ph.voice_config("medium"), ph.voice_config("high")
    blob_m, blob_h = ph.synthetic_blob(cfg_m, 1234), ph.synthetic_blob(cfg_h, 1234)
    print("medium params:", blob_m.size, " high params:", blob_h.size)
    hf_crosscheck(cfg_m, blob_m, cfg_h, blob_h)
    ref_m, ref_h = Ref(cfg_m, blob_m), Ref(cfg_h, blob_h)
    mods = {}
    with torch.no_grad():
        sd = kd.case_seed("mod", 0)
        for T in (3, 14, 40):
            q, k, v = (t(kd.sym(sd + j + 10 * T, (1, 192, T))) for j in range(3))
            ek, ev = t(kd.sym(sd + 5, (9, 96), 0.1)), t(kd.sym(sd + 6, (9, 96), 0.1))
            mods[f"rel_attention.T{T}"] = Ref.attention(q, k, v, ek, ev, 2, 4).numpy().reshape(-1)
        x, y = t(kd.sym(sd + 20, (1, 192, 14), 2.0)), t(kd.sym(sd + 21, (1, 192, 14), 2.0))
        g, b = t(1 + kd.sym(sd + 22, (192,), 0.1)), t(kd.sym(sd + 23, (192,), 0.1))
        mods["add_layernorm"] = Ref.layernorm(x + y, g, b).numpy().reshape(-1)
        # WaveNet layer
        C, T, K = 192, 42, 5
        x, sk = t(kd.sym(sd + 30, (1, C, T))), t(kd.sym(sd + 31, (1, C, T)))
        w_in, b_in = t(kd.weight(sd + 32, (2 * C, C, K), C * K)), t(kd.sym(sd + 33, (2 * C,), 0.1))
        w_rs, b_rs = t(kd.weight(sd + 34, (2 * C, C, 1), C)), t(kd.sym(sd + 35, (2 * C,), 0.1))
        xo, so = Ref.wn_layer(x, sk, w_in, b_in, w_rs, b_rs, K, 1, False)
        mods["wavenet_layer.x"], mods["wavenet_layer.skip"] = xo.numpy().reshape(-1), so.numpy().reshape(-1)
        _, so = Ref.wn_layer(x, None, w_in, b_in, w_rs[:C], b_rs[:C], K, 1, True)
        mods["wavenet_layer.last_skip"] = so.numpy().reshape(-1)
        # ResBlocks
        C, T, K = 32, 150, 7
        x = t(kd.sym(sd + 40, (1, C, T)))
        ws = [t(kd.weight(sd + 41 + i, (C, C, K), C * K)) for i in range(2)]
        bs = [t(kd.sym(sd + 45 + i, (C,), 0.1)) for i in range(2)]
        mods["resblock2"] = Ref.resblock(2, x, K, [3, 12], ws, bs).numpy().reshape(-1)
        C, T, K = 64, 60, 3
        x = t(kd.sym(sd + 50, (1, C, T)))
        ws = [t(kd.weight(sd + 51 + i, (C, C, K), C * K)) for i in range(6)]
        bs = [t(kd.sym(sd + 60 + i, (C,), 0.1)) for i in range(6)]
        mods["resblock1"] = Ref.resblock(1, x, K, [1, 3, 5], ws, bs).numpy().reshape(-1)
        # blob-driven modules
        mods["generator_medium.F6"] = ref_m.generator(t(kd.sym(sd + 70, (1, 192, 6)))).numpy().reshape(-1)
        mods["generator_high.F4"] = ref_h.generator(t(kd.sym(sd + 71, (1, 192, 4)))).numpy().reshape(-1)
        mods["flow_reverse.F20"] = ref_m.flow_reverse(t(kd.sym(sd + 72, (1, 192, 20)))).numpy().reshape(-1)
        enc, stats = ref_m.text_encoder(kd.FIXTURE_IDS)
        mods["text_encoder.enc"], mods["text_encoder.stats"] = enc.numpy().reshape(-1), stats.numpy().reshape(-1)
        # whole utterance, factor 1 (14 ids, 3 frames each)
        ids, dur = kd.FIXTURE_IDS, [3] * 14
        noise = kd.sym(sd + 80, (192, 42), 1.7320508)
        r = ref_m.synthesize(ids, dur, noise, 0.667)
        for k_, v_ in r.items():
            mods["synth_f1." + k_] = v_.numpy().reshape(-1)
        # ragged durations incl. zero-length phonemes
        dur2 = [0, 5, 1, 2, 0, 4, 3, 1, 2, 6, 0, 1, 2, 3]
        noise2 = kd.sym(sd + 81, (192, int(np.sum(dur2))), 1.7320508)
        r = ref_m.synthesize(ids, dur2, noise2, 0.667)
        mods["synth_ragged.audio"] = r["audio"].numpy().reshape(-1)
        mods["synth_ragged.z"] = r["z"].numpy().reshape(-1)
    np.savez_compressed(os.path.join(out_dir, "modules.npz"), **{k: np.ascontiguousarray(v, np.float32) for k, v in mods.items()})
    # stochastic duration predictor (torch restatement, cross-checked against transformers' VitsStochasticDurationPredictor)
    dp = {}
    with torch.no_grad():
        sd = kd.case_seed("dp", 0)
        for name, ref, cfg in (("medium", ref_m, cfg_m), ("high", ref_h, cfg_h)):
            for fct in (1, 3):
                ids = kd.FIXTURE_IDS * fct
                nz = kd.sym(sd + fct, (2, len(ids)), 1.7320508)
                for nw in (0.8, 0.0):
                    d, lw = ref.durations(ids, nz, nw, 1.0)
                    dp[f"{name}.f{fct}.nw{nw}.logw"] = lw.astype(np.float32)
                    dp[f"{name}.f{fct}.nw{nw}.dur"] = d.astype(np.float32)
    np.savez_compressed(os.path.join(out_dir, "dp.npz"), **{k: np.ascontiguousarray(v, np.float32) for k, v in dp.items()})
    for f in ("ops_kat.npz", "modules.npz", "dp.npz"):
        print(f, os.path.getsize(os.path.join(out_dir, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
