"""GPU: the BASELINE.json configurations at their REAL sizes, each against the fp32 oracle (not against this library's own
output): configs[2] factor 64 (encoder / flow taps AND the whole waveform, attention at T = 896), configs[3] the 32 mixed-length utterances
(one by one on slots and bucketed by shape), configs[4] high voice + bf16 generator at factor 8, and the streamed
utterance. The oracle legs are the slow part (≈ 1–15 s each on 16 host threads); the file runs in about a minute."""
import numpy as np
import pytest

import katdata as kd
import oracle as orc
import piper_hip as ph
from conftest import OP_TOL, WAVE_TOL, assert_close
from piper_hip import distributed as phd

pytestmark = pytest.mark.gpu

SD = kd.case_seed("cfg", 0)


def utt(factor, seed, inter=192):
    ids = kd.FIXTURE_IDS * factor
    dur = [3] * len(ids)
    return ids, dur, kd.sym(seed, (inter, 3 * len(ids)), 1.7320508)


def snr_db(x, ref):
    x, ref = np.asarray(x, np.float64), np.asarray(ref, np.float64)
    return 10.0 * np.log10((ref ** 2).sum() / max(((x - ref) ** 2).sum(), 1e-300))


@pytest.fixture(scope="module")
def rt_medium(backend, voices):
    cfg, blob = voices["medium"]
    rt = ph.HipRuntime(backend, cfg, blob)
    yield rt
    rt.close()


def test_config2_factor64_whole_utterance_vs_oracle(rt_medium, voices):
    """BASELINE configs[2]: 896 ids, 2 688 frames, 688 128 samples. ONE oracle run of the whole utterance (≈ 185 GFLOP on the
    host threads): enc_out, m_p / logs_p, z_p, z AND the whole waveform against it — the generator at this size is compared
    with the oracle itself, not with this library's own op-level composition."""
    cfg, blob = voices["medium"]
    ids, dur, noise = utt(64, SD + 1)
    T, F, I, H = len(ids), 3 * len(ids), cfg.inter, cfg.hidden
    rt_medium.prepare(0, ids, dur, noise, 0.667)
    rt_medium.launch(0)
    audio = rt_medium.collect(0)
    assert audio.size == F * cfg.hop == 688128 and np.all(np.isfinite(audio))
    ref, taps = orc.synthesize(cfg, blob, ids, dur, noise, 0.667, taps=True)
    assert_close(rt_medium.tap(0, "enc_out", H * T), taps["enc_out"], OP_TOL, "enc_out @T=896")
    assert_close(rt_medium.tap(0, "m_p", I * T), taps["m_p"], OP_TOL, "m_p @T=896")
    assert_close(rt_medium.tap(0, "logs_p", I * T), taps["logs_p"], OP_TOL, "logs_p @T=896")
    assert_close(rt_medium.tap(0, "z_p", I * F), taps["z_p"], OP_TOL, "z_p @F=2688")
    assert_close(rt_medium.tap(0, "z", I * F), taps["z"], OP_TOL, "z @F=2688")
    err = float(np.abs(audio - ref).max())
    print(f"factor 64 waveform vs oracle: max|Δ| = {err:.3e} over {audio.size} samples")
    assert_close(audio, ref, WAVE_TOL, "factor 64 waveform (688 128 samples) vs oracle")


@pytest.mark.parametrize("T", [896, 301])
def test_rel_attention_long_rows_vs_oracle(T, backend):
    """The fused attention core at the factor-64 length (the op-level suite stops at T = 300)."""
    H, d, w = 2, 96, 4
    q, k, v = (kd.sym(SD + 10 + j, (1, H * d, T)) for j in range(3))
    ek, ev = kd.sym(SD + 15, (2 * w + 1, d), 0.1), kd.sym(SD + 16, (2 * w + 1, d), 0.1)
    ref = orc.rel_attention(q, k, v, ek, ev, H, d, T, w)
    qb, kb, vb, ekb, evb = (backend.uploadFloat32(a) for a in (q, k, v, ek, ev))
    out, shp = backend.relAttentionF32(qb, kb, vb, ekb, evb, 1, H, d, T, w)
    assert shp == [1, H * d, T]
    assert_close(backend.downloadFloat32(out), ref, OP_TOL, f"rel_attention T={T}")
    for b in (qb, kb, vb, ekb, evb, out):
        b.free()


def test_config3_batch32_mixed_lengths_vs_oracle(rt_medium, voices):
    """BASELINE configs[3]: the 32 shuffled factors [1,2,3,4,6,8,12,16]×4 — every utterance is synthesised twice (one by
    one on rotating slots, and bucketed by shape through prepare_batch); factors 12 and 16 and four others are compared
    with the oracle waveform, all 32 must agree between the two routes to fp32 summation order."""
    cfg, blob = voices["medium"]
    factors = phd.batch32_factors()
    assert len(factors) == 32
    utts = [utt(f, SD + 100 + i) for i, f in enumerate(factors)]
    single = []
    for i, (ids, dur, noise) in enumerate(utts):  # one by one, 4 slots in flight
        sl = i % 4
        if i >= 4:
            single.append(rt_medium.collect(sl))
        rt_medium.prepare(sl, ids, dur, noise, 0.667)
        rt_medium.launch(sl)
    # the last four are still in flight, in slot order (28..31 → slots 0..3)
    for sl in range(4):
        single.append(rt_medium.collect(sl))
    assert len(single) == 32 and all(a.size == 14 * f * 3 * cfg.hop for a, f in zip(single, factors))
    by_factor = {}
    for i, f in enumerate(factors):
        by_factor.setdefault(f, []).append(i)
    bucketed = [None] * 32
    for sl, (f, idxs) in enumerate(sorted(by_factor.items())):  # 8 shapes → 8 launches of 4
        rt_medium.prepare_batch(4 + sl, [utts[i] for i in idxs], 0.667)
        rt_medium.launch(4 + sl)
    for sl, (f, idxs) in enumerate(sorted(by_factor.items())):
        a = rt_medium.collect(4 + sl).reshape(len(idxs), -1)
        for k, i in enumerate(idxs):
            bucketed[i] = a[k]
    for i in range(32):
        assert_close(bucketed[i], single[i], 5e-5, f"utterance {i} (factor {factors[i]}): bucketed vs one-by-one")
    # ≤ 2 launches for the whole batch: ragged batches of the 16 longest and the 16 shortest utterances
    order = sorted(range(32), key=lambda i: -factors[i])
    two = [None] * 32
    for sl, idxs in enumerate((order[:16], order[16:])):
        rt_medium.prepare_batch(12 + sl, [utts[i] for i in idxs], 0.667)
        rt_medium.launch(12 + sl)
    for sl, idxs in enumerate((order[:16], order[16:])):
        a = rt_medium.collect(12 + sl)
        off = 0
        for i in idxs:
            n = 14 * factors[i] * 3 * cfg.hop
            two[i] = a[off:off + n]
            off += n
        assert off == a.size
    for i in range(32):
        assert_close(two[i], single[i], 5e-5, f"utterance {i} (factor {factors[i]}): 2-launch ragged batch vs one-by-one")
    checked = set()
    for want in (12, 16, 1, 3, 6, 8):
        i = next(j for j, f in enumerate(factors) if f == want and j not in checked)
        checked.add(i)
        ids, dur, noise = utts[i]
        ref = orc.synthesize(cfg, blob, ids, dur, noise, 0.667)
        assert_close(single[i], ref, WAVE_TOL, f"utterance {i} (factor {want}) one-by-one vs oracle")
        assert_close(bucketed[i], ref, WAVE_TOL, f"utterance {i} (factor {want}) bucketed vs oracle")
        assert_close(two[i], ref, WAVE_TOL, f"utterance {i} (factor {want}) 2-launch ragged batch vs oracle")


def test_config4_high_bf16_factor8_vs_oracle(backend, voices):
    """BASELINE configs[4] at its real size: high voice, factor 8 (112 ids, 336 frames, 86 016 samples), bf16 generator.
    Stated tolerance: waveform SNR ≥ 35 dB against the fp32 ORACLE; the fp32 path of the same voice within WAVE_TOL."""
    cfg, blob = voices["high"]
    ids, dur, noise = utt(8, SD + 200, cfg.inter)
    ref, taps = orc.synthesize(cfg, blob, ids, dur, noise, 0.667, taps=True)
    rt = ph.HipRuntime(backend, cfg, blob)
    try:
        fp32 = rt.synthesize(ids, dur, noise, 0.667)
        assert_close(fp32, ref, WAVE_TOL, "high fp32 factor 8 vs oracle")
        rt.set_precision("bf16")
        rt.prepare(0, ids, dur, noise, 0.667)
        rt.launch(0)
        bf = rt.collect(0)
        assert_close(rt.tap(0, "z", cfg.inter * 336), taps["z"], OP_TOL, "z (fp32 part) under bf16 precision")
        s = snr_db(bf, ref)
        print(f"high bf16 factor 8: SNR vs fp32 oracle = {s:.1f} dB, max|Δ| = {np.abs(bf - ref).max():.3e}")
        assert bf.size == 86016 and s >= 35.0, s
    finally:
        rt.close()


@pytest.mark.parametrize("factor,chunk", [(8, 64), (2, 20)])
def test_streamed_utterance_vs_oracle(factor, chunk, rt_medium, voices):
    """stream_begin / stream_next against the ORACLE waveform (test_gpu_voice compares it with synthesize())."""
    cfg, blob = voices["medium"]
    ids, dur, noise = utt(factor, SD + 300 + factor)
    ref = orc.synthesize(cfg, blob, ids, dur, noise, 0.667)
    chunks = list(rt_medium.synthesize_stream(ids, dur, noise, 0.667, chunkFrames=chunk, slot=12))
    assert_close(np.concatenate(chunks), ref, WAVE_TOL, f"streamed factor {factor} chunk {chunk} vs oracle")
