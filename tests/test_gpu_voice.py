"""GPU: whole-utterance path (piper_hip_voice_*) vs oracle, golden vectors and size-independent properties."""
import ctypes as C
import numpy as np
import pytest

import katdata as kd
import oracle as orc
import piper_hip as ph
from conftest import OP_TOL, WAVE_TOL, assert_close

pytestmark = pytest.mark.gpu

SD = kd.case_seed("mod", 0)


@pytest.fixture(scope="module")
def rt_medium(backend, voices):
    cfg, blob = voices["medium"]
    rt = ph.HipRuntime(backend, cfg, blob)
    yield rt
    rt.close()


def run_with_taps(rt, ids, dur, noise, ns=0.667, slot=0):
    rt.prepare(slot, ids, dur, noise, ns)
    rt.launch(slot)
    audio = rt.collect(slot)
    T, F = len(ids), int(np.sum(dur))
    I, H = rt.cfg.inter, rt.cfg.hidden
    taps = {n: rt.tap(slot, n, sz) for n, sz in (("enc_out", H * T), ("m_p", I * T), ("logs_p", I * T), ("z_p", I * F), ("z", I * F))}
    return audio, taps


def test_factor1_vs_golden_and_oracle(rt_medium, golden_mods, voices):
    cfg, blob = voices["medium"]
    ids, dur = kd.FIXTURE_IDS, [3] * 14
    noise = kd.sym(SD + 80, (192, 42), 1.7320508)
    audio, taps = run_with_taps(rt_medium, ids, dur, noise)
    ref_audio, ref_taps = orc.synthesize(cfg, blob, ids, dur, noise, 0.667, taps=True)
    for k in ("enc_out", "m_p", "logs_p", "z_p", "z"):
        assert_close(taps[k], ref_taps[k], OP_TOL, k + " vs oracle")
        assert_close(taps[k], golden_mods["synth_f1." + k], OP_TOL, k + " vs golden")
    assert audio.size == 42 * 256 == rt_medium.num_samples(ids, dur)
    assert_close(audio, ref_audio, WAVE_TOL, "audio vs oracle")
    assert_close(audio, golden_mods["synth_f1.audio"], WAVE_TOL, "audio vs golden")


def test_ragged_durations(rt_medium, golden_mods):
    dur = [0, 5, 1, 2, 0, 4, 3, 1, 2, 6, 0, 1, 2, 3]
    noise = kd.sym(SD + 81, (192, sum(dur)), 1.7320508)
    audio, taps = run_with_taps(rt_medium, kd.FIXTURE_IDS, dur, noise)
    assert_close(taps["z"], golden_mods["synth_ragged.z"], OP_TOL)
    assert_close(audio, golden_mods["synth_ragged.audio"], WAVE_TOL)


def test_gather_semantics_of_out_of_range_ids(rt_medium, voices):
    """gather_axis0_f32_2d (gather.metal:44-58) / CPUBackend.gather: a negative id wraps once (+n_vocab), what is still out of range
    gathers a zero vector — the embedding kernel and the oracle both follow it."""
    cfg, blob = voices["medium"]
    ids = [1, -1, 20, cfg.n_vocab + 5, -cfg.n_vocab, 2, -cfg.n_vocab - 3]
    dur = [2] * len(ids)
    noise = kd.sym(SD + 90, (cfg.inter, sum(dur)), 1.7320508)
    audio, taps = run_with_taps(rt_medium, ids, dur, noise)
    ref_audio, ref_taps = orc.synthesize(cfg, blob, ids, dur, noise, 0.667, taps=True)
    assert_close(taps["enc_out"], ref_taps["enc_out"], OP_TOL, "enc_out with wrapped / out-of-range ids")
    assert_close(audio, ref_audio, WAVE_TOL)
    # the oracle's side of the rule, stated on its own: −1 ≡ n_vocab − 1 and −n_vocab ≡ 0 give the same encoder output
    a1 = orc.text_encoder(cfg, blob, [1, -1, 20, -cfg.n_vocab, 2])[0]
    a2 = orc.text_encoder(cfg, blob, [1, cfg.n_vocab - 1, 20, 0, 2])[0]
    assert np.array_equal(a1, a2)


def test_collect_into_page_locked_buffer(rt_medium):
    """piper_hip_host_alloc: collect() into a page-locked destination (direct DMA) returns the same samples as into a pageable one."""
    ids, dur = kd.FIXTURE_IDS * 2, [3] * 28
    noise = kd.sym(SD + 91, (192, sum(dur)), 1.7320508)
    rt_medium.prepare(5, ids, dur, noise, 0.667)
    rt_medium.launch(5)
    a = rt_medium.collect(5).copy()
    buf = rt_medium.pinned_empty(a.size + 7)
    buf[:] = -2.0
    rt_medium.launch(5)
    b = rt_medium.collect(5, out=buf)
    assert np.array_equal(a, b) and np.all(buf[a.size:] == -2.0)


@pytest.mark.parametrize("factor", [12, 30])
def test_collect_paths_for_long_waveforms(factor, rt_medium):
    """collect() of a waveform above 1 MB: a pageable destination goes through the slot's page-locked buffer in 1 MB chunks (the host copies chunk k while
    k + 1 is on the wire), a page-locked one is written by the copy kernel (≤ 16 MB) — both must return what the short-waveform path returns sample for sample
    (the waveform itself is compared with the oracle at these sizes in test_gpu_configs.py). factor 12: 0.5 MB (single kernel copy), factor 30: 1.3 MB."""
    ids = kd.FIXTURE_IDS * factor
    dur = [3] * len(ids)
    noise = kd.sym(SD + 92 + factor, (192, sum(dur)), 1.7320508)
    rt_medium.prepare(6, ids, dur, noise, 0.667)
    rt_medium.launch(6)
    a = rt_medium.collect(6).copy()  # pageable np.empty
    assert a.size == sum(dur) * 256 and np.all(np.isfinite(a))
    buf = rt_medium.pinned_empty(a.size + 5)
    buf[:] = -3.0
    rt_medium.launch(6)
    b = rt_medium.collect(6, out=buf)
    assert np.array_equal(a, b) and np.all(buf[a.size:] == -3.0)
    rt_medium.launch(6)
    c = rt_medium.collect(6, out=np.full(a.size + 3, -4.0, np.float32))  # pageable, caller-provided, larger than needed
    assert np.array_equal(a, c)


def test_memory_reserve_slab_serves_the_plans(voices):
    """piper_hip_memory_reserve: one slab taken up front; plan arenas are carved from it, so `reserved` does not move while requests of new shapes arrive."""
    cfg, blob = voices["medium"]
    b = ph.HipBackend(0)
    try:
        b.memory_reserve(3 << 30)
        r0 = b.memory_stats()["reserved"]
        assert r0 >= 3 << 30
        rt = ph.HipRuntime(b, cfg, blob)  # voice_create would reserve 8 GiB: a no-op after the explicit reservation
        try:
            assert b.memory_stats()["reserved"] - r0 < (1 << 30)  # the weights (blob + packed images) come from the slab or beside it, nothing like 8 GiB more
            r1 = b.memory_stats()["reserved"]
            for T in (14, 40, 77):
                a = rt.synthesize(kd.FIXTURE_IDS * (T // 14) + kd.FIXTURE_IDS[:T % 14], [2] * T, None, 0.667)
                assert a.size == 2 * T * cfg.hop
            assert b.memory_stats()["reserved"] == r1, "a plan arena went to the driver although the slab has room"
        finally:
            rt.close()
    finally:
        b.close()


def test_synthesize_api_and_no_noise(rt_medium, voices):
    cfg, blob = voices["medium"]
    ids, dur = [1, 20, 0, 120, 2], [2, 1, 3, 1, 2]
    a = rt_medium.synthesize(ids, dur, None, 0.667)  # noise NULL ⇒ zeros
    assert_close(a, orc.synthesize(cfg, blob, ids, dur, None, 0.667), WAVE_TOL)
    # the same (T,F) again with a different noise scale must not replay stale scalars from the captured graph
    nz = kd.sym(9, (192, 9), 1.0)
    b1 = rt_medium.synthesize(ids, dur, nz, 0.25)
    assert_close(b1, orc.synthesize(cfg, blob, ids, dur, nz, 0.25), WAVE_TOL)
    b2 = rt_medium.synthesize(ids, dur, nz, 1.0)
    assert_close(b2, orc.synthesize(cfg, blob, ids, dur, nz, 1.0), WAVE_TOL)
    assert not np.array_equal(b1, b2)


def test_factor8_vs_oracle(rt_medium, voices):
    """BASELINE configs[1]: 112 ids, 336 frames, 86 016 samples."""
    cfg, blob = voices["medium"]
    ids, dur = kd.FIXTURE_IDS * 8, [3] * 112
    noise = kd.sym(SD + 90, (192, 336), 1.7320508)
    audio, taps = run_with_taps(rt_medium, ids, dur, noise)
    ref_audio, ref_taps = orc.synthesize(cfg, blob, ids, dur, noise, 0.667, taps=True)
    for k in ("enc_out", "z_p", "z"):
        assert_close(taps[k], ref_taps[k], OP_TOL, k)
    assert audio.size == 86016
    assert_close(audio, ref_audio, WAVE_TOL, "audio")


def test_layernorm_with_common_mode_offset_vs_oracle(backend, voices):
    """the LayerNorm folded into the convs (statistics in the producer's epilogue, normalise-on-load in the consumer) used a
    one-pass variance Σy²/C − mean², which loses (mean / sigma)²·6e-8 — harmless on the seeded weights, unpinned for a voice whose residual
    stream carries a common-mode offset. Here the output projection and the second FFN conv of every encoder layer get a bias of +60 on
    every channel (mean / sigma of the LayerNorm input ≈ 50 … 100): the encoder output must still match the oracle's two-pass chain
    (ReduceMean / Sub / Pow / ReduceMean, GraphExecutor.swift:2071-2125) at OP_TOL."""
    cfg, blob0 = voices["medium"]
    blob = blob0.copy()
    for t in ph.blob_layout(cfg):
        if t["name"].endswith(("conv_o.bias", "conv_2.bias")) and t["name"].startswith("enc_p.encoder"):
            blob[t["offset"]:t["offset"] + t["count"]] += 60.0
    rt = ph.HipRuntime(backend, cfg, blob)
    try:
        for factor in (1, 8):
            ids, dur = kd.FIXTURE_IDS * factor, [3] * (14 * factor)
            noise = kd.sym(SD + 95, (192, 42 * factor), 1.7320508)
            audio, taps = run_with_taps(rt, ids, dur, noise)
            ref_audio, ref_taps = orc.synthesize(cfg, blob, ids, dur, noise, 0.667, taps=True)
            assert_close(taps["enc_out"], ref_taps["enc_out"], OP_TOL, f"enc_out with a +60 common-mode offset (factor {factor})")
            assert_close(audio, ref_audio, WAVE_TOL, f"audio with a +60 common-mode offset (factor {factor})")
    finally:
        rt.close()


def test_errors(rt_medium):
    with pytest.raises(ph.ShapeMismatch):
        rt_medium.synthesize([1, 2], [0, 0])  # zero frames
    with pytest.raises(ph.ShapeMismatch):
        rt_medium.synthesize([1] * 5000, [1] * 5000)  # > 4096 ids (PiperCLI.swift:394)
    with pytest.raises(ph.InvalidArgument):
        rt_medium.launch(11)  # slot never prepared


def op_level_generator(b, cfg, blob, z, F):
    """The HiFi-GAN generator executed op by op through the MetalBackend-shaped API (no fused voice path)."""
    lay = {e["name"]: e for e in ph.blob_layout(cfg)}

    def W(n):
        e = lay[n]
        return b.uploadFloat32(blob[e["offset"]:e["offset"] + e["count"]]), e["shape"]

    zd = b.uploadFloat32(z)
    w, ws = W("dec.conv_pre.weight")
    bias, _ = W("dec.conv_pre.bias")
    x, shp = b.conv1dF32(zd, [1, cfg.inter, F], w, ws, bias, padL=3, padR=3)
    ch = cfg.up_initial
    for u in range(cfg.n_ups):
        x = b.leakyReluF32(x, int(np.prod(shp)), 0.1)
        w, ws = W(f"dec.ups.{u}.weight")
        bias, _ = W(f"dec.ups.{u}.bias")
        k, s = cfg.up_kernels[u], cfg.up_rates[u]
        x, shp = b.convTranspose1dF32(x, shp, w, ws, bias, stride=s, padL=(k - s) // 2, padR=(k - s) // 2)
        ch //= 2
        xs = None
        for j in range(cfg.n_rb):
            rb = u * cfg.n_rb + j
            dils = [cfg.rb_dilations[j][d] for d in range(cfg.rb_n_dil)]
            names = []
            for d in range(cfg.rb_n_dil):
                names += [f"dec.resblocks.{rb}.convs1.{d}", f"dec.resblocks.{rb}.convs2.{d}"] if cfg.resblock_type == 1 \
                    else [f"dec.resblocks.{rb}.convs.{d}"]
            wl = [W(n + ".weight")[0] for n in names]
            bl = [W(n + ".bias")[0] for n in names]
            r = b.hifiganResblockF32(cfg.resblock_type, x, 1, ch, shp[2], cfg.rb_kernels[j], dils, wl, bl, 0.1)
            xs = r if xs is None else b.addF32(xs, shp, r, shp)[0]
        x, _ = b.divF32(xs, shp, b.uploadFloat32(np.array([cfg.n_rb], np.float32)), [1])
    x = b.leakyReluF32(x, int(np.prod(shp)), 0.01)
    w, ws = W("dec.conv_post.weight")
    x, shp = b.conv1dF32(x, shp, w, ws, None, padL=3, padR=3)
    x = b.tanhF32(x, int(np.prod(shp)))
    return b.downloadFloat32(x, int(np.prod(shp)))


@pytest.mark.parametrize("factor", [8, 64])
def test_full_size_properties(factor, backend, rt_medium, voices):
    """BASELINE full sizes (factor 64 = 896 ids, 2 688 frames, 688 128 samples) via size-independent properties:
    (1) finite, bounded by tanh; (2) bit-identical on replay and across slots (no races, no stale state);
    (3) the fused generator equals the op-by-op composition through the C-ABI on the same z."""
    cfg, blob = voices["medium"]
    T = 14 * factor
    ids, dur = kd.FIXTURE_IDS * factor, [3] * T
    F = 3 * T
    noise = kd.sym(SD + 100 + factor, (192, F), 1.7320508)
    audio, taps = run_with_taps(rt_medium, ids, dur, noise, slot=0)
    assert audio.size == F * 256
    assert np.all(np.isfinite(audio)) and np.max(np.abs(audio)) <= 1.0
    assert float(np.std(audio)) > 1e-3
    rt_medium.launch(0)
    again = rt_medium.collect(0)
    assert np.array_equal(audio, again), "graph replay is not deterministic"
    other, _ = run_with_taps(rt_medium, ids, dur, noise, slot=3)
    assert np.array_equal(audio, other), "slots disagree"
    ref = op_level_generator(backend, cfg, blob, taps["z"], F)
    assert_close(audio, ref, 2e-4, "fused generator vs op-level composition")


def test_high_voice_vs_oracle(backend, voices):
    cfg, blob = voices["high"]
    rt = ph.HipRuntime(backend, cfg, blob)
    try:
        ids, dur = kd.FIXTURE_IDS, [1] * 14
        noise = kd.sym(SD + 120, (192, 14), 1.7320508)
        audio, taps = run_with_taps(rt, ids, dur, noise)
        ref_audio, ref_taps = orc.synthesize(cfg, blob, ids, dur, noise, 0.667, taps=True)
        assert_close(taps["z"], ref_taps["z"], OP_TOL)
        assert audio.size == 14 * 256
        assert_close(audio, ref_audio, WAVE_TOL)
    finally:
        rt.close()


def test_overlapped_slots(rt_medium, voices):
    """Several prepared utterances of different lengths launched back-to-back overlap on the GPU and stay correct."""
    cfg, blob = voices["medium"]
    utts = []
    for s, f in enumerate((1, 2, 3, 1)):
        ids, dur = kd.FIXTURE_IDS * f, [3] * (14 * f)
        noise = kd.sym(SD + 200 + s, (192, 42 * f), 1.7320508)
        utts.append((ids, dur, noise))
        rt_medium.prepare(4 + s, ids, dur, noise, 0.667)
    for s in range(4):
        rt_medium.launch(4 + s)
    for s, (ids, dur, noise) in enumerate(utts):
        assert_close(rt_medium.collect(4 + s), orc.synthesize(cfg, blob, ids, dur, noise, 0.667), WAVE_TOL, f"slot {4 + s}")


def test_max_phonemes_cap(rt_medium):
    """4096 ids is the reference's --max-phonemes cap (PiperCLI.swift:394): the largest accepted utterance must run
    (attention with a 4096-wide score strip, 1 M-sample decoder rows) and stay finite and deterministic."""
    T = 4096
    ids = (kd.FIXTURE_IDS * 300)[:T]
    dur = [1] * T
    noise = kd.sym(SD + 300, (192, T), 1.0)
    rt_medium.prepare(9, ids, dur, noise, 0.667)
    rt_medium.launch(9)
    a = rt_medium.collect(9)
    assert a.size == T * 256 and np.all(np.isfinite(a)) and np.max(np.abs(a)) <= 1.0
    rt_medium.launch(9)
    assert np.array_equal(a, rt_medium.collect(9))


def test_time_subset_and_profile(rt_medium):
    ids, dur = kd.FIXTURE_IDS * 2, [3] * 28
    rt_medium.prepare(10, ids, dur, None, 0.667)
    rt_medium.launch(10)
    ref = rt_medium.collect(10)
    us, n, fl, by = rt_medium.time_subset(10, "conv_mfma", iters=3)
    assert n > 60 and us > 0 and fl > 1e9 and by > 1e6
    us_all, n_all, fl_all, _ = rt_medium.time_subset(10, "", iters=3)
    assert n_all > n and fl_all >= fl
    st = rt_medium.profile(10, iters=2)
    assert any(s["name"].startswith("(event floor") for s in st) and sum(s["flops"] for s in st) == pytest.approx(fl_all)
    # "a|b|": exactly the named launches, replayed as one graph (how bench.py times a kernel family)
    gates = [s["name"] for s in st if s["name"].endswith(".in_gate")]
    assert len(gates) == 16
    us_g, n_g, fl_g, _ = rt_medium.time_subset(10, "|".join(gates) + "|", iters=3)
    assert n_g == 16 and us_g > 0 and fl_g == pytest.approx(sum(s["flops"] for s in st if s["name"] in gates))
    assert rt_medium.time_subset(10, "no such launch|", iters=1)[1] == 0
    # profiling replays non-idempotent steps; a normal launch afterwards must still be exact
    rt_medium.launch(10)
    assert np.array_equal(ref, rt_medium.collect(10))


def test_batched_utterances_share_one_schedule(rt_medium, voices):
    """N utterances of identical shape (same T, same Σ durations — ids, durations and noise all differ) run as ONE
    schedule with a batch dimension; each waveform must match the oracle's for that utterance."""
    cfg, blob = voices["medium"]
    T, F = 28, 84
    rng = np.random.RandomState(7)
    utts = []
    for b in range(4):
        ids = list(rng.randint(0, 130, size=T))
        dur = [3] * T
        for _ in range(20):  # move frames around, keeping the total
            i, j = rng.randint(0, T, size=2)
            if dur[i] > 0:
                dur[i] -= 1
                dur[j] += 1
        assert sum(dur) == F
        utts.append((ids, dur, kd.sym(SD + 400 + b, (192, F), 1.7320508)))
    utts[2] = (utts[2][0], utts[2][1], None)  # one item without injected noise
    rt_medium.prepare_batch(11, utts, 0.667)
    assert rt_medium.lib.piper_hip_voice_batch_size(rt_medium.voice, 11) == 4
    rt_medium.launch(11)
    audio = rt_medium.collect(11).reshape(4, F * 256)
    for b, (ids, dur, noise) in enumerate(utts):
        assert_close(audio[b], orc.synthesize(cfg, blob, ids, dur, noise, 0.667), WAVE_TOL, f"batch item {b}")
    z = rt_medium.tap(11, "z", 4 * 192 * F).reshape(4, 192, F)
    _, taps = orc.synthesize(cfg, blob, utts[1][0], utts[1][1], utts[1][2], 0.667, taps=True)
    assert_close(z[1], taps["z"], OP_TOL, "z of batch item 1")


def test_ragged_batch_and_bucketed_plans(rt_medium, voices):
    """Items of DIFFERENT (T, F) in one prepare_batch: the plan is the bucket of the longest item, every length-aware kernel
    reads the per-item true lengths from device memory (zero padding / key exclusion at each item's own end — the x_mask /
    y_mask semantics of the reference graph). Each item must equal the oracle run alone; taps come back compacted."""
    cfg, blob = voices["medium"]
    rng = np.random.RandomState(21)
    shapes = [(5, [2, 1, 3, 1, 2]), (37, None), (14, [3] * 14), (50, None), (1, [4])]
    utts = []
    for b, (T, dur) in enumerate(shapes):
        ids = rng.randint(0, 130, size=T).tolist()
        if dur is None:
            dur = rng.randint(0, 5, size=T).tolist()
            dur[0] = 1
        utts.append((ids, dur, kd.sym(SD + 900 + b, (192, int(np.sum(dur))), 1.7320508)))
    rt_medium.prepare_batch(12, utts, 0.667)
    info = rt_medium.plan_info(12)
    assert info["bucket_t"] == 64 and info["bucket_f"] % 16 == 0 and info["bucket_f"] >= max(int(np.sum(u[1])) for u in utts)
    rt_medium.launch(12)
    audio = rt_medium.collect(12)
    z = rt_medium.tap(12, "z", sum(192 * int(np.sum(u[1])) for u in utts))
    enc = rt_medium.tap(12, "enc_out", sum(192 * len(u[0]) for u in utts))
    off = zoff = eoff = 0
    for b, (ids, dur, noise) in enumerate(utts):
        F = int(np.sum(dur))
        ref, taps = orc.synthesize(cfg, blob, ids, dur, noise, 0.667, taps=True)
        assert_close(enc[eoff:eoff + 192 * len(ids)], taps["enc_out"], OP_TOL, f"ragged item {b}: enc_out")
        assert_close(z[zoff:zoff + 192 * F], taps["z"], OP_TOL, f"ragged item {b}: z")
        assert_close(audio[off:off + F * 256], ref, WAVE_TOL, f"ragged item {b}: audio")
        off += F * 256
        zoff += 192 * F
        eoff += 192 * len(ids)
    assert off == audio.size
    # a different utterance that fits the SAME bucket reuses the cached plan (no new graph): "warm" prepare
    before = rt_medium.plan_info(12)["cached_plans"]
    ids2, dur2 = rng.randint(0, 130, size=60).tolist(), [1] * 60
    utts2 = [(ids2, dur2, None)] + utts[1:]
    assert max(int(np.sum(u[1])) for u in utts2) <= info["bucket_f"]
    rt_medium.prepare_batch(12, utts2, 0.667)
    assert rt_medium.plan_info(12)["cached_plans"] == before and rt_medium.plan_info(12)["bucket_t"] == 64
    rt_medium.launch(12)
    a2 = rt_medium.collect(12)
    assert_close(a2[:60 * 256], orc.synthesize(cfg, blob, ids2, dur2, None, 0.667), WAVE_TOL, "warm plan, new utterance")


def test_key_split_attention_with_true_lengths(rt_medium, voices):
    """Rows of more than one key tile run the attention core key-split (parts + merge kernel); the number of parts a block sees
    depends on the item's TRUE length, not on the bucket: a 2-item batch in the T = 304 bucket whose items end inside the third and
    inside the second key tile, each against the oracle run alone (enc_out tap and waveform)."""
    cfg, blob = voices["medium"]
    rng = np.random.RandomState(33)
    utts = []
    for b, T in enumerate((300, 150)):
        ids = rng.randint(0, 130, size=T).tolist()
        dur = [1] * T
        utts.append((ids, dur, kd.sym(SD + 950 + b, (192, T), 1.7320508)))
    rt_medium.prepare_batch(13, utts, 0.667)
    assert rt_medium.plan_info(13)["bucket_t"] == 304
    rt_medium.launch(13)
    audio = rt_medium.collect(13)
    enc = rt_medium.tap(13, "enc_out", sum(192 * len(u[0]) for u in utts))
    off = eoff = 0
    for b, (ids, dur, noise) in enumerate(utts):
        ref, taps = orc.synthesize(cfg, blob, ids, dur, noise, 0.667, taps=True)
        assert_close(enc[eoff:eoff + 192 * len(ids)], taps["enc_out"], OP_TOL, f"key-split item {b}: enc_out")
        assert_close(audio[off:off + len(ids) * 256], ref, WAVE_TOL, f"key-split item {b}: audio")
        off += len(ids) * 256
        eoff += 192 * len(ids)


def test_plan_cache_is_bounded_and_lru(backend, voices):
    """More distinct buckets than the cache holds: idle plans are evicted least-recently-used first, attached ones never."""
    cfg, blob = voices["medium"]
    rt = ph.HipRuntime(backend, cfg, blob)
    try:
        rt.set_plan_cache(12)
        rt.prepare(0, kd.FIXTURE_IDS, [3] * 14, None, 0.667)     # stays attached to slot 0 throughout
        rt.launch(0)
        first = rt.collect(0)
        for T in range(16, 16 * 30, 16):                          # 29 more buckets through slot 1
            rt.prepare(1, [1] * T, [1] * T, None, 0.667)
            if T % 64 == 0:                                       # some are run (eager first launch, then the captured graph), some only prepared
                rt.launch(1)
                a = rt.collect(1)
                rt.launch(1)
                assert np.array_equal(rt.collect(1), a)
        info = rt.plan_info(1)
        assert info["cached_plans"] <= 12 and info["cached_plans"] >= 2
        rt.launch(0)                                              # slot 0's plan survived every eviction
        assert np.array_equal(rt.collect(0), first)
    finally:
        rt.close()


def snr_db(x, ref):
    x, ref = np.asarray(x, np.float64), np.asarray(ref, np.float64)
    return 10.0 * np.log10((ref ** 2).sum() / max(((x - ref) ** 2).sum(), 1e-300))


BF16_MIN_SNR_DB = 35.0  # stated tolerance of the bf16 generator: waveform SNR vs the fp32 result


@pytest.mark.parametrize("quality", ["medium", "high"])
def test_bf16_generator_snr(quality, voices, backend):
    """PIPER_HIP_PRECISION_BF16: generator convs on bf16 operands, fp32 accumulate / residual stream (config 5).
    Checked against the fp32 ORACLE waveform (not against our own fp32 path)."""
    cfg, blob = voices[quality]
    rt = ph.HipRuntime(backend, cfg, blob)
    try:
        ids, dur = kd.FIXTURE_IDS * 2, [3] * 28
        F = sum(dur)
        noise = kd.sym(SD + 500, (cfg.inter, F), 1.7320508)
        ref = orc.synthesize(cfg, blob, ids, dur, noise, 0.667)
        fp32 = rt.synthesize(ids, dur, noise, 0.667)
        rt.set_precision("bf16")
        assert rt.lib.piper_hip_voice_precision(rt.voice) == 1
        rt.prepare(0, ids, dur, noise, 0.667)
        rt.launch(0)
        bf = rt.collect(0)
        # encoder + flow are untouched by the precision switch: z is still the fp32 result
        _, taps = orc.synthesize(cfg, blob, ids, dur, noise, 0.667, taps=True)
        assert_close(rt.tap(0, "z", cfg.inter * F), taps["z"].reshape(-1), OP_TOL, "z under bf16 precision")
        s = snr_db(bf, ref)
        print(f"{quality}: bf16 generator SNR vs fp32 oracle = {s:.1f} dB; max|Δ| = {np.abs(bf - ref).max():.3e}")
        assert s >= BF16_MIN_SNR_DB, s
        assert not np.array_equal(bf, fp32)  # it really is a different arithmetic
        # batch of 2 through the bf16 schedule
        rt.prepare_batch(1, [(ids, dur, noise), (ids[::-1], dur, None)], 0.667)
        rt.launch(1)
        both = rt.collect(1).reshape(2, -1)
        assert snr_db(both[0], ref) >= BF16_MIN_SNR_DB
        assert snr_db(both[1], orc.synthesize(cfg, blob, ids[::-1], dur, None, 0.667)) >= BF16_MIN_SNR_DB
        # and back: the fp32 path is bit-identical to what it produced before the switch
        rt.set_precision("f32")
        assert np.array_equal(rt.synthesize(ids, dur, noise, 0.667), fp32)
    finally:
        rt.close()


def test_voice_loaded_from_onnx_file(backend, voices, tmp_path):
    """`.onnx` (written by tests/onnx_writer.py) → library's own loader → voice → waveform == oracle on the source blob."""
    import onnx_writer as ow
    cfg, blob = voices["medium"]
    lay = [dict(name=t["name"], offset=t["offset"], count=t["count"], shape=list(t["shape"])) for t in ph.blob_layout(cfg)]
    path = tmp_path / "voice.onnx"
    path.write_bytes(ow.piper_voice_onnx(cfg, blob, lay, weight_norm={"dec.ups.1.weight"}))
    cfg2, blob2, _ = ph.load_voice(path)
    rt = ph.HipRuntime(backend, cfg2, blob2)
    try:
        ids, dur = kd.FIXTURE_IDS, [3] * 14
        noise = kd.sym(SD + 600, (cfg.inter, 42), 1.7320508)
        assert_close(rt.synthesize(ids, dur, noise, 0.667), orc.synthesize(cfg, blob, ids, dur, noise, 0.667), WAVE_TOL, "onnx voice")
    finally:
        rt.close()


@pytest.mark.parametrize("quality", ["medium", "high"])
def test_large_batch_takes_the_per_conv_schedules(quality, voices, backend):
    """NB·F above the merge threshold (1536): fp32 runs one launch per ResBlock conv with the MRF mean fused into its
    producer, bf16 additionally uses parallel graph branches for ResBlock1. Both must still match the oracle."""
    cfg, blob = voices[quality]
    rt = ph.HipRuntime(backend, cfg, blob)
    try:
        T, F, NB = 28, 84, 20
        rng = np.random.RandomState(11)
        utts = [(list(rng.randint(0, 130, size=T)), [3] * T, kd.sym(SD + 700 + b, (cfg.inter, F), 1.7320508)) for b in range(NB)]
        refs = {b: orc.synthesize(cfg, blob, utts[b][0], utts[b][1], utts[b][2], 0.667) for b in (0, NB - 1)}
        rt.prepare_batch(2, utts, 0.667)
        rt.launch(2)
        audio = rt.collect(2).reshape(NB, -1)
        for b, ref in refs.items():
            assert_close(audio[b], ref, WAVE_TOL, f"fp32 batch item {b}")
        rt.set_precision("bf16")
        rt.prepare_batch(2, utts, 0.667)
        rt.launch(2)
        audio = rt.collect(2).reshape(NB, -1)
        for b, ref in refs.items():
            assert snr_db(audio[b], ref) >= BF16_MIN_SNR_DB, (b, snr_db(audio[b], ref))
    finally:
        rt.close()


def test_c_host_program_end_to_end(rt_medium, tmp_path):
    """examples/synth_demo.c (plain C over the C-ABI, no Python) writes the same 16-bit samples as the ctypes path."""
    import os
    import subprocess
    import wave
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "piper-swift_amd", "lib")
    exe, wav = tmp_path / "synth_demo", tmp_path / "c_host.wav"
    subprocess.check_call(["gcc", "-std=c99", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "synth_demo.c"),
                           "-L" + lib, "-lpiper_hip", "-Wl,-rpath," + lib, "-o", str(exe)])
    out = subprocess.run([str(exe), str(wav), "2"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    ids, dur = kd.FIXTURE_IDS * 2, [3] * 28
    ref = ph.pcm16(rt_medium.synthesize(ids, dur, None, 0.667))
    with wave.open(str(wav), "rb") as w:
        assert w.getframerate() == 22050 and w.getnframes() == ref.size
        got = np.frombuffer(w.readframes(ref.size), "<i2")
    assert np.array_equal(got, ref)
    # the same host with the durations predicted and the noise drawn on the device (ids are all it supplies)
    wav2 = tmp_path / "c_host_predict.wav"
    out = subprocess.run([str(exe), str(wav2), "2", "-", "predict"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    rt_medium.prepare(3, ids, None, None, 0.667, noise_mode="device", seed=1234, length_scale=1.0, noise_w=0.8)
    rt_medium.launch(3)
    ref2 = ph.pcm16(rt_medium.collect(3))
    with wave.open(str(wav2), "rb") as w:
        assert w.getnframes() == ref2.size
        assert np.array_equal(np.frombuffer(w.readframes(ref2.size), "<i2"), ref2)


def test_cli_scale_bench_and_one_shot(rt_medium, tmp_path):
    """examples/piper_hip_cli.c: the reference CLI's --scale-bench (PiperCLI.swift:381-551: same flags, same JSON keys, ids tiled and truncated at
    --max-phonemes, percentiles by linear interpolation) and a one-shot ids → WAV, both in plain C over the C-ABI."""
    import json
    import os
    import subprocess
    import wave
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "piper-swift_amd", "lib")
    exe = tmp_path / "piper_hip_cli"
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "piper_hip_cli.c"),
                           "-L" + lib, "-lpiper_hip", "-Wl,-rpath," + lib, "-o", str(exe)])
    env = dict(os.environ, PIPER_BENCH_GPU_TIMING="1")
    out = subprocess.run([str(exe), "--scale-bench", "--warmup", "1", "--iters", "4", "--scale-factors", "1,2,8", "--max-phonemes", "20"],
                         capture_output=True, text=True, timeout=180, env=env)
    assert out.returncode == 0, out.stderr
    j = json.loads(out.stdout)
    assert j["backend"] == "piper-hip" and j["mode"] == "scale-bench" and j["warmup"] == 1 and j["iters"] == 4 and j["max_phonemes"] == 20
    assert j["scale_factors"] == [1, 2, 8] and j["base_test_phonemes"] == 14 and j["sample_rate"] == 22050
    assert [r["factor"] for r in j["results"]] == [1, 2, 8]
    assert [r["phoneme_count"] for r in j["results"]] == [14, 20, 20]  # tiled, truncated at --max-phonemes
    for r in j["results"]:
        assert 0 < r["ms_p50"] <= r["ms_p95"] <= r["ms_max"] and r["ms_mean"] > 0
        assert 0 < r["gpu_ms_mean"] < r["ms_mean"] and 0 < r["gpu_busy_fraction_mean"] < 1 and r["max_rss_max"] > 0
        assert abs(r["audio_sec"] - r["phoneme_count"] * 3 * 256 / 22050) < 1e-3  # 3 pinned frames per id
    # one shot: ids → 16-bit WAV, the same samples as the ctypes path (pinned 3 frames per id, device noise with the reference's seed)
    wav = tmp_path / "cli.wav"
    ids = kd.FIXTURE_IDS + kd.FIXTURE_IDS[:5]
    out = subprocess.run([str(exe), "--phoneme-ids", ",".join(str(i) for i in ids), "--output", str(wav)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    rt_medium.prepare(3, ids, [3] * len(ids), None, 0.667, noise_mode="device", seed=1234)
    rt_medium.launch(3)
    ref = ph.pcm16(rt_medium.collect(3))
    with wave.open(str(wav), "rb") as w:
        assert w.getframerate() == 22050 and w.getnframes() == ref.size
        assert np.array_equal(np.frombuffer(w.readframes(ref.size), "<i2"), ref)


def test_bf16_fused_pairs_equal_the_two_launch_path(tmp_path):
    """rb_pair_bf16_kernel against conv_bf16_kernel × 2 (PIPER_HIP_NO_RB_PAIR is read once per process, so: two child processes):
    the rounding points are the same, what differs is fp32 summation order and the bf16 roundings it flips — mutual SNR ≥ 45 dB and
    each side ≥ 40 dB against the fp32 oracle on a bucketed (non-multiple) length."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "probe", "bf16_pair_ab.py")
    a, b = str(tmp_path / "fused.npy"), str(tmp_path / "unfused.npy")
    env = dict(os.environ)
    env.pop("PIPER_HIP_NO_RB_PAIR", None)
    subprocess.check_call([sys.executable, tool, "run", a], env=env, timeout=300)
    subprocess.check_call([sys.executable, tool, "run", b], env=dict(env, PIPER_HIP_NO_RB_PAIR="1", PIPER_HIP_TUNING="1"), timeout=300)
    out = subprocess.run([sys.executable, tool, "cmp", a, b], capture_output=True, text=True, timeout=300)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    import re
    snrs = [float(x) for x in re.findall(r"SNR(?: vs fp32 oracle)? ([0-9.]+) dB", out.stdout)]
    assert len(snrs) == 3 and snrs[0] >= 45.0 and min(snrs[1:]) >= 40.0, snrs


@pytest.mark.parametrize("quality,factor,chunk", [("medium", 8, 64), ("medium", 8, 50), ("medium", 1, 64), ("high", 4, 32)])
def test_streaming_generator_matches_whole_utterance(quality, factor, chunk, voices, backend):
    """stream_begin / stream_next: encoder + flow once, generator window by window with the receptive field as halo.
    Concatenated chunks must equal synthesize() up to fp32 summation order (tile splits depend on the window length)."""
    cfg, blob = voices[quality]
    rt = ph.HipRuntime(backend, cfg, blob)
    try:
        ids = kd.FIXTURE_IDS * factor
        dur = [3] * len(ids)
        F = sum(dur)
        noise = kd.sym(SD + 800 + factor, (cfg.inter, F), 1.7320508)
        whole = rt.synthesize(ids, dur, noise, 0.667)
        chunks = list(rt.synthesize_stream(ids, dur, noise, 0.667, chunkFrames=chunk, slot=3))
        assert len(chunks) == -(-F // chunk)
        assert all(c.size == chunk * cfg.hop for c in chunks[:-1])
        streamed = np.concatenate(chunks)
        assert streamed.size == whole.size
        assert_close(streamed, whole, 2e-5, f"streamed vs whole ({quality}, factor {factor}, chunk {chunk})")
        halo = rt.lib.piper_hip_voice_receptive_field(rt.voice)
        assert 8 <= halo <= 24
        # a second stream on the same slot (cached graphs) gives the same samples
        again = np.concatenate(list(rt.synthesize_stream(ids, dur, noise, 0.667, chunkFrames=chunk, slot=3)))
        assert np.array_equal(again, streamed)
        if quality == "medium" and factor == 8 and chunk == 64:
            rt.set_precision("bf16")
            whole_b = rt.synthesize(ids, dur, noise, 0.667)
            streamed_b = np.concatenate(list(rt.synthesize_stream(ids, dur, noise, 0.667, chunkFrames=chunk, slot=3)))
            assert snr_db(streamed_b, whole_b) >= 40.0, snr_db(streamed_b, whole_b)
    finally:
        rt.close()


def test_streaming_errors(rt_medium):
    ids, dur = kd.FIXTURE_IDS, [3] * 14
    with pytest.raises(ph.ExecutionError):
        list(rt_medium.synthesize_stream(ids, dur, None, 0.667, chunkFrames=0))
    got = C.c_int64()
    assert rt_medium.lib.piper_hip_voice_stream_next(rt_medium.voice, 9, None, 0, C.byref(got)) != 0  # no stream on slot 9


def test_soak_many_shapes_memory_plateaus(backend, voices):
    """A serving process sees many (T, F): schedules are rebuilt per shape, the pool recycles their buffers. Two passes over
    the same 60 random shapes must not grow device memory in the second pass, results stay correct, trim gives memory back."""
    cfg, blob = voices["medium"]
    rt = ph.HipRuntime(backend, cfg, blob)
    try:
        rng = np.random.RandomState(5)
        shapes = []
        for _ in range(60):
            T = int(rng.randint(3, 120))
            dur = rng.randint(0, 6, size=T).astype(int).tolist()
            if sum(dur) == 0:
                dur[0] = 2
            shapes.append((rng.randint(0, 130, size=T).tolist(), dur))
        check = {7, 31, 59}

        def one_pass(verify):
            for i, (ids, dur) in enumerate(shapes):
                audio = rt.synthesize(ids, dur, None, 0.667)
                assert audio.size == sum(dur) * cfg.hop and np.all(np.isfinite(audio))
                if verify and i in check:
                    assert_close(audio, orc.synthesize(cfg, blob, ids, dur, None, 0.667), WAVE_TOL, f"soak item {i}")
        one_pass(True)
        first = backend.memory_stats()
        one_pass(False)
        second = backend.memory_stats()
        assert second["reserved"] <= first["reserved"] * 1.05 + (1 << 20), (first, second)
        backend.memory_trim()
        trimmed = backend.memory_stats()
        assert trimmed["reserved"] <= second["reserved"] and trimmed["reserved"] >= trimmed["live"]
        assert np.all(np.isfinite(rt.synthesize(*shapes[0], None, 0.667)))  # still works after the trim
    finally:
        rt.close()
