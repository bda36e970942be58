"""The stochastic duration predictor (Piper `dp`: DDSConv stacks, reverse ConvFlows with the inverse rational-quadratic
spline, ElementwiseAffine; exp · length_scale · ceil). Reference call site: PiperMetalRuntime.synthesize runs the whole graph,
the op arms this restates are GraphExecutor.swift:1946 (Erf), :1957 (Softplus), :1917 (Softmax), :2604 (CumSum), :2493 (GatherND),
:2379 (ScatterND), :1699 (Where), :2470 (NonZero), :699 (GatherElements).

CPU: the C oracle against tests/golden/dp.npz (torch restatement, itself 0.0 from transformers' VitsStochasticDurationPredictor).
GPU: piper_hip_voice_predict_durations and prepare(durations = NULL) against the oracle."""
import os

import numpy as np
import pytest

import katdata as kd
import oracle as orc
import piper_hip as ph
from conftest import OP_TOL, WAVE_TOL, assert_close

SD = kd.case_seed("dp", 0)
LOGW_TOL = 2e-4  # |Δ logw|: the spline's quadratic root amplifies fp32 summation-order noise of the 1×1 convs ≈ 20×


def dp_noise(fct):
    return kd.sym(SD + fct, (2, 14 * fct), 1.7320508)


def ceil_safe(logw, length_scale=1.0, margin=2e-3):
    """Mask of positions whose exp(logw)·length_scale is not within `margin` of an integer (ceil is discontinuous there)."""
    w = np.exp(np.asarray(logw, np.float64)) * length_scale
    return np.abs(w - np.round(w)) > margin


@pytest.mark.parametrize("quality", ["medium", "high"])
@pytest.mark.parametrize("fct", [1, 3])
@pytest.mark.parametrize("nw", [0.8, 0.0])
def test_oracle_duration_predictor_vs_golden(quality, fct, nw, voices):
    cfg, blob = voices[quality]
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "dp.npz"))
    ids = kd.FIXTURE_IDS * fct
    enc, _ = orc.text_encoder(cfg, blob, ids)
    lw = orc.duration_logw(cfg, blob, enc, dp_noise(fct), nw)
    ref = g[f"{quality}.f{fct}.nw{nw}.logw"]
    assert np.abs(lw - ref).max() <= LOGW_TOL, np.abs(lw - ref).max()
    d = orc.durations_from_logw(lw, 1.0)
    ok = ceil_safe(ref)
    assert ok.sum() >= ok.size - 2
    assert np.array_equal(d[ok], g[f"{quality}.f{fct}.nw{nw}.dur"].astype(np.int32)[ok])


def test_oracle_durations_length_scale_and_floor():
    lw = np.array([-30.0, 0.0, np.log(2.0), np.log(2.5), 1.0], np.float32)
    assert orc.durations_from_logw(lw, 1.0).tolist() == [1, 1, 2, 3, 3]  # ceil(e^-30) = 1, ceil(1) = 1
    assert orc.durations_from_logw(lw, 2.0).tolist() == [1, 2, 4, 5, 6]


# ------------------------------------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def rts(backend, voices):
    out = {}
    for q in ("medium", "high"):
        cfg, blob = voices[q]
        out[q] = ph.HipRuntime(backend, cfg, blob)
    yield out
    for r in out.values():
        r.close()


@pytest.mark.gpu
@pytest.mark.parametrize("quality", ["medium", "high"])
@pytest.mark.parametrize("fct,nw,ls", [(1, 0.8, 1.0), (3, 0.8, 1.3), (3, 0.0, 1.0), (8, 0.8, 0.9), (64, 0.8, 1.0)])
def test_predict_durations_vs_oracle(quality, fct, nw, ls, rts, voices):
    cfg, blob = voices[quality]
    ids = kd.FIXTURE_IDS * fct
    nz = kd.sym(SD + fct, (2, len(ids)), 1.7320508)
    enc, _ = orc.text_encoder(cfg, blob, ids)
    lw_ref = orc.duration_logw(cfg, blob, enc, nz, nw)
    d_ref = orc.durations_from_logw(lw_ref, ls)
    (d, lw), = rts[quality].predict_durations([(ids, nz)], noise_w=nw, length_scale=ls)
    err = np.abs(lw - lw_ref).max()
    print(f"{quality} f{fct} nw={nw}: max|Δ logw| = {err:.2e}")
    assert err <= LOGW_TOL, err
    ok = ceil_safe(lw_ref, ls)
    assert ok.sum() >= ok.size * 0.98
    assert np.array_equal(d[ok], d_ref[ok])
    assert np.all(np.abs(d - d_ref) <= 1)


@pytest.mark.gpu
def test_predict_durations_ragged_batch(rts, voices):
    """Five utterances of different lengths in ONE predictor launch equal the one-by-one results (to fp32 summation order: the
    bucket decides the conv tile shapes; padding columns never enter a valid column's sums) and the oracle within LOGW_TOL."""
    cfg, blob = voices["medium"]
    rt = rts["medium"]
    items = []
    for i, n in enumerate((14, 5, 37, 1, 112)):
        ids = (kd.FIXTURE_IDS * 8)[:n]
        items.append((ids, kd.sym(SD + 50 + i, (2, n), 1.7320508)))
    batch = rt.predict_durations(items, noise_w=0.8)
    for (ids, nz), (d, lw) in zip(items, batch):
        (d1, lw1), = rt.predict_durations([(ids, nz)], noise_w=0.8)
        ok = ceil_safe(lw1)
        assert np.array_equal(d[ok], d1[ok])
        assert_close(lw, lw1, 2e-5, f"ragged vs single, {len(ids)} ids")
        enc, _ = orc.text_encoder(cfg, blob, ids)
        assert np.abs(lw - orc.duration_logw(cfg, blob, enc, nz, 0.8)).max() <= LOGW_TOL


@pytest.mark.gpu
def test_predict_durations_device_noise(rts, voices):
    """noise_mode DEVICE: the `dp` tensor is RandomNormalLike(seed) over [1, 2, T] drawn on the device — the oracle gets the
    same tensor from orc_random_normal_like."""
    cfg, blob = voices["medium"]
    ids = kd.FIXTURE_IDS * 3
    nz = orc.random_normal_like(2 * len(ids), 1234).reshape(2, -1)
    enc, _ = orc.text_encoder(cfg, blob, ids)
    lw_ref = orc.duration_logw(cfg, blob, enc, nz, 0.8)
    (d, lw), = rts["medium"].predict_durations([(ids, None)], noise_w=0.8, noise_mode="device", seed=1234)
    assert np.abs(lw - lw_ref).max() <= LOGW_TOL
    (_, lw2), = rts["medium"].predict_durations([(ids, None)], noise_w=0.8, noise_mode="device", seed=99)
    assert np.abs(lw2 - lw).max() > 1e-3  # another seed is another draw


@pytest.mark.gpu
def test_predict_durations_null_noise_is_deterministic_mean(rts, voices):
    """Injected mode without a dp_noise tensor = zeros (the reference's deterministic override with noiseW = 0)."""
    cfg, blob = voices["medium"]
    ids = kd.FIXTURE_IDS
    enc, _ = orc.text_encoder(cfg, blob, ids)
    lw_ref = orc.duration_logw(cfg, blob, enc, None, 0.8)
    (_, lw), = rts["medium"].predict_durations([(ids, None)], noise_w=0.8)
    assert np.abs(lw - lw_ref).max() <= LOGW_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("quality", ["medium", "high"])
def test_synthesize_with_predicted_durations(quality, rts, voices):
    """prepare(durations = NULL): the predictor's frames-per-id feed the expansion on the device; waveform against the oracle
    run with the SAME durations (read back through piper_hip_voice_durations), which are themselves checked above."""
    cfg, blob = voices[quality]
    rt = rts[quality]
    ids = kd.FIXTURE_IDS * 2
    nz = kd.sym(SD + 70, (2, len(ids)), 1.7320508)
    rt.prepare(1, ids, None, None, 0.667, length_scale=1.1, noise_w=0.8, dp_noise=nz, noise_mode="device", seed=4321)
    dur = rt.durations(1)
    assert dur.size == len(ids) and dur.min() >= 0 and dur.sum() >= 1
    (d_pred, _), = rt.predict_durations([(ids, nz)], noise_w=0.8, length_scale=1.1)
    assert np.array_equal(dur, d_pred)
    rt.launch(1)
    audio = rt.collect(1)
    F = int(dur.sum())
    assert audio.size == F * cfg.hop
    noise = orc.random_normal_like(cfg.inter * F, 4321).reshape(cfg.inter, F)
    ref = orc.synthesize(cfg, blob, ids, dur.tolist(), noise, 0.667)
    assert_close(audio, ref, WAVE_TOL, f"{quality}: predicted durations, device noise")


@pytest.mark.gpu
def test_batch_mixes_supplied_and_predicted_durations(rts, voices):
    cfg, blob = voices["medium"]
    rt = rts["medium"]
    ids_a, ids_b = kd.FIXTURE_IDS * 2, kd.FIXTURE_IDS
    dur_a = [2] * len(ids_a)
    na = kd.sym(SD + 80, (cfg.inter, sum(dur_a)), 1.7320508)
    rt.prepare_batch(2, [(ids_a, dur_a, na), (ids_b, None, None)], 0.667)  # item b: predicted durations, zero z noise
    dur = rt.durations(2)
    assert dur[:len(ids_a)].tolist() == dur_a
    db = dur[len(ids_a):]
    (d_pred, _), = rt.predict_durations([(ids_b, None)], noise_w=0.8)
    assert np.array_equal(db, d_pred)
    rt.launch(2)
    audio = rt.collect(2)
    na_s, nb_s = sum(dur_a) * cfg.hop, int(db.sum()) * cfg.hop
    assert audio.size == na_s + nb_s
    assert_close(audio[:na_s], orc.synthesize(cfg, blob, ids_a, dur_a, na, 0.667), WAVE_TOL, "item a")
    zb = np.zeros((cfg.inter, int(db.sum())), np.float32)
    assert_close(audio[na_s:], orc.synthesize(cfg, blob, ids_b, db.tolist(), zb, 0.667), WAVE_TOL, "item b")


@pytest.mark.gpu
@pytest.mark.parametrize("quality", ["medium", "high"])
def test_bounded_prepare_matches_the_two_step_path_and_the_oracle(quality, rts, voices):
    """piper_hip_voice_prepare_batch_bounded: the frame count never visits the host between the predictor and the flow — generate_path
    runs on the device, the plan is the bucket of the caller's bound. Same waveform, durations and length as prepare(durations = NULL)
    (which reads the durations back and picks the plan on the host) — bit for bit when both land in the same bucket; and against the oracle."""
    cfg, blob = voices[quality]
    rt = rts[quality]
    ids = kd.FIXTURE_IDS * 3
    nz = kd.sym(SD + 170, (2, len(ids)), 1.7320508)
    kw = dict(length_scale=1.2, noise_w=0.8, dp_noise=nz, noise_mode="device", seed=77)
    rt.prepare(1, ids, None, None, 0.667, **kw)
    rt.launch(1)
    two_step = rt.collect(1).copy()
    dur = rt.durations(1).copy()
    F = int(dur.sum())
    for bound in (F, F + 37, 4 * F):  # exactly enough, another bucket, a much larger plan
        rt.prepare(3, ids, None, None, 0.667, max_frames=bound, **kw)
        per, cap = rt.prepared_samples(3)
        assert cap >= F * cfg.hop and cap % cfg.hop == 0  # capacity of the bucket until collect
        rt.launch(3)
        audio = rt.collect(3)
        assert audio.size == F * cfg.hop
        assert rt.prepared_samples(3) == ([F * cfg.hop], F * cfg.hop)
        assert np.array_equal(rt.durations(3), dur)
        if bound == F:  # the same bucket as the two-step path: the same kernels in the same order
            assert np.array_equal(audio, two_step), (bound, np.abs(audio - two_step).max())
        else:  # a larger bucket may pick other kernels (tile shapes follow the row length): same values up to summation order
            assert_close(audio, two_step, WAVE_TOL, f"{quality}: bound {bound} vs two-step")
    noise = orc.random_normal_like(cfg.inter * F, 77).reshape(cfg.inter, F)
    ref = orc.synthesize(cfg, blob, ids, dur.tolist(), noise, 0.667)
    assert_close(two_step, ref, WAVE_TOL, f"{quality}: bounded prepare")


@pytest.mark.gpu
def test_bounded_prepare_ragged_batch_zero_noise_and_replay(rts, voices):
    cfg, blob = voices["medium"]
    rt = rts["medium"]
    ids_a, ids_b = kd.FIXTURE_IDS * 4, kd.FIXTURE_IDS
    na, nb = kd.sym(SD + 171, (2, len(ids_a)), 1.7320508), kd.sym(SD + 172, (2, len(ids_b)), 1.7320508)
    (da, _), (db, _) = rt.predict_durations([(ids_a, na), (ids_b, nb)], noise_w=0.8)
    Fa, Fb = int(da.sum()), int(db.sum())
    for rep in range(2):  # the second pass replays both captured graphs on the same plans
        rt.prepare_batch_bounded(4, [(ids_a, na), (ids_b, nb)], max(Fa, Fb) + 5, noise_mode="injected")  # injected + no tensor = zero z noise
        rt.launch(4)
        audio = rt.collect(4)
        assert rt.prepared_samples(4)[0] == [Fa * cfg.hop, Fb * cfg.hop]
        assert np.array_equal(rt.durations(4), np.concatenate([da, db]))
        assert audio.size == (Fa + Fb) * cfg.hop
        assert_close(audio[:Fa * cfg.hop], orc.synthesize(cfg, blob, ids_a, da.tolist(), np.zeros((cfg.inter, Fa), np.float32), 0.667), WAVE_TOL, f"item a, pass {rep}")
        assert_close(audio[Fa * cfg.hop:], orc.synthesize(cfg, blob, ids_b, db.tolist(), np.zeros((cfg.inter, Fb), np.float32), 0.667), WAVE_TOL, f"item b, pass {rep}")
    # a supplied-durations request on the same slot id afterwards (the slot gives the predictor plan back)
    dur = [2] * len(ids_b)
    nzb = kd.sym(SD + 173, (cfg.inter, sum(dur)), 1.7320508)
    rt.prepare(4, ids_b, dur, nzb, 0.667)
    rt.launch(4)
    assert_close(rt.collect(4), orc.synthesize(cfg, blob, ids_b, dur, nzb, 0.667), WAVE_TOL, "supplied durations after a bounded request")


@pytest.mark.gpu
def test_bounded_prepare_reports_a_prediction_over_the_bound(rts, voices):
    cfg, _ = voices["medium"]
    rt = rts["medium"]
    ids = kd.FIXTURE_IDS * 2
    nz = kd.sym(SD + 174, (2, len(ids)), 1.7320508)
    (d, _), = rt.predict_durations([(ids, nz)], noise_w=0.8)
    F = int(d.sum())
    assert F > 8
    rt.prepare(5, ids, None, None, 0.667, dp_noise=nz, noise_mode="device", max_frames=F - 3)
    rt.launch(5)
    with pytest.raises(ph.ExecutionError, match=f"wants {F} frames"):
        rt.collect(5)
    rt.prepare(5, ids, None, None, 0.667, dp_noise=nz, noise_mode="device", max_frames=F)  # the slot is usable again
    rt.launch(5)
    assert rt.collect(5).size == F * cfg.hop
    with pytest.raises(ph.ExecutionError, match="predicts the durations"):
        rt.prepare(5, ids, [1] * len(ids), None, 0.667, max_frames=64)
    with pytest.raises(ph.ExecutionError, match="max_frames"):
        rt.prepare(5, ids, None, None, 0.667, max_frames=0)


@pytest.mark.gpu
def test_voice_without_predictor_refuses(backend, voices):
    cfg, blob = voices["medium"]
    cfg2 = ph.voice_config("medium")
    cfg2.dp_present = 0
    n = ph.blob_floats(cfg2)
    rt = ph.HipRuntime(backend, cfg2, blob[:n])
    try:
        with pytest.raises(ph.UnsupportedOp):
            rt.predict_durations([(kd.FIXTURE_IDS, None)])
        with pytest.raises(ph.UnsupportedOp):
            rt.prepare(0, kd.FIXTURE_IDS, None)
        a = rt.synthesize(kd.FIXTURE_IDS, [3] * 14, None, 0.667)  # supplied durations still work
        assert a.size == 42 * cfg.hop
    finally:
        rt.close()


def test_predictor_geometry_is_checked_when_the_voice_is_described():
    """dp_bins up to 32 and hidden up to 4096 used to pass piper_hip_voice_create and fail only at the first predict, inside
    the plan build. The config check now covers what the predictor's kernels cover (16 bins, 16 … 256 channels) — host-only."""
    cfg = ph.voice_config("medium")
    assert ph.blob_floats(cfg) > 0
    cfg.dp_bins = 20
    with pytest.raises(ph.UnsupportedOp):
        ph.blob_floats(cfg)
    cfg = ph.voice_config("medium")
    cfg.hidden, cfg.n_heads = 384, 4
    with pytest.raises(ph.UnsupportedOp):
        ph.blob_floats(cfg)
    cfg.dp_present = 0  # the same geometry without the predictor is fine (durations are then supplied)
    assert ph.blob_floats(cfg) > 0


@pytest.mark.gpu
def test_noise_without_durations_is_refused(rts, voices):
    """`noise` is [inter, F] and with durations == NULL the caller cannot know F — the buffer would be over-read or
    mis-strided. Refused with InvalidArgument; the documented routes (predict_durations first, or device noise) work."""
    rt = rts["medium"]
    ids = kd.FIXTURE_IDS
    with pytest.raises(ph.InvalidArgument):
        rt.prepare(3, ids, None, kd.sym(5, (192, 42), 1.0), 0.667)
    durs = rt.predict_durations([(ids, None)], noise_mode="device", seed=7)[0][0]
    F = int(np.sum(durs))
    rt.prepare(3, ids, durs, kd.sym(5, (192, F), 1.0), 0.667)
    rt.launch(3)
    assert rt.collect(3).size == F * 256
