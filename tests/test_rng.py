"""RandomNormalLike (elementwise.metal:132-163): the reference's noise source. CPU: the oracle restatement against hand-computed
known answers of the integer stream (the reference holds no vectors for it) and distribution sanity; GPU: kernel vs oracle —
integer draws bit-exact, floats to 1e-6 — and the voice path with device-generated noise vs the oracle fed the same noise."""
import numpy as np
import pytest

import katdata as kd
import oracle as orc


def xorshift32(x):
    x ^= (x << 13) & 0xFFFFFFFF
    x ^= x >> 17
    x ^= (x << 5) & 0xFFFFFFFF
    return x


def test_oracle_rng_integer_stream_known_answers():
    """Independent restatement in Python integers (no C, no numpy wraparound): state0 = seed ^ (gid·747796405 + 2891336453) mod 2^32."""
    z, raw = orc.random_normal_like(5000, 1234, draws=True)
    for gid in (0, 1, 2, 17, 4999):
        st = (1234 ^ ((gid * 747796405 + 2891336453) & 0xFFFFFFFF)) & 0xFFFFFFFF
        u0 = xorshift32(st)
        u1 = xorshift32(u0)
        assert (int(raw[gid, 0]), int(raw[gid, 1])) == (u0, u1)
        f0 = (np.float32(u0) + np.float32(1)) / np.float32(4294967296.0)
        f1 = (np.float32(u1) + np.float32(1)) / np.float32(4294967296.0)
        ref = np.sqrt(np.float32(-2) * np.log(f0)) * np.cos(np.float32(6.28318530718) * f1)
        assert abs(float(z[gid]) - float(ref)) < 2e-6
    assert np.all(np.isfinite(z))
    # seed high bits are ignored (MetalBackend.swift:3404 passes seedLo only)
    assert np.array_equal(orc.random_normal_like(100, 1234 + (7 << 32)), z[:100])
    assert not np.array_equal(orc.random_normal_like(100, 1235), z[:100])


def test_oracle_rng_is_roughly_standard_normal():
    z = orc.random_normal_like(192 * 336, 1234)
    assert abs(float(z.mean())) < 0.02 and abs(float(z.std()) - 1.0) < 0.02
    assert 3.0 < float(np.abs(z).max()) < 7.0


@pytest.mark.gpu
def test_gpu_rng_matches_oracle(backend):
    n = 192 * 336 + 5
    z_ref, raw_ref = orc.random_normal_like(n, 1234, draws=True)
    assert np.array_equal(backend.randomDraws(n, 1234), raw_ref), "integer stream differs"
    buf = backend.randomNormalLike([1, n], 1234)
    z = backend.downloadFloat32(buf)
    buf.free()
    assert float(np.max(np.abs(z - z_ref))) <= 1e-6 * max(1.0, float(np.abs(z_ref).max())) * 4
    # a different seed is a different stream
    b2 = backend.randomNormalLike([n], 99)
    assert not np.array_equal(backend.downloadFloat32(b2), z)
    b2.free()


@pytest.mark.gpu
def test_voice_with_device_generated_noise(backend, voices):
    """noise == NULL + PIPER_HIP_NOISE_DEVICE: the `main` RandomNormalLike tensor [1, inter, F] is generated on the device
    inside the path-expansion kernel (element index c·F + f, seed 1234 like GraphExecutor.swift:2656-2659). Feeding the oracle
    the oracle's own RandomNormalLike tensor must give the same waveform."""
    import piper_hip as ph
    from conftest import OP_TOL, WAVE_TOL, assert_close
    cfg, blob = voices["medium"]
    rt = ph.HipRuntime(backend, cfg, blob)
    try:
        ids, dur = kd.FIXTURE_IDS * 2, [3] * 28
        F = 84
        noise = orc.random_normal_like(cfg.inter * F, 1234).reshape(cfg.inter, F)
        ref, taps = orc.synthesize(cfg, blob, ids, dur, noise, 0.667, taps=True)
        rt.prepare(0, ids, dur, None, 0.667, noise_mode="device", seed=1234)
        rt.launch(0)
        audio = rt.collect(0)
        assert_close(rt.tap(0, "z_p", cfg.inter * F), taps["z_p"], OP_TOL, "z_p with device noise")
        assert_close(audio, ref, WAVE_TOL, "waveform with device noise")
        # injected mode with NULL still means zeros (parity runs), and differs
        zero = rt.synthesize(ids, dur, None, 0.667)
        assert_close(zero, orc.synthesize(cfg, blob, ids, dur, None, 0.667), WAVE_TOL)
        assert not np.array_equal(zero, audio)
        # the same slot replayed with another seed: the captured graph must see the new seed
        rt.prepare(0, ids, dur, None, 0.667, noise_mode="device", seed=7)
        rt.launch(0)
        other = rt.collect(0)
        n7 = orc.random_normal_like(cfg.inter * F, 7).reshape(cfg.inter, F)
        assert_close(other, orc.synthesize(cfg, blob, ids, dur, n7, 0.667), WAVE_TOL, "seed 7")
    finally:
        rt.close()
