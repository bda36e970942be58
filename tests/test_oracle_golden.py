"""CPU: pin the oracle (oracle/piper_oracle.c) against the committed golden vectors (tools/gen_golden.py)."""
import numpy as np
import pytest

import katdata as kd
import oracle as orc
from conftest import OP_TOL, WAVE_TOL, assert_close, subsample_like


@pytest.mark.parametrize("idx", range(len(kd.CONV1D_CASES)), ids=[c[0] for c in kd.CONV1D_CASES])
def test_conv1d(idx, golden_ops):
    name, cin, cout, k, d, pl, pr, s, g, bias, L, n = kd.CONV1D_CASES[idx]
    x, w, b = kd.conv1d_inputs(idx)
    y = orc.conv1d(x, w, b, s, d, pl, pr, g)
    assert_close(subsample_like(y), golden_ops["conv1d." + name], OP_TOL, name)


@pytest.mark.parametrize("idx", range(len(kd.CONVT_CASES)), ids=[c[0] for c in kd.CONVT_CASES])
def test_convtranspose1d(idx, golden_ops):
    name, cin, cout, k, s, pl, pr, op, d, g, bias, L, n = kd.CONVT_CASES[idx]
    x, w, b = kd.convt_inputs(idx)
    y = orc.convtranspose1d(x, w, b, s, d, pl, pr, op, g)
    assert_close(subsample_like(y), golden_ops["convt." + name], OP_TOL, name)


@pytest.mark.parametrize("idx", range(len(kd.MATMUL_CASES)), ids=[c[0] for c in kd.MATMUL_CASES])
def test_matmul(idx, golden_ops):
    a, b = kd.matmul_inputs(idx)
    assert_close(subsample_like(orc.matmul(a, b)), golden_ops["matmul." + kd.MATMUL_CASES[idx][0]], OP_TOL)


@pytest.mark.parametrize("idx", range(len(kd.SOFTMAX_CASES)), ids=[c[0] for c in kd.SOFTMAX_CASES])
def test_softmax(idx, golden_ops):
    y = orc.softmax(kd.softmax_input(idx))
    assert_close(subsample_like(y), golden_ops["softmax." + kd.SOFTMAX_CASES[idx][0]], 1e-6)
    assert np.allclose(y.sum(-1), 1.0, atol=1e-5)


@pytest.mark.parametrize("idx", range(len(kd.BINARY_CASES)), ids=[c[0] for c in kd.BINARY_CASES])
def test_binary(idx, golden_ops):
    name = kd.BINARY_CASES[idx][0]
    a, b = kd.binary_inputs(idx)
    for op, nm in ((0, "add"), (1, "sub"), (2, "mul")):
        assert_close(subsample_like(orc.binary(op, a, b)), golden_ops[f"binary.{name}.{nm}"], 1e-6, nm)
    assert_close(subsample_like(orc.binary(3, a, np.abs(b) + np.float32(0.5))), golden_ops[f"binary.{name}.div"], 1e-6, "div")


def test_unary(golden_ops):
    x = kd.unary_input()
    for op, nm, alpha in ((0, "relu", 0), (1, "leakyrelu", 0.1), (2, "tanh", 0), (3, "sigmoid", 0), (9, "erf", 0),
                          (7, "softplus", 0)):
        assert_close(orc.unary(op, x, alpha), golden_ops["unary." + nm], 1e-6, nm)
    assert_close(orc.unary(4, np.minimum(x, 80)), golden_ops["unary.exp"], 1e-5, "exp")


def test_layout_ops_numpy():
    """Pad / Slice / Transpose / Expand / Concat / Split / ReduceMean against numpy (exact)."""
    x = kd.sym(5, (2, 3, 4, 5))
    assert np.array_equal(orc.pad(x, [0, 1, 0, 2, 1, 0, 3, 0]), np.pad(x, ((0, 1), (1, 0), (0, 3), (2, 0))))
    assert np.array_equal(orc.slice_(x, 3, 1, 4), x[..., 1:4])
    assert np.array_equal(orc.slice_(x, 1, 2, -1, -1), x[:, ::-1])
    assert np.array_equal(orc.slice_(x, 2, 0, 4, 2), x[:, :, 0:4:2])
    assert np.array_equal(orc.transpose(x, [0, 2, 1, 3]), x.transpose(0, 2, 1, 3))
    assert np.array_equal(orc.transpose(x, [3, 0, 2, 1]), x.transpose(3, 0, 2, 1))
    e = kd.sym(6, (1, 1, 4, 5))
    assert np.array_equal(orc.expand(e, [2, 3, 4, 5]), np.broadcast_to(e, (2, 3, 4, 5)))
    a, b = kd.sym(7, (2, 3, 6)), kd.sym(8, (2, 5, 6))
    c = orc.concat2_axis1(a, b)
    assert np.array_equal(c, np.concatenate([a, b], 1))
    s0, s1 = orc.split2_axis1(c, 3)
    assert np.array_equal(s0, a) and np.array_equal(s1, b)
    assert np.allclose(orc.reduce_mean_lastdim(x), x.mean(-1), atol=1e-6)


def _mod_inputs():
    return kd.case_seed("mod", 0)


@pytest.mark.parametrize("T", [3, 14, 40])
def test_rel_attention(T, golden_mods):
    sd = _mod_inputs()
    q, k, v = (kd.sym(sd + j + 10 * T, (1, 192, T)) for j in range(3))
    ek, ev = kd.sym(sd + 5, (9, 96), 0.1), kd.sym(sd + 6, (9, 96), 0.1)
    assert_close(orc.rel_attention(q, k, v, ek, ev, 2, 96, T, 4), golden_mods[f"rel_attention.T{T}"], OP_TOL)


def test_add_layernorm(golden_mods):
    sd = _mod_inputs()
    x, y = kd.sym(sd + 20, (1, 192, 14), 2.0), kd.sym(sd + 21, (1, 192, 14), 2.0)
    g, b = 1 + kd.sym(sd + 22, (192,), 0.1), kd.sym(sd + 23, (192,), 0.1)
    assert_close(orc.add_layernorm(x, y, g, b), golden_mods["add_layernorm"], OP_TOL)


def wn_inputs():
    sd = _mod_inputs()
    C, T, K = 192, 42, 5
    return dict(x=kd.sym(sd + 30, (1, C, T)), sk=kd.sym(sd + 31, (1, C, T)), w_in=kd.weight(sd + 32, (2 * C, C, K), C * K),
                b_in=kd.sym(sd + 33, (2 * C,), 0.1), w_rs=kd.weight(sd + 34, (2 * C, C, 1), C), b_rs=kd.sym(sd + 35, (2 * C,), 0.1),
                C=C, T=T, K=K)


def test_wavenet_layer(golden_mods):
    i = wn_inputs()
    xo, so = orc.wavenet_layer(i["x"], i["sk"], i["w_in"], i["b_in"], i["w_rs"], i["b_rs"], i["K"], 1, False)
    assert_close(xo, golden_mods["wavenet_layer.x"], OP_TOL)
    assert_close(so, golden_mods["wavenet_layer.skip"], OP_TOL)
    C = i["C"]
    _, so = orc.wavenet_layer(i["x"], None, i["w_in"], i["b_in"], i["w_rs"][:C], i["b_rs"][:C], i["K"], 1, True)
    assert_close(so, golden_mods["wavenet_layer.last_skip"], OP_TOL)


def rb_inputs(type_):
    sd = _mod_inputs()
    if type_ == 2:
        C, T, K, dils, base, n = 32, 150, 7, [3, 12], 40, 2
        ws = [kd.weight(sd + 41 + i, (C, C, K), C * K) for i in range(n)]
        bs = [kd.sym(sd + 45 + i, (C,), 0.1) for i in range(n)]
    else:
        C, T, K, dils, base, n = 64, 60, 3, [1, 3, 5], 50, 6
        ws = [kd.weight(sd + 51 + i, (C, C, K), C * K) for i in range(n)]
        bs = [kd.sym(sd + 60 + i, (C,), 0.1) for i in range(n)]
    return kd.sym(sd + base, (1, C, T)), K, dils, ws, bs


@pytest.mark.parametrize("type_", [1, 2])
def test_resblock(type_, golden_mods):
    x, K, dils, ws, bs = rb_inputs(type_)
    assert_close(orc.hifigan_resblock(type_, x, K, dils, ws, bs), golden_mods[f"resblock{type_}"], OP_TOL)


def test_generator(golden_mods, voices):
    sd = _mod_inputs()
    cfg, blob = voices["medium"]
    assert_close(orc.generator(cfg, blob, kd.sym(sd + 70, (1, 192, 6))), golden_mods["generator_medium.F6"], WAVE_TOL)
    cfg, blob = voices["high"]
    assert_close(orc.generator(cfg, blob, kd.sym(sd + 71, (1, 192, 4))), golden_mods["generator_high.F4"], WAVE_TOL)


def test_flow_reverse(golden_mods, voices):
    cfg, blob = voices["medium"]
    assert_close(orc.flow_reverse(cfg, blob, kd.sym(_mod_inputs() + 72, (1, 192, 20))), golden_mods["flow_reverse.F20"], OP_TOL)


def test_text_encoder(golden_mods, voices):
    cfg, blob = voices["medium"]
    enc, stats = orc.text_encoder(cfg, blob, kd.FIXTURE_IDS)
    assert_close(enc, golden_mods["text_encoder.enc"], OP_TOL)
    assert_close(stats, golden_mods["text_encoder.stats"], OP_TOL)


def test_synthesize_factor1(golden_mods, voices):
    cfg, blob = voices["medium"]
    noise = kd.sym(_mod_inputs() + 80, (192, 42), 1.7320508)
    audio, taps = orc.synthesize(cfg, blob, kd.FIXTURE_IDS, [3] * 14, noise, 0.667, taps=True)
    for k in ("enc_out", "m_p", "logs_p", "z_p", "z"):
        assert_close(taps[k], golden_mods["synth_f1." + k], OP_TOL, k)
    assert_close(audio, golden_mods["synth_f1.audio"], WAVE_TOL, "audio")
    assert audio.size == 42 * 256


def test_synthesize_ragged(golden_mods, voices):
    cfg, blob = voices["medium"]
    dur = [0, 5, 1, 2, 0, 4, 3, 1, 2, 6, 0, 1, 2, 3]
    noise = kd.sym(_mod_inputs() + 81, (192, sum(dur)), 1.7320508)
    audio, taps = orc.synthesize(cfg, blob, kd.FIXTURE_IDS, dur, noise, 0.667, taps=True)
    assert_close(taps["z"], golden_mods["synth_ragged.z"], OP_TOL)
    assert_close(audio, golden_mods["synth_ragged.audio"], WAVE_TOL)
