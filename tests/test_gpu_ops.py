"""GPU parity tests proper: the HIP path through the C-ABI vs the oracle and the committed golden vectors.

Mirrors how the reference would test MetalBackend per op (it has no such tests: SURVEY.md §4)."""
import numpy as np
import pytest

import katdata as kd
import oracle as orc
import piper_hip as ph
from conftest import OP_TOL, assert_close, subsample_like
from test_oracle_golden import rb_inputs, wn_inputs

SD = 1234 + 600011  # seeds of the bf16 cases (disjoint from katdata.case_seed ranges)

pytestmark = pytest.mark.gpu


def up(b, a):
    return b.uploadFloat32(np.ascontiguousarray(a, np.float32))


def dl(b, buf, shape):
    return b.downloadFloat32(buf, int(np.prod(shape)) if len(shape) else 1).reshape(shape)


@pytest.mark.parametrize("idx", range(len(kd.CONV1D_CASES)), ids=[c[0] for c in kd.CONV1D_CASES])
def test_conv1d(idx, backend, golden_ops):
    name, cin, cout, k, d, pl, pr, s, g, bias, L, n = kd.CONV1D_CASES[idx]
    x, w, b = kd.conv1d_inputs(idx)
    out, shp = backend.conv1dF32(up(backend, x), list(x.shape), up(backend, w), list(w.shape), None if b is None else up(backend, b),
                                 stride=s, dilation=d, padL=pl, padR=pr, groups=g)
    ref = orc.conv1d(x, w, b, s, d, pl, pr, g)
    assert shp == list(ref.shape)
    y = dl(backend, out, shp)
    assert_close(y, ref, OP_TOL, name + " vs oracle")
    assert_close(subsample_like(y), golden_ops["conv1d." + name], OP_TOL, name + " vs golden")


@pytest.mark.parametrize("idx", range(len(kd.CONVT_CASES)), ids=[c[0] for c in kd.CONVT_CASES])
def test_convtranspose1d(idx, backend, golden_ops):
    name, cin, cout, k, s, pl, pr, op, d, g, bias, L, n = kd.CONVT_CASES[idx]
    x, w, b = kd.convt_inputs(idx)
    out, shp = backend.convTranspose1dF32(up(backend, x), list(x.shape), up(backend, w), list(w.shape),
                                          None if b is None else up(backend, b), stride=s, dilation=d, padL=pl, padR=pr,
                                          outputPadding=op, groups=g)
    ref = orc.convtranspose1d(x, w, b, s, d, pl, pr, op, g)
    assert shp == list(ref.shape)
    y = dl(backend, out, shp)
    assert_close(y, ref, OP_TOL, name + " vs oracle")
    assert_close(subsample_like(y), golden_ops["convt." + name], OP_TOL, name + " vs golden")


@pytest.mark.parametrize("idx", range(len(kd.MATMUL_CASES)), ids=[c[0] for c in kd.MATMUL_CASES])
def test_matmul(idx, backend, golden_ops):
    a, b = kd.matmul_inputs(idx)
    out, shp = backend.matmulF32(up(backend, a), list(a.shape), up(backend, b), list(b.shape))
    ref = orc.matmul(a, b)
    assert shp == list(ref.shape)
    y = dl(backend, out, shp)
    assert_close(y, ref, OP_TOL)
    assert_close(subsample_like(y), golden_ops["matmul." + kd.MATMUL_CASES[idx][0]], OP_TOL)


@pytest.mark.parametrize("idx", range(len(kd.SOFTMAX_CASES)), ids=[c[0] for c in kd.SOFTMAX_CASES])
def test_softmax(idx, backend, golden_ops):
    x = kd.softmax_input(idx)
    out, shp = backend.softmaxLastDimF32(up(backend, x), list(x.shape))
    y = dl(backend, out, shp)
    assert_close(y, orc.softmax(x), 1e-6)
    assert_close(subsample_like(y), golden_ops["softmax." + kd.SOFTMAX_CASES[idx][0]], 1e-6)


def test_softmax_long_rows(backend):
    x = kd.sym(17, (3, 5000), 6.0)  # block-per-row kernel (> 2048 columns)
    out, shp = backend.softmaxLastDimF32(up(backend, x), list(x.shape))
    assert_close(dl(backend, out, shp), orc.softmax(x), 1e-6)


def test_unary(backend, golden_ops):
    x = kd.unary_input()
    xd = up(backend, x)
    for op, nm, alpha in ((ph.RELU, "relu", 0.0), (ph.LEAKYRELU, "leakyrelu", 0.1), (ph.TANH, "tanh", 0.0),
                          (ph.SIGMOID, "sigmoid", 0.0), (ph.ERF, "erf", 0.0), (ph.SOFTPLUS, "softplus", 0.0)):
        y = backend.downloadFloat32(backend.unaryF32(op, xd, x.size, alpha))
        assert_close(y, orc.unary(op, x, alpha), 2e-6, nm + " vs oracle")
        assert_close(y, golden_ops["unary." + nm], 2e-6, nm + " vs golden")
    xe = np.minimum(x, 80)
    y = backend.downloadFloat32(backend.unaryF32(ph.EXP, up(backend, xe), x.size))
    assert_close(y, orc.unary(ph.EXP, xe), 1e-5, "exp")
    for op in (ph.NEG, ph.CEIL):
        assert np.array_equal(backend.downloadFloat32(backend.unaryF32(op, xd, x.size)), orc.unary(op, x))
    xs = np.abs(x)
    assert_close(backend.downloadFloat32(backend.unaryF32(ph.SQRT, up(backend, xs), x.size)), orc.unary(ph.SQRT, xs), 1e-6)
    # odd length + misaligned view exercise the scalar tail
    y = backend.downloadFloat32(backend.unaryF32(ph.TANH, xd, 4097))
    assert_close(y, orc.unary(ph.TANH, x[:4097]), 2e-6)


@pytest.mark.parametrize("idx", range(len(kd.BINARY_CASES)), ids=[c[0] for c in kd.BINARY_CASES])
def test_binary(idx, backend, golden_ops):
    name = kd.BINARY_CASES[idx][0]
    a, b = kd.binary_inputs(idx)
    ad, bd = up(backend, a), up(backend, b)
    for op, nm in ((ph.ADD, "add"), (ph.SUB, "sub"), (ph.MUL, "mul")):
        out, shp = backend.binaryBroadcastF32(op, ad, list(a.shape), bd, list(b.shape))
        y = dl(backend, out, shp)
        assert shp == list(np.broadcast_shapes(a.shape, b.shape))
        assert np.array_equal(y, orc.binary(op, a, b)), nm
        assert_close(subsample_like(y), golden_ops[f"binary.{name}.{nm}"], 1e-6)
    b2 = np.abs(b) + np.float32(0.5)
    out, shp = backend.divF32(ad, list(a.shape), up(backend, b2), list(b.shape))
    assert_close(dl(backend, out, shp), orc.binary(ph.DIV, a, b2), 1e-6)
    out, shp = backend.binaryBroadcastF32(ph.POW, up(backend, np.abs(a) + 0.1), list(a.shape), bd, list(b.shape))
    assert_close(dl(backend, out, shp), orc.binary(ph.POW, np.abs(a) + np.float32(0.1), b), 1e-5)


def test_layout_ops(backend):
    x = kd.sym(5, (2, 3, 4, 5))
    xd = up(backend, x)
    out, shp = backend.padConstantF32(xd, list(x.shape), [0, 1, 0, 2, 1, 0, 3, 0])
    assert np.array_equal(dl(backend, out, shp), orc.pad(x, [0, 1, 0, 2, 1, 0, 3, 0]))
    out, shp = backend.sliceF32(xd, list(x.shape), 3, 1, 4)
    assert np.array_equal(dl(backend, out, shp), x[..., 1:4])
    out, shp = backend.sliceF32(xd, list(x.shape), 1, 2, -1, -1)  # VITS Flip
    assert np.array_equal(dl(backend, out, shp), x[:, ::-1])
    out, shp = backend.sliceF32(xd, list(x.shape), 2, 0, 4, 2)
    assert np.array_equal(dl(backend, out, shp), x[:, :, 0:4:2])
    for perm in ([0, 2, 1, 3], [3, 0, 2, 1], [0, 1, 3, 2]):
        out, shp = backend.transposeF32(xd, list(x.shape), perm)
        assert np.array_equal(dl(backend, out, shp), x.transpose(perm))
    e = kd.sym(6, (1, 1, 4, 5))
    out = backend.expandF32(up(backend, e), list(e.shape), [2, 3, 4, 5])
    assert np.array_equal(dl(backend, out, [2, 3, 4, 5]), np.broadcast_to(e, (2, 3, 4, 5)))
    a, b = kd.sym(7, (2, 3, 6)), kd.sym(8, (2, 5, 6))
    out, shp = backend.concat2Axis1F32(up(backend, a), list(a.shape), up(backend, b), list(b.shape))
    c = dl(backend, out, shp)
    assert np.array_equal(c, np.concatenate([a, b], 1))
    (o0, s0), (o1, s1) = backend.split2Axis1F32(out, shp, 3)
    assert np.array_equal(dl(backend, o0, s0), a) and np.array_equal(dl(backend, o1, s1), b)
    out, shp = backend.reduceMeanLastDimF32(xd, list(x.shape))
    assert_close(dl(backend, out, shp), orc.reduce_mean_lastdim(x), 1e-6)


def test_rel_position_skew_as_ops(backend):
    """rel→abs / abs→rel executed the reference's way (Pad, Reshape(alias), Pad, Reshape, Slice) on the GPU ops."""
    h, L = 2, 7
    x = kd.sym(31, (1, h, L, 2 * L - 1))
    o1, s1 = backend.padConstantF32(up(backend, x), list(x.shape), [0, 0, 0, 0, 0, 0, 0, 1])
    o2, s2 = backend.padConstantF32(o1, [1, h, L * 2 * L], [0, 0, 0, 0, 0, L - 1])
    o3, s3 = backend.sliceF32(o2, [1, h, L + 1, 2 * L - 1], 2, 0, L)
    o4, s4 = backend.sliceF32(o3, s3, 3, L - 1, 2 * L - 1)
    got = dl(backend, o4, s4)
    ref = np.zeros((1, h, L, L), np.float32)
    for i in range(L):
        for j in range(L):
            ref[0, :, i, j] = x[0, :, i, j - i + L - 1]
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("T", [1, 3, 14, 40, 112, 300])
def test_rel_attention(T, backend, golden_mods):
    sd = kd.case_seed("mod", 0)
    q, k, v = (kd.sym(sd + j + 10 * T, (1, 192, T)) for j in range(3))
    ek, ev = kd.sym(sd + 5, (9, 96), 0.1), kd.sym(sd + 6, (9, 96), 0.1)
    out, shp = backend.relAttentionF32(up(backend, q), up(backend, k), up(backend, v), up(backend, ek), up(backend, ev), 1, 2, 96,
                                       T, 4)
    y = dl(backend, out, shp)
    assert_close(y, orc.rel_attention(q, k, v, ek, ev, 2, 96, T, 4), OP_TOL, "vs oracle")
    if f"rel_attention.T{T}" in golden_mods:
        assert_close(y, golden_mods[f"rel_attention.T{T}"], OP_TOL, "vs golden")


def test_rel_attention_equals_unfused_ops(backend):
    """Fused attention == the MatMul/Pad/Slice/Softmax composition run through the op-level C-ABI."""
    T, H, d, w = 20, 2, 96, 4
    sd = 4242
    q, k, v = (kd.sym(sd + j, (1, H * d, T)) for j in range(3))
    ek, ev = kd.sym(sd + 5, (2 * w + 1, d), 0.1), kd.sym(sd + 6, (2 * w + 1, d), 0.1)
    b = backend
    fused, _ = b.relAttentionF32(up(b, q), up(b, k), up(b, v), up(b, ek), up(b, ev), 1, H, d, T, w)
    qt, _ = b.transposeF32(up(b, q), [1, H, d, T], [0, 1, 3, 2])
    vt, _ = b.transposeF32(up(b, v), [1, H, d, T], [0, 1, 3, 2])
    qs, _ = b.divF32(qt, [1, H, T, d], up(b, np.array([np.sqrt(np.float32(d))], np.float32)), [1])
    scores, _ = b.matmulF32(qs, [1, H, T, d], up(b, k), [1, H, d, T])
    pad = T - (w + 1)
    ekp, s_ = b.padConstantF32(up(b, ek), [1, 2 * w + 1, d], [0, pad, 0, 0, pad, 0])
    ekT, _ = b.transposeF32(ekp, [1, 1, 2 * T - 1, d], [0, 1, 3, 2])
    rl, _ = b.matmulF32(qs, [1, H, T, d], ekT, [1, 1, d, 2 * T - 1])  # lead-dim broadcast, not materialised
    o1, _ = b.padConstantF32(rl, [1, H, T, 2 * T - 1], [0, 0, 0, 0, 0, 0, 0, 1])
    o2, _ = b.padConstantF32(o1, [1, H, T * 2 * T], [0, 0, 0, 0, 0, T - 1])
    o3, s3 = b.sliceF32(o2, [1, H, T + 1, 2 * T - 1], 2, 0, T)
    loc, _ = b.sliceF32(o3, s3, 3, T - 1, 2 * T - 1)
    sc2, _ = b.addF32(scores, [1, H, T, T], loc, [1, H, T, T])
    p, _ = b.softmaxLastDimF32(sc2, [1, H, T, T])
    o, _ = b.matmulF32(p, [1, H, T, T], vt, [1, H, T, d])
    a1, _ = b.padConstantF32(p, [1, H, T, T], [0, 0, 0, 0, 0, 0, 0, T - 1])
    a2, _ = b.padConstantF32(a1, [1, H, T * (2 * T - 1)], [0, 0, T, 0, 0, 0])
    rw, _ = b.sliceF32(a2, [1, H, T, 2 * T], 3, 1, 2 * T)
    evp, _ = b.padConstantF32(up(b, ev), [1, 2 * w + 1, d], [0, pad, 0, 0, pad, 0])
    o2_, _ = b.matmulF32(rw, [1, H, T, 2 * T - 1], evp, [1, 1, 2 * T - 1, d])
    osum, _ = b.addF32(o, [1, H, T, d], o2_, [1, H, T, d])
    ot, _ = b.transposeF32(osum, [1, H, T, d], [0, 1, 3, 2])
    assert_close(b.downloadFloat32(fused), b.downloadFloat32(ot), 2e-5, "fused vs unfused composition")


@pytest.mark.parametrize("T,N", [(1, 1), (14, 1), (40, 2), (112, 1), (130, 3), (896, 1)])
def test_attention_block_equals_composition(T, N, backend):
    """attention_block_f32 (one launch) vs the ORACLE's rel_attention → conv1d (k = 1) → add_layernorm per batch item."""
    H, d, w = 2, 96, 4
    Cc = H * d
    q, k, v, x = (kd.sym(SD + 2000 + j + 10 * T, (N, Cc, T)) for j in range(4))
    ek, ev = kd.sym(SD + 2005, (2 * w + 1, d), 0.1), kd.sym(SD + 2006, (2 * w + 1, d), 0.1)
    wo, bo = kd.weight(SD + 2007, (Cc, Cc, 1), Cc), kd.sym(SD + 2008, (Cc,), 0.1)
    g, be = 1 + kd.sym(SD + 2009, (Cc,), 0.1), kd.sym(SD + 2010, (Cc,), 0.1)
    out, shp = backend.attentionBlockF32(*(up(backend, a) for a in (q, k, v, ek, ev, wo, bo, x, g, be)), N, H, d, T, w)
    got = dl(backend, out, shp)
    for n in range(N):
        att = orc.rel_attention(q[n:n + 1], k[n:n + 1], v[n:n + 1], ek, ev, H, d, T, w)
        y = orc.conv1d(att, wo, bo)
        ref = orc.add_layernorm(x[n:n + 1], y, g, be)
        assert_close(got[n], ref[0], OP_TOL, f"attention block T={T} item {n}")
    with pytest.raises(ph.UnsupportedOp):  # outside the fused kernel's geometry: the caller composes the ops
        backend.attentionBlockF32(*(up(backend, a) for a in (q[:, :128], k[:, :128], v[:, :128], ek[:, :64], ev[:, :64], wo[:128, :128], bo[:128],
                                                            x[:, :128], g[:128], be[:128])), N, 2, 64, T, w)


def test_add_layernorm(backend, golden_mods):
    sd = kd.case_seed("mod", 0)
    x, y = kd.sym(sd + 20, (1, 192, 14), 2.0), kd.sym(sd + 21, (1, 192, 14), 2.0)
    g, be = 1 + kd.sym(sd + 22, (192,), 0.1), kd.sym(sd + 23, (192,), 0.1)
    out, shp = backend.addLayerNormF32(up(backend, x), up(backend, y), up(backend, g), up(backend, be), 1, 192, 14)
    got = dl(backend, out, shp)
    assert_close(got, orc.add_layernorm(x, y, g, be), OP_TOL)
    assert_close(got, golden_mods["add_layernorm"], OP_TOL)
    # no residual, ragged T, batch 2
    x2 = kd.sym(sd + 24, (2, 192, 45), 3.0)
    out, shp = backend.addLayerNormF32(up(backend, x2), None, up(backend, g), up(backend, be), 2, 192, 45)
    ref = np.concatenate([orc.add_layernorm(x2[i:i + 1], None, g, be) for i in range(2)])
    assert_close(dl(backend, out, shp), ref, OP_TOL)


def test_wavenet_layer(backend, golden_mods):
    i = wn_inputs()
    b = backend
    C, T, K = i["C"], i["T"], i["K"]
    xo, so = b.wavenetLayerF32(up(b, i["x"]), up(b, i["sk"]), up(b, i["w_in"]), up(b, i["b_in"]), up(b, i["w_rs"]), up(b, i["b_rs"]),
                               1, C, T, K, 1, False)
    rx, rs = orc.wavenet_layer(i["x"], i["sk"], i["w_in"], i["b_in"], i["w_rs"], i["b_rs"], K, 1, False)
    assert_close(b.downloadFloat32(xo), rx, OP_TOL, "x vs oracle")
    assert_close(b.downloadFloat32(so), rs, OP_TOL, "skip vs oracle")
    assert_close(b.downloadFloat32(xo), golden_mods["wavenet_layer.x"], OP_TOL)
    assert_close(b.downloadFloat32(so), golden_mods["wavenet_layer.skip"], OP_TOL)
    _, so = b.wavenetLayerF32(up(b, i["x"]), None, up(b, i["w_in"]), up(b, i["b_in"]), up(b, i["w_rs"][:C]), up(b, i["b_rs"][:C]), 1, C,
                              T, K, 1, True)
    assert_close(b.downloadFloat32(so), golden_mods["wavenet_layer.last_skip"], OP_TOL)


@pytest.mark.parametrize("type_", [1, 2])
def test_hifigan_resblock(type_, backend, golden_mods):
    x, K, dils, ws, bs = rb_inputs(type_)
    b = backend
    _, C, T = x.shape
    out = b.hifiganResblockF32(type_, up(b, x), 1, C, T, K, dils, [up(b, w) for w in ws], [up(b, v) for v in bs], 0.1)
    got = b.downloadFloat32(out)
    assert_close(got, orc.hifigan_resblock(type_, x, K, dils, ws, bs), OP_TOL, "vs oracle")
    assert_close(got, golden_mods[f"resblock{type_}"], OP_TOL, "vs golden")


def test_error_behaviour(backend):
    """Contract violations raise the ExecutionError the Swift code throws (MetalBackend.swift:1162-1181, 1233-1250)."""
    b = backend
    x = up(b, np.zeros((1, 4, 8), np.float32))
    w = up(b, np.zeros((6, 3, 3), np.float32))
    with pytest.raises(ph.ShapeMismatch):
        b.conv1dF32(x, [1, 4, 8], w, [6, 3, 3], None)  # weight C_in mismatch
    with pytest.raises(ph.ShapeMismatch):
        b.conv1dF32(x, [1, 4, 8], w, [6, 2, 3], None, groups=3)  # groups do not divide C_in
    with pytest.raises(ph.ShapeMismatch):
        b.conv1dF32(x, [4, 8], w, [6, 4, 3], None)
    with pytest.raises(ph.ShapeMismatch):
        b.matmulF32(x, [1, 4, 8], w, [1, 7, 3])  # inner dim
    with pytest.raises(ph.ShapeMismatch):
        b.matmulF32(x, [4, 8], w, [1, 8, 3])  # rank mismatch
    with pytest.raises(ph.ShapeMismatch):
        b.matmulF32(x, [2, 2, 8], w, [3, 8, 2])  # lead dims not broadcastable
    with pytest.raises(ph.ShapeMismatch):
        b.softmaxLastDimF32(x, [4, 0])
    with pytest.raises(ph.ShapeMismatch):
        b.addF32(x, [1, 4, 8], w, [1, 3, 8])
    with pytest.raises(ph.ShapeMismatch):
        b.convTranspose1dF32(x, [1, 4, 8], w, [5, 3, 3], None)  # weight[0] != C_in
    with pytest.raises(ph.UnsupportedOp):
        b.unaryF32(77, x, 32)
    # empty tensors still return a (≥1 byte) buffer and a shape (MetalBackend.swift:34-36, 1187-1189)
    out, shp = b.conv1dF32(x, [0, 4, 8], up(b, np.zeros((6, 4, 3), np.float32)), [6, 4, 3], None)
    assert shp == [0, 6, 6] and out.ptr


def test_stream_semantics(backend):
    """commandBuffer != nil ⇒ encode only, results visible after flush (MetalBackend.swift:1223-1226, 841-852)."""
    b = backend
    cb = b.makeCommandBuffer()
    x = kd.sym(3, (1, 64, 500))
    w = kd.weight(4, (64, 64, 3), 192)
    xd, wd = up(b, x), up(b, w)
    cur, shp = xd, list(x.shape)
    for _ in range(4):
        cur, shp = b.conv1dF32(cur, shp, wd, list(w.shape), None, padL=1, padR=1, commandBuffer=cb)
        cur = b.unaryF32(ph.TANH, cur, int(np.prod(shp)), commandBuffer=cb)
    b.flush(cb)
    ref = x
    for _ in range(4):
        ref = np.tanh(orc.conv1d(ref, w, None, 1, 1, 1, 1, 1))
    assert_close(dl(b, cur, shp), ref, OP_TOL)


def test_free_is_stream_ordered(backend):
    """piper_hip_free while a NON-blocking op that reads the buffer is still queued on a user stream (the reference drops
    intermediates while the command buffer is still encoding, GraphExecutor.swift:216-225): the block must not be handed to
    an upload on the default stream before that op has run. Round-1 defect: it was recycled at once."""
    b = backend
    cb = b.makeCommandBuffer()
    n_big = 256 << 20                      # 1 GiB per tensor: one NEG pass ≈ 0.5 ms
    big_in = b.allocateBuffer(n_big * 4)
    keep = [big_in]
    for _ in range(8):                     # ≈ 4 ms of queued work on stream cb
        keep.append(b.unaryF32(ph.NEG, keep[-1], n_big, commandBuffer=cb))
    x = kd.sym(11, (1 << 20,))
    y = kd.sym(12, (1 << 20,))
    xd = b.uploadFloat32(x)
    out = b.unaryF32(ph.NEG, xd, x.size, commandBuffer=cb)  # queued behind that work, reads xd
    xd.free()                                               # the host drops xd while the read is still queued
    yd = b.uploadFloat32(y)                                 # same size class: the old pool handed xd's block straight back
    b.flush(cb)
    assert np.array_equal(b.downloadFloat32(out), -x), "the queued op saw its input after the block was recycled"
    assert np.array_equal(b.downloadFloat32(yd), y)
    for d in (yd, out, *keep):
        d.free()


# Long rows take the LDS-tiled bulk kernel (conv_tile_kernel); the KAT table above only reaches the streaming kernel.
BULK_CONV = [  # (Cin, Cout, K, dil, L, lrelu)
    (32, 32, 7, 12, 20000, True), (32, 32, 3, 1, 16384 + 5, True), (64, 64, 5, 6, 9000, True), (64, 64, 11, 5, 9000, True),
    (128, 128, 7, 3, 6000, False), (256, 256, 3, 1, 4500, True), (32, 64, 1, 1, 20000, False), (48, 40, 5, 2, 20001, True),
]


@pytest.mark.parametrize("case", BULK_CONV, ids=[f"c{c[0]}x{c[1]}_k{c[2]}_d{c[3]}_L{c[4]}" for c in BULK_CONV])
def test_conv1d_bulk(case, backend):
    cin, cout, k, d, L, act = case
    sd = 9000 + cin + 7 * k + d
    x = kd.sym(sd, (1, cin, L))
    w = kd.weight(sd + 1, (cout, cin, k), cin * k)
    b = kd.sym(sd + 2, (cout,), 0.1)
    pad = (k * d - d) // 2
    bk = backend
    if act:  # through the fused ResBlock entry point (LeakyReLU prologue + residual epilogue) when square
        if cin == cout:
            out = bk.hifiganResblockF32(2, up(bk, x), 1, cin, L, k, [d], [up(bk, w)], [up(bk, b)], 0.1)
            ref = orc.hifigan_resblock(2, x, k, [d], [w], [b], 0.1)
            assert_close(bk.downloadFloat32(out), ref, OP_TOL)
            return
        xa = bk.leakyReluF32(up(bk, x), x.size, 0.1)
        x = orc.unary(ph.LEAKYRELU, x, 0.1)
    else:
        xa = up(bk, x)
    out, shp = bk.conv1dF32(xa, list(x.shape), up(bk, w), list(w.shape), up(bk, b), dilation=d, padL=pad, padR=pad)
    assert_close(dl(bk, out, shp), orc.conv1d(x, w, b, 1, d, pad, pad, 1), OP_TOL)


@pytest.mark.parametrize("case", [(64, 32, 8, 4, 2, 6000), (128, 64, 16, 8, 4, 1500), (64, 32, 4, 2, 1, 9000)],
                         ids=["k8s4", "k16s8", "k4s2"])
def test_convtranspose1d_bulk(case, backend):
    cin, cout, k, s, pad, L = case
    x = kd.sym(7000 + k, (1, cin, L))
    w = kd.weight(7001 + k, (cin, cout, k), cin * k // s)
    b = kd.sym(7002 + k, (cout,), 0.1)
    out, shp = backend.convTranspose1dF32(up(backend, x), list(x.shape), up(backend, w), list(w.shape), up(backend, b), stride=s,
                                          padL=pad, padR=pad)
    assert_close(dl(backend, out, shp), orc.convtranspose1d(x, w, b, s, 1, pad, pad, 0, 1), OP_TOL)


# ---- bf16-operand variants. Checker: the fp32 oracle on operands rounded to bf16 on the
# host — products of two bf16 values are exact in fp32, so only the accumulation order differs (OP_TOL).
@pytest.mark.parametrize("idx", range(len(kd.CONV_BF16_CASES)))
def test_conv1d_bf16(idx, backend):
    Cin, Cout, K, d, pl, pr, L, N, has_b = kd.CONV_BF16_CASES[idx]
    x = kd.sym(SD + 900 + idx, (N, Cin, L))
    w = kd.weight(SD + 950 + idx, (Cout, Cin, K), Cin * K)
    b = kd.sym(SD + 990 + idx, (Cout,), 0.1) if has_b else None
    out, shp = backend.conv1dBF16(up(backend, x), list(x.shape), up(backend, w), list(w.shape),
                                  None if b is None else up(backend, b), dilation=d, padL=pl, padR=pr)
    ref = orc.conv1d(kd.bf16_round(x), kd.bf16_round(w), b, 1, d, pl, pr)
    assert shp == list(ref.shape)
    assert_close(dl(backend, out, shp), ref, OP_TOL, f"conv1d_bf16 case {idx}")
    # and it IS a bf16 computation: the unrounded fp32 result differs by about 2^-9 relative
    full = orc.conv1d(x, w, b, 1, d, pl, pr)
    err = np.abs(ref - full).max() / max(1.0, np.abs(full).max())
    assert err < 2e-2


@pytest.mark.parametrize("idx", range(len(kd.CONVT_BF16_CASES)))
def test_convtranspose1d_bf16(idx, backend):
    Cin, Cout, K, s, L, N = kd.CONVT_BF16_CASES[idx]
    pad = (K - s) // 2
    x = kd.sym(SD + 1000 + idx, (N, Cin, L))
    w = kd.weight(SD + 1050 + idx, (Cin, Cout, K), Cin * K // s)
    b = kd.sym(SD + 1090 + idx, (Cout,), 0.1)
    out, shp = backend.convTranspose1dBF16(up(backend, x), list(x.shape), up(backend, w), list(w.shape), up(backend, b),
                                           stride=s, padL=pad, padR=pad)
    ref = orc.convtranspose1d(kd.bf16_round(x), kd.bf16_round(w), b, s, 1, pad, pad)
    assert shp == list(ref.shape)
    assert_close(dl(backend, out, shp), ref, OP_TOL, f"convT_bf16 case {idx}")


def test_bf16_variants_refuse_uncovered_geometry(backend):
    x = up(backend, np.zeros((1, 32, 16), np.float32))
    w = up(backend, np.zeros((32, 32, 3), np.float32))
    with pytest.raises(ph.UnsupportedOp):
        backend.conv1dBF16(x, [1, 32, 16], w, [32, 32, 3], None, stride=2)
    with pytest.raises(ph.UnsupportedOp):
        backend.conv1dBF16(up(backend, np.zeros((1, 20, 16), np.float32)), [1, 20, 16],
                           up(backend, np.zeros((8, 20, 3), np.float32)), [8, 20, 3], None)
    with pytest.raises(ph.UnsupportedOp):
        backend.convTranspose1dBF16(x, [1, 32, 16], w, [32, 32, 3], None, stride=2)  # K % stride != 0
    with pytest.raises(ph.ShapeMismatch):
        backend.conv1dBF16(x, [1, 32, 16], w, [32, 16, 3], None)


# ---- long rows: conv1dF32 / convTranspose1dF32 route to the fp32 window kernel (conv_win.hip)
@pytest.mark.parametrize("idx", range(len(kd.CONV_WIN_CASES)))
def test_conv1d_long_rows_window_kernel(idx, backend):
    Cin, Cout, K, d, pl, pr, L, N, has_b = kd.CONV_WIN_CASES[idx]
    x = kd.sym(SD + 1200 + idx, (N, Cin, L))
    w = kd.weight(SD + 1250 + idx, (Cout, Cin, K), Cin * K)
    b = kd.sym(SD + 1290 + idx, (Cout,), 0.1) if has_b else None
    out, shp = backend.conv1dF32(up(backend, x), list(x.shape), up(backend, w), list(w.shape),
                                 None if b is None else up(backend, b), dilation=d, padL=pl, padR=pr)
    ref = orc.conv1d(x, w, b, 1, d, pl, pr)
    assert shp == list(ref.shape)
    assert_close(dl(backend, out, shp), ref, OP_TOL, f"conv1d window case {idx}")


@pytest.mark.parametrize("idx", range(len(kd.CONVT_WIN_CASES)))
def test_convtranspose1d_long_rows_window_kernel(idx, backend):
    Cin, Cout, K, s, L, N = kd.CONVT_WIN_CASES[idx]
    pad = (K - s) // 2
    x = kd.sym(SD + 1300 + idx, (N, Cin, L))
    w = kd.weight(SD + 1350 + idx, (Cin, Cout, K), Cin * K // s)
    b = kd.sym(SD + 1390 + idx, (Cout,), 0.1)
    out, shp = backend.convTranspose1dF32(up(backend, x), list(x.shape), up(backend, w), list(w.shape), up(backend, b),
                                          stride=s, padL=pad, padR=pad)
    ref = orc.convtranspose1d(x, w, b, s, 1, pad, pad)
    assert shp == list(ref.shape)
    assert_close(dl(backend, out, shp), ref, OP_TOL, f"convT window case {idx}")


def test_convtranspose1d_chunk_pipelined_kernel_on_request():
    """conv_pipe_kernel's ConvTranspose path is off by default (the window kernel is faster since round 2) and selected by
    PIPER_HIP_PIPE_CT_MIN_GFLOP; the switch is read once per process, so the cases run in a child process with it set."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "probe", "convt_pipe_check.py")
    env = dict(os.environ, PIPER_HIP_PIPE_CT_MIN_GFLOP="0", PIPER_HIP_TUNING="1")  # switches are honoured only with PIPER_HIP_TUNING=1
    env.pop("PIPER_HIP_NO_PIPE", None)
    out = subprocess.run([sys.executable, tool, "0", "4", "5"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    rows = re.findall(r"case (\d+): max_abs_err ([0-9.e+-]+) ref_max ([0-9.e+-]+)", out.stdout)
    assert [int(r[0]) for r in rows] == [0, 4, 5], out.stdout
    for _, err, ref_max in rows:
        assert float(err) <= OP_TOL * max(1.0, float(ref_max)), out.stdout


def test_comm_single_rank_world(backend):
    """piper_hip_comm_* over rccl.h with a world of one (all a one-GPU box can hold): the communicator reports its own size,
    the in-place broadcast leaves the root's data untouched, MAX of one value is that value, barrier returns."""
    uid = ph.comm_unique_id()
    assert len(uid) == ph.COMM_ID_BYTES and any(uid)
    c = ph.Comm(backend, uid, 0, 1)
    try:
        assert (c.rank, c.world) == (0, 1)
        x = kd.sym(kd.case_seed("cfg", 77), (1 << 20,))
        buf = backend.uploadFloat32(x)
        c.broadcast_f32(buf, x.size, root=0)
        assert np.array_equal(backend.downloadFloat32(buf), x)
        assert c.max(3.25) == 3.25
        c.barrier()
        with pytest.raises(ph.InvalidArgument):
            c.broadcast_f32(buf, x.size, root=1)
        buf.free()
    finally:
        c.close()
    with pytest.raises(ph.InvalidArgument):
        ph.Comm(backend, uid, 2, 2)


# (type, C, T, K, dilations, batch): the fused pair kernel's geometry (C ∈ {32, 64}, T % 4 == 0) — T below / at / across the
# 224-column block edge, odd pair counts (a pair + a single conv), every Piper kernel size and dilation set
RB_PAIR_CASES = [
    (2, 32, 224, 3, [1, 3], 1), (2, 32, 1000, 5, [2, 6], 1), (2, 32, 452, 7, [3, 12], 2), (2, 32, 8, 7, [1, 3], 1),
    (2, 64, 228, 3, [1, 3], 1), (2, 64, 700, 7, [1, 3], 2), (2, 64, 448, 5, [1, 2, 3], 1),
    (1, 64, 520, 11, [1, 3, 5], 1), (1, 64, 224, 3, [1, 3, 5], 2), (1, 32, 900, 7, [1, 3, 5], 1),
]


@pytest.mark.parametrize("case", RB_PAIR_CASES, ids=lambda c: f"rb{c[0]}_C{c[1]}_T{c[2]}_K{c[3]}_d{'-'.join(map(str, c[4]))}_n{c[5]}")
def test_hifigan_resblock_fused_pairs(case, backend):
    """rb_pair_kernel (two chained convs, intermediate in LDS) against the oracle's conv-by-conv ResBlock."""
    type_, Cc, T, K, dils, N = case
    sd = kd.case_seed("cfg", 500 + Cc + T + K)
    nconv = len(dils) * (2 if type_ == 1 else 1)
    x = kd.sym(sd, (N, Cc, T))
    ws = [kd.weight(sd + 1 + i, (Cc, Cc, K), Cc * K) for i in range(nconv)]
    bs = [kd.sym(sd + 40 + i, (Cc,), 0.1) for i in range(nconv)]
    b = backend
    out = b.hifiganResblockF32(type_, up(b, x), N, Cc, T, K, dils, [up(b, w) for w in ws], [up(b, v) for v in bs], 0.1)
    got = b.downloadFloat32(out).reshape(N, Cc, T)
    for n in range(N):
        assert_close(got[n], orc.hifigan_resblock(type_, x[n:n + 1], K, dils, ws, bs)[0], OP_TOL, f"item {n} vs oracle")


@pytest.mark.gpu
def test_tuning_switches_need_an_explicit_opt_in_and_are_reported():
    """≈ 45 PIPER_HIP_* A/B switches are read by the dispatchers. An inherited environment must not silently change which
    kernel runs: a switch is honoured only with PIPER_HIP_TUNING=1, and piper_hip_config_string() says which ones were."""
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np, piper_hip as ph, katdata as kd\n"
            "b = ph.HipBackend(0)\n"
            "x = b.uploadFloat32(kd.sym(1, (1, 192, 112))); w = b.uploadFloat32(kd.sym(2, (256, 192, 7), 0.05))\n"
            "b.conv1dF32(x, [1, 192, 112], w, [256, 192, 7], None, 1, 1, 3, 3, 1)\n"
            "print('CFG[' + ph.config_string() + ']')\n") % (os.path.join(ROOT, "piper-swift_amd", "python"), os.path.join(ROOT, "tests"))
    base = {k: v for k, v in os.environ.items() if not k.startswith("PIPER_HIP_")}
    out = subprocess.run([sys.executable, "-c", code], env=dict(base, PIPER_HIP_NO_SHORT="1"), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "PIPER_HIP_NO_SHORT=1" not in out.stdout and "ignored" in out.stdout, out.stdout
    out = subprocess.run([sys.executable, "-c", code], env=dict(base, PIPER_HIP_NO_SHORT="1", PIPER_HIP_TUNING="1"), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "PIPER_HIP_NO_SHORT=1" in out.stdout, out.stdout
    out = subprocess.run([sys.executable, "-c", code], env=base, capture_output=True, text=True, timeout=300)
    assert "CFG[]" in out.stdout, out.stdout
