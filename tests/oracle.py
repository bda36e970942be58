"""ctypes wrapper over oracle/libpiper_oracle.so — TEST INFRASTRUCTURE (the checker), never the product path."""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "libpiper_oracle.so")
_lib = None

f32p = C.POINTER(C.c_float)
longp = C.POINTER(C.c_long)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle")], stdout=subprocess.DEVNULL)
        _lib = C.CDLL(_SO)
        _lib.orc_conv1d.restype = C.c_long
        _lib.orc_convtranspose1d.restype = C.c_long
        _lib.orc_synthesize.restype = C.c_long
        _lib.orc_voice_blob_floats.restype = C.c_size_t
    return _lib


def _f(a):
    return None if a is None else a.ctypes.data_as(f32p)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _longs(v):
    return (C.c_long * len(v))(*[int(x) for x in v])


def conv1d(x, w, b, stride=1, dil=1, padL=0, padR=0, groups=1):
    x, w = _c(x), _c(w)
    b = None if b is None else _c(b)
    N, Cin, L = x.shape
    Cout, _, K = w.shape
    Lout = (L + padL + padR - dil * (K - 1) - 1)
    Lout = int(Lout / stride) + 1  # truncation toward zero like Swift Int division
    assert Lout >= 0
    y = np.zeros((N, Cout, max(Lout, 0)), np.float32)
    r = lib().orc_conv1d(_f(x), C.c_long(N), C.c_long(Cin), C.c_long(L), _f(w), C.c_long(Cout), C.c_long(K), _f(b),
                         C.c_long(stride), C.c_long(dil), C.c_long(padL), C.c_long(padR), C.c_long(groups), _f(y))
    assert r == Lout, (r, Lout)
    return y


def convtranspose1d(x, w, b, stride=1, dil=1, padL=0, padR=0, outPad=0, groups=1):
    x, w = _c(x), _c(w)
    b = None if b is None else _c(b)
    N, Cin, L = x.shape
    _, cog, K = w.shape
    Lout = (L - 1) * stride - padL - padR + dil * (K - 1) + outPad + 1
    y = np.zeros((N, cog * groups, Lout), np.float32)
    r = lib().orc_convtranspose1d(_f(x), C.c_long(N), C.c_long(Cin), C.c_long(L), _f(w), C.c_long(cog), C.c_long(K), _f(b),
                                  C.c_long(stride), C.c_long(dil), C.c_long(padL), C.c_long(padR), C.c_long(outPad),
                                  C.c_long(groups), _f(y))
    assert r == Lout
    return y


def expand(x, out_shape):
    x = _c(x)
    y = np.zeros(out_shape, np.float32)
    assert lib().orc_expand(_f(x), _longs(x.shape), _longs(out_shape), C.c_int(len(out_shape)), _f(y)) == 0
    return y


def matmul(a, b):
    """MatMul arm: rank-4 lead broadcast materialised with expand, then equal-lead batched matmul."""
    a, b = _c(a), _c(b)
    r = a.ndim
    assert b.ndim == r
    lead = [max(a.shape[i], b.shape[i]) for i in range(r - 2)]
    if list(a.shape[:-2]) != lead:
        a = expand(a, lead + list(a.shape[-2:]))
    if list(b.shape[:-2]) != lead:
        b = expand(b, lead + list(b.shape[-2:]))
    batch = int(np.prod(lead)) if lead else 1
    M, K = a.shape[-2:]
    N = b.shape[-1]
    c = np.zeros(lead + [M, N], np.float32)
    lib().orc_matmul(_f(a), _f(b), _f(c), C.c_long(batch), C.c_long(M), C.c_long(N), C.c_long(K))
    return c


def softmax(x):
    x = _c(x)
    y = np.zeros_like(x)
    cols = x.shape[-1]
    lib().orc_softmax_lastdim(_f(x), _f(y), C.c_long(x.size // cols), C.c_long(cols))
    return y


def unary(op, x, alpha=0.0):
    x = _c(x)
    y = np.zeros_like(x)
    assert lib().orc_unary(C.c_int(op), C.c_float(alpha), _f(x), _f(y), C.c_long(x.size)) == 0
    return y


def binary(op, a, b):
    a, b = _c(a), _c(b)
    oshape = np.broadcast_shapes(a.shape, b.shape)
    out = np.zeros(oshape, np.float32)
    osh = (C.c_long * 4)()
    r = lib().orc_binary_broadcast(C.c_int(op), _f(a), _longs(a.shape), C.c_int(a.ndim), _f(b), _longs(b.shape),
                                   C.c_int(b.ndim), _f(out), osh)
    assert r == len(oshape) and list(osh)[:r] == list(oshape)
    return out


def random_normal_like(n, seed=1234, draws=False):
    out = np.zeros(n, np.float32)
    raw = np.zeros(2 * n, np.uint32)
    lib().orc_random_normal_like(C.c_uint64(seed), C.c_long(n), _f(out), raw.ctypes.data_as(C.POINTER(C.c_uint32)))
    return (out, raw.reshape(-1, 2)) if draws else out


def pad(x, pads, value=0.0):
    x = _c(x)
    r = x.ndim
    oshape = [x.shape[d] + pads[d] + pads[r + d] for d in range(r)]
    out = np.zeros(oshape, np.float32)
    osh = (C.c_long * 4)()
    assert lib().orc_pad_constant(_f(x), _longs(x.shape), C.c_int(r), _longs(pads), C.c_float(value), _f(out), osh) == 0
    return out


def slice_(x, axis, start, end, step=1):
    x = _c(x)
    out = np.zeros(max(x.size, 1), np.float32)
    osh = (C.c_long * 4)()
    assert lib().orc_slice(_f(x), _longs(x.shape), C.c_int(x.ndim), C.c_int(axis), C.c_long(start), C.c_long(end),
                           C.c_long(step), _f(out), osh) == 0
    shp = list(osh)[:x.ndim]
    return out[:int(np.prod(shp))].reshape(shp)


def transpose(x, perm):
    x = _c(x)
    out = np.zeros([x.shape[p] for p in perm], np.float32)
    osh = (C.c_long * 4)()
    pm = (C.c_int * len(perm))(*perm)
    assert lib().orc_transpose(_f(x), _longs(x.shape), C.c_int(x.ndim), pm, _f(out), osh) == 0
    return out


def reduce_mean_lastdim(x):
    x = _c(x)
    cols = x.shape[-1]
    y = np.zeros(x.shape[:-1], np.float32)
    lib().orc_reduce_mean_lastdim(_f(x), _f(y), C.c_long(x.size // cols), C.c_long(cols))
    return y


def concat2_axis1(a, b):
    a, b = _c(a), _c(b)
    N, Ca, L = a.shape
    Cb = b.shape[1]
    out = np.zeros((N, Ca + Cb, L), np.float32)
    lib().orc_concat2_axis1(_f(a), C.c_long(N), C.c_long(Ca), _f(b), C.c_long(Cb), C.c_long(L), _f(out))
    return out


def split2_axis1(x, c0):
    x = _c(x)
    N, Cc, L = x.shape
    o0, o1 = np.zeros((N, c0, L), np.float32), np.zeros((N, Cc - c0, L), np.float32)
    lib().orc_split2_axis1(_f(x), C.c_long(N), C.c_long(Cc), C.c_long(L), C.c_long(c0), _f(o0), _f(o1))
    return o0, o1


def rel_attention(q, k, v, ek, ev, heads, d, T, window):
    q, k, v, ek, ev = map(_c, (q, k, v, ek, ev))
    out = np.zeros((1, heads * d, T), np.float32)
    assert lib().orc_rel_attention(_f(q), _f(k), _f(v), _f(ek), _f(ev), C.c_long(heads), C.c_long(d), C.c_long(T),
                                   C.c_long(window), _f(out)) == 0
    return out


def add_layernorm(x, y, gamma, beta, eps=1e-5):
    x = _c(x)
    y = None if y is None else _c(y)
    _, Cc, T = x.shape
    out = np.zeros_like(x)
    assert lib().orc_add_layernorm(_f(x), _f(y), _f(_c(gamma)), _f(_c(beta)), C.c_long(Cc), C.c_long(T), C.c_float(eps),
                                   _f(out)) == 0
    return out


def wavenet_layer(x, skip_in, w_in, b_in, w_rs, b_rs, K, dil, last):
    x = _c(x)
    _, Cc, T = x.shape
    xo, so = np.zeros_like(x), np.zeros_like(x)
    skip_in = None if skip_in is None else _c(skip_in)
    assert lib().orc_wavenet_layer(_f(x), _f(skip_in), _f(_c(w_in)), _f(_c(b_in)), _f(_c(w_rs)), _f(_c(b_rs)), C.c_long(Cc),
                                   C.c_long(T), C.c_long(K), C.c_long(dil), C.c_int(int(last)), _f(xo), _f(so)) == 0
    return (None if last else xo), so


def hifigan_resblock(type_, x, K, dils, weights, biases, slope=0.1):
    x = _c(x)
    _, Cc, T = x.shape
    ws = [_c(w) for w in weights]
    bs = [_c(b) for b in biases]
    wp = (f32p * len(ws))(*[_f(w) for w in ws])
    bp = (f32p * len(bs))(*[_f(b) for b in bs])
    d = (C.c_int * len(dils))(*dils)
    out = np.zeros_like(x)
    assert lib().orc_hifigan_resblock(C.c_int(type_), _f(x), C.c_long(Cc), C.c_long(T), C.c_long(K), d, C.c_int(len(dils)), wp,
                                      bp, C.c_float(slope), _f(out)) == 0
    return out


def generator(cfg, blob, z):
    z = _c(z)
    F = z.shape[-1]
    audio = np.zeros(F * cfg.hop, np.float32)
    assert lib().orc_generator_forward(C.byref(cfg), _f(blob), _f(z), C.c_long(F), _f(audio)) == 0
    return audio


def flow_reverse(cfg, blob, zp):
    zp = _c(zp)
    F = zp.shape[-1]
    z = np.zeros((cfg.inter, F), np.float32)
    assert lib().orc_flow_reverse_forward(C.byref(cfg), _f(blob), _f(zp), C.c_long(F), _f(z)) == 0
    return z


def text_encoder(cfg, blob, ids):
    ids = np.ascontiguousarray(ids, np.int64)
    T = len(ids)
    enc = np.zeros((cfg.hidden, T), np.float32)
    stats = np.zeros((2 * cfg.inter, T), np.float32)
    assert lib().orc_text_encoder_forward(C.byref(cfg), _f(blob), ids.ctypes.data_as(C.POINTER(C.c_int64)), C.c_long(T),
                                          _f(enc), _f(stats)) == 0
    return enc, stats


def synthesize(cfg, blob, ids, durations, noise, noise_scale, taps=False):
    ids = np.ascontiguousarray(ids, np.int64)
    dur = np.ascontiguousarray(durations, np.int32)
    T, F = len(ids), int(dur.sum())
    I = cfg.inter
    audio = np.zeros(F * cfg.hop, np.float32)
    nz = None if noise is None else _c(noise)
    t = dict(enc_out=np.zeros((cfg.hidden, T), np.float32), m_p=np.zeros((I, T), np.float32),
             logs_p=np.zeros((I, T), np.float32), z_p=np.zeros((I, F), np.float32), z=np.zeros((I, F), np.float32))
    n = lib().orc_synthesize(C.byref(cfg), _f(blob), ids.ctypes.data_as(C.POINTER(C.c_int64)), C.c_long(T),
                             dur.ctypes.data_as(C.POINTER(C.c_int32)), _f(nz), C.c_float(noise_scale), _f(audio),
                             _f(t["enc_out"]), _f(t["m_p"]), _f(t["logs_p"]), _f(t["z_p"]), _f(t["z"]))
    assert n == audio.size, (n, audio.size)
    return (audio, t) if taps else audio


def duration_logw(cfg, blob, enc_out, dp_noise, noise_w):
    enc = _c(enc_out)
    T = enc.shape[-1]
    nz = None if dp_noise is None else _c(dp_noise)
    out = np.zeros(T, np.float32)
    assert lib().orc_duration_logw(C.byref(cfg), _f(blob), _f(enc), C.c_long(T), _f(nz), C.c_float(noise_w), _f(out)) == 0
    return out


def durations_from_logw(logw, length_scale=1.0):
    lw = _c(logw)
    d = np.zeros(lw.size, np.int32)
    lib().orc_durations_from_logw(_f(lw), C.c_long(lw.size), C.c_float(length_scale), d.ctypes.data_as(C.POINTER(C.c_int32)))
    return d
