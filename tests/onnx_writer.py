"""Minimal ONNX (protobuf wire format) WRITER for tests: builds Piper-shaped `.onnx` files from a voice blob.

No `onnx` package exists offline and no Piper voice can be fetched, so the loader (csrc/onnx_loader.cpp) is tested
against files written here, by an independent encoder: field numbers from the public onnx.proto
(ModelProto 1 ir_version / 7 graph / 8 opset_import; GraphProto 1 node / 2 name / 5 initializer; NodeProto 1 input /
2 output / 3 name / 4 op_type / 5 attribute; AttributeProto 1 name / 3 i / 8 ints / 20 type; TensorProto 1 dims /
2 data_type / 4 float_data / 7 int64_data / 8 name / 9 raw_data)."""
import struct

import numpy as np


def varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def key(field, wire):
    return varint((field << 3) | wire)


def ld(field, payload):
    return key(field, 2) + varint(len(payload)) + payload


def vi(field, v):
    return key(field, 0) + varint(v)


def tensor(name, dims, data, encoding="raw", packed_dims=True):
    """encoding: raw (raw_data, little-endian) | float_data (packed field 4) | float_unpacked (one fixed32 per element)."""
    out = b""
    if packed_dims:
        out += ld(1, b"".join(varint(d) for d in dims))
    else:
        out += b"".join(vi(1, d) for d in dims)
    a = np.ascontiguousarray(data)
    if a.dtype == np.float32:
        out += vi(2, 1)
        if encoding == "raw":
            out += ld(9, a.astype("<f4").tobytes())
        elif encoding == "float_data":
            out += ld(4, a.astype("<f4").tobytes())
        else:
            out += b"".join(key(4, 5) + struct.pack("<f", float(x)) for x in a.ravel())
    elif a.dtype == np.int64:
        out += vi(2, 7) + ld(9, a.astype("<i8").tobytes())
    else:
        raise TypeError(a.dtype)
    return out + ld(8, name.encode())


def attr_ints(name, values):
    return ld(1, name.encode()) + ld(8, b"".join(varint(v) for v in values)) + vi(20, 7)


def attr_int(name, v):
    return ld(1, name.encode()) + vi(3, v) + vi(20, 2)


def node(op, inputs, outputs, attrs=(), name=""):
    out = b"".join(ld(1, i.encode()) for i in inputs) + b"".join(ld(2, o.encode()) for o in outputs)
    if name:
        out += ld(3, name.encode())
    out += ld(4, op.encode())
    return out + b"".join(ld(5, a) for a in attrs)


def model(nodes, initializers, opset=15, ir_version=8, extra_unknown=True):
    g = b"".join(ld(1, n) for n in nodes) + ld(2, b"torch_jit") + b"".join(ld(5, t) for t in initializers)
    m = vi(1, ir_version) + ld(2, b"pytorch") + ld(3, b"2.0")  # producer name/version: skipped by the loader
    if extra_unknown:
        m += key(6, 1) + struct.pack("<q", 1) + key(15, 5) + struct.pack("<I", 7)  # fields the loader must skip by wire type
    m += ld(7, g) + ld(8, ld(1, b"") + vi(2, opset))
    return m


def scope_of(module):
    """torch.onnx scope name of a module path: 'flow.flows.0.enc.in_layers.1' → '/flow/flows.0/enc/in_layers.1'."""
    out = ""
    for part in module.split("."):
        out += ("." if part.isdigit() and out else "/") + part
    return out


def piper_voice_onnx(cfg, blob, layout, weight_norm=(), encodings=("raw", "float_data"), anonymous=(), extra_inits=()):
    """A Piper-shaped model: every blob tensor as an initializer (alternating encodings), Conv / ConvTranspose nodes for the
    generator carrying strides / dilations, some unrelated nodes and int64 initializers, optional weight-norm pairs."""
    inits, nodes = [], []
    anon = {}  # module → {"weight": anonymous name, "bias": …}: what the exporter leaves of a constant-folded weight-norm conv
    for i, t in enumerate(layout):
        name = t["name"]
        data = blob[t["offset"]:t["offset"] + t["count"]].reshape(t["shape"])
        module, _, leaf = name.rpartition(".")
        if module in anonymous and leaf in ("weight", "bias"):
            an = f"onnx::Conv_{9000 + i}"
            anon.setdefault(module, {})[leaf] = an
            inits.append(tensor(an, list(data.shape), data))
            continue
        if name in weight_norm:  # w = g·v/‖v‖  ⇒  store v = 3·w, g = ‖w‖ (per output row)
            w = data.reshape(data.shape[0], -1).astype(np.float64)
            g = np.sqrt((w ** 2).sum(1)).astype(np.float32)
            inits.append(tensor(name + "_g", [data.shape[0]] + [1] * (data.ndim - 1), g))
            inits.append(tensor(name + "_v", list(data.shape), (3.0 * data).astype(np.float32)))
        else:
            inits.append(tensor(name, list(data.shape), data, encodings[i % len(encodings)], packed_dims=(i % 3 != 0)))
    inits.append(tensor("dp.flows.3.some_int64_shape", [3], np.array([1, -1, 2], np.int64)))
    for nm, dims, data in extra_inits:
        inits.append(tensor(nm, dims, data))
    for module, names in anon.items():  # only the scope-named node still says which module these belong to
        if module.startswith("dec."):
            continue  # generator nodes are written below (with their anonymous inputs)
        nodes.append(node("Conv", [module + "_in", names["weight"], names["bias"]], [module + "_out"],
                          [attr_ints("dilations", [1]), attr_int("group", 1), attr_ints("strides", [1])], name=scope_of(module) + "/Conv"))
    pad = lambda k, d: (k * d - d) // 2
    nodes.append(node("Gather", ["enc_p.emb.weight", "input"], ["/enc_p/emb/Gather_output_0"]))
    ch = cfg.up_initial
    for u in range(cfg.n_ups):
        k, s = cfg.up_kernels[u], cfg.up_rates[u]
        um = anon.get(f"dec.ups.{u}", {})
        nodes.append(node("ConvTranspose", [f"x{u}", um.get("weight", f"dec.ups.{u}.weight"), um.get("bias", f"dec.ups.{u}.bias")], [f"y{u}"],
                          [attr_ints("dilations", [1]), attr_int("group", 1), attr_ints("kernel_shape", [k]),
                           attr_ints("pads", [(k - s) // 2] * 2), attr_ints("strides", [s])], name=f"/dec/ups.{u}/ConvTranspose"))
        ch //= 2
        for j in range(cfg.n_rb):
            rb = u * cfg.n_rb + j
            for d in range(cfg.rb_n_dil):
                dil = cfg.rb_dilations[j][d]
                kk = cfg.rb_kernels[j]
                names = [f"dec.resblocks.{rb}.convs1.{d}", f"dec.resblocks.{rb}.convs2.{d}"] if cfg.resblock_type == 1 \
                    else [f"dec.resblocks.{rb}.convs.{d}"]
                for q, nm in enumerate(names):
                    dd = dil if q == 0 else 1
                    wname = nm + (".weight_v" if nm + ".weight" in weight_norm else ".weight")
                    am = anon.get(nm, {})
                    nodes.append(node("Conv", [f"a{rb}_{d}_{q}", am.get("weight", wname), am.get("bias", nm + ".bias")], [f"b{rb}_{d}_{q}"],
                                      [attr_ints("dilations", [dd]), attr_int("group", 1), attr_ints("kernel_shape", [kk]),
                                       attr_ints("pads", [pad(kk, dd)] * 2), attr_ints("strides", [1])],
                                      name=(scope_of(nm) + "/Conv") if am else ""))
    nodes.append(node("Tanh", ["z"], ["output"]))
    return model(nodes, inits)
