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


def attr_float(name, v):
    return ld(1, name.encode()) + key(2, 5) + struct.pack("<f", float(v)) + vi(20, 1)


def attr_tensor(name, t):
    return ld(1, name.encode()) + ld(5, t) + vi(20, 4)


def value_info(name):
    return ld(1, name.encode())


def node(op, inputs, outputs, attrs=(), name=""):
    out = b"".join(ld(1, i.encode()) for i in inputs) + b"".join(ld(2, o.encode()) for o in outputs)
    if name:
        out += ld(3, name.encode())
    out += ld(4, op.encode())
    return out + b"".join(ld(5, a) for a in attrs)


def model(nodes, initializers, opset=15, ir_version=8, extra_unknown=True, inputs=("input", "input_lengths", "scales"), outputs=("output",)):
    g = b"".join(ld(1, n) for n in nodes) + ld(2, b"torch_jit") + b"".join(ld(5, t) for t in initializers)
    g += b"".join(ld(11, value_info(i)) for i in inputs) + b"".join(ld(12, value_info(o)) for o in outputs)
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


class GraphBuilder:
    """Emits the node list of a Piper VITS inference export (`VitsModel.infer`, opset 15) the way torch.onnx lays it out: scope-named
    nodes, scalars as Constant nodes, LayerNorm as the ReduceMean … Div chain, the relative-attention skew as Pad / Reshape / Slice,
    `torch.flip` as a step −1 Slice, masks as Mul. Shape plumbing (Shape / Gather / Concat of int64) is reduced to a few
    representative nodes — the verifier walks dataflow, not positions. `mut` names ONE deliberate defect for the refusal tests."""

    def __init__(self, cfg, wn, mut=None):
        self.c, self.wn, self.mut, self.nodes, self.k = cfg, wn, mut, [], 0

    def t(self, hint="t"):
        self.k += 1
        return f"/{hint}_output_{self.k}"

    def n(self, op, ins, attrs=(), name="", outs=None, hint=None):
        outs = outs or [self.t(hint or op)]
        self.nodes.append(node(op, ins, outs, attrs, name or f"/{op}_{len(self.nodes)}"))
        return outs[0] if len(outs) == 1 else outs

    def const(self, value, dtype=np.float32):
        a = np.asarray(value, dtype)
        return self.n("Constant", [], [attr_tensor("value", tensor("", list(a.shape), a))], hint="Constant")

    def conv(self, x, module, k, dil=1, pads=None, stride=1, transpose=False, scope=None):
        w, b = self.wn(module, "weight"), self.wn(module, "bias")
        pads = [(k * dil - dil) // 2] * 2 if pads is None else pads
        attrs = [attr_ints("dilations", [dil]), attr_int("group", 1), attr_ints("kernel_shape", [k]), attr_ints("pads", pads), attr_ints("strides", [stride])]
        op = "ConvTranspose" if transpose else "Conv"
        return self.n(op, [x, w] + ([b] if b else []), attrs, name=(scope or scope_of(module)) + "/" + op)

    def layer_norm(self, x, module):
        sc = scope_of(module)
        t = self.n("Transpose", [x], [attr_ints("perm", [0, 2, 1])], name=sc + "/Transpose")
        mu = self.n("ReduceMean", [t], [attr_ints("axes", [-1]), attr_int("keepdims", 1)], name=sc + "/ReduceMean")
        d = self.n("Sub", [t, mu], name=sc + "/Sub")
        sq = self.n("Pow", [d, self.const(2.0)], name=sc + "/Pow")
        var = self.n("ReduceMean", [sq], [attr_ints("axes", [-1]), attr_int("keepdims", 1)], name=sc + "/ReduceMean_1")
        eps = 1e-4 if self.mut == "ln_eps" and module.endswith("norm_layers_2.1") else 1e-5
        sd = self.n("Sqrt", [self.n("Add", [var, self.const(eps)], name=sc + "/Add")], name=sc + "/Sqrt")
        y = self.n("Div", [d, sd], name=sc + "/Div")
        y = self.n("Mul", [y, module + ".gamma"], name=sc + "/Mul")
        y = self.n("Add", [y, module + ".beta"], name=sc + "/Add_1")
        return self.n("Transpose", [y], [attr_ints("perm", [0, 2, 1])], name=sc + "/Transpose_1")

    def skew(self, x, sc, tag):  # Pad → Reshape → Pad → Reshape → Slice (relative ↔ absolute position, modeling_vits.py:963-997)
        shp = self.const([1, 2, -1], np.int64)
        p1 = self.n("Pad", [x, self.const([0] * 8, np.int64)], [], name=f"{sc}/Pad_{tag}")
        r1 = self.n("Reshape", [p1, shp], name=f"{sc}/Reshape_{tag}")
        p2 = self.n("Pad", [r1, self.const([0] * 6, np.int64)], [], name=f"{sc}/Pad_{tag}b")
        r2 = self.n("Reshape", [p2, self.const([1, 2, -1, 7], np.int64)], name=f"{sc}/Reshape_{tag}b")
        return self.n("Slice", [r2, self.const([0, 0], np.int64), self.const([9, 9], np.int64), self.const([2, 3], np.int64)], name=f"{sc}/Slice_{tag}")

    def rel_emb(self, name, sc, tag):  # _get_relative_embeddings: Pad + Slice of the [1, 2w+1, d] table
        p = self.n("Pad", [name, self.const([0] * 6, np.int64)], [], name=f"{sc}/Pad_rel_{tag}")
        return self.n("Slice", [p, self.const([0], np.int64), self.const([9], np.int64), self.const([1], np.int64)], name=f"{sc}/Slice_rel_{tag}")

    def build(self):
        c, H = self.c, self.c.hidden
        d = H // c.n_heads
        # ---- text encoder ----
        e = self.n("Gather", ["enc_p.emb.weight", "input"], name="/enc_p/emb/Gather")  # the first node (ONNXParsingTests.swift:36)
        if self.mut == "extra_first":
            self.nodes.insert(0, node("Identity", ["input"], ["/id0"], (), "/Identity_0"))
        e = self.n("Mul", [e, self.const(float(np.sqrt(H)) * (2.0 if self.mut == "emb_scale" else 1.0))], name="/enc_p/Mul")
        x = self.n("Transpose", [e], [attr_ints("perm", [0, 2, 1])], name="/enc_p/Transpose")
        # sequence mask: Shape / Gather / Range / Less / Cast / Unsqueeze
        tlen = self.n("Gather", [self.n("Shape", [x], name="/enc_p/Shape"), self.const(2, np.int64)], name="/enc_p/Gather_1")
        rng = self.n("Range", [self.const(0, np.int64), tlen, self.const(1, np.int64)], name="/enc_p/Range")
        lt = self.n("Less", [rng, self.n("Unsqueeze", ["input_lengths", self.const([1], np.int64)], name="/enc_p/Unsqueeze")], name="/enc_p/Less")
        x_mask = self.n("Cast", [self.n("Unsqueeze", [lt, self.const([1], np.int64)], name="/enc_p/Unsqueeze_1")], [attr_int("to", 1)], name="/enc_p/Cast")
        attn_mask = self.n("Mul", [self.n("Unsqueeze", [x_mask, self.const([2], np.int64)], name="/enc_p/encoder/Unsqueeze"),
                                   self.n("Unsqueeze", [x_mask, self.const([-1], np.int64)], name="/enc_p/encoder/Unsqueeze_1")], name="/enc_p/encoder/Mul")
        x = self.n("Mul", [x, x_mask], name="/enc_p/encoder/Mul_1")
        for l in range(c.n_layers):
            A = f"enc_p.encoder.attn_layers.{l}"
            sc = scope_of(A)
            heads = self.const([1, c.n_heads, d, -1], np.int64)
            qkv = []
            for nm in ("conv_q", "conv_k", "conv_v"):
                y = self.conv(x, f"{A}.{nm}", 1)
                y = self.n("Reshape", [y, heads], name=f"{sc}/Reshape_{nm}")
                qkv.append(self.n("Transpose", [y], [attr_ints("perm", [0, 1, 3, 2])], name=f"{sc}/Transpose_{nm}"))
            q, k, v = qkv
            qs = self.n("Div", [q, self.const(float(np.sqrt(d)) * (2.0 if self.mut == "q_scale" and l == 1 else 1.0))], name=f"{sc}/Div")
            scores = self.n("MatMul", [qs, self.n("Transpose", [k], [attr_ints("perm", [0, 1, 3, 2])], name=f"{sc}/Transpose_kT")], name=f"{sc}/MatMul")
            rk = self.n("Unsqueeze", [self.rel_emb(f"{A}.emb_rel_k", sc, "k"), self.const([0], np.int64)], name=f"{sc}/Unsqueeze_rk")
            rel_logits = self.n("MatMul", [qs, self.n("Transpose", [rk], [attr_ints("perm", [0, 1, 3, 2])], name=f"{sc}/Transpose_rk")], name=f"{sc}/MatMul_1")
            scores = self.n("Add", [scores, self.skew(rel_logits, sc, "r2a")], name=f"{sc}/Add")
            scores = self.n("Where", [self.n("Equal", [attn_mask, self.const(0.0)], name=f"{sc}/Equal"), self.const(-1e4), scores], name=f"{sc}/Where")
            p_attn = self.n("Relu" if self.mut == "softmax_op" and l == 2 else "Softmax", [scores],
                            [] if self.mut == "softmax_op" and l == 2 else [attr_int("axis", 1 if self.mut == "softmax_axis" and l == 0 else 3)], name=f"{sc}/Softmax")
            out = self.n("MatMul", [p_attn, v], name=f"{sc}/MatMul_2")
            rv = self.n("Unsqueeze", [self.rel_emb(f"{A}.emb_rel_v", sc, "v"), self.const([0], np.int64)], name=f"{sc}/Unsqueeze_rv")
            out = self.n("Add", [out, self.n("MatMul", [self.skew(p_attn, sc, "a2r"), rv], name=f"{sc}/MatMul_3")], name=f"{sc}/Add_1")
            out = self.n("Transpose", [out], [attr_ints("perm", [0, 1, 3, 2])], name=f"{sc}/Transpose_out")
            out = self.n("Reshape", [out, self.const([1, H, -1], np.int64)], name=f"{sc}/Reshape_out")
            y = self.conv(out, f"{A}.conv_o", 1)
            x = self.layer_norm(self.n("Add", [x, y], name=f"/enc_p/encoder/Add_{2 * l}"), f"enc_p.encoder.norm_layers_1.{l}")
            Fm = f"enc_p.encoder.ffn_layers.{l}"
            fs, kf = scope_of(Fm), c.ffn_kernel
            pl, pr = (kf - 1) // 2, kf // 2
            hcur = self.n("Mul", [x, x_mask], name=f"{fs}/Mul")
            hcur = self.n("Pad", [hcur, self.const([0, 0, pl, 0, 0, pr], np.int64)], [], name=f"{fs}/Pad")
            hcur = self.conv(hcur, f"{Fm}.conv_1", kf, pads=[0, 0])
            hcur = self.n("Tanh" if self.mut == "ffn_act" and l == 0 else "Relu", [hcur], name=f"{fs}/Relu")
            hcur = self.n("Mul", [hcur, x_mask], name=f"{fs}/Mul_1")
            hcur = self.n("Pad", [hcur, self.const([0, 0, pl, 0, 0, pr], np.int64)], [], name=f"{fs}/Pad_1")
            hcur = self.conv(hcur, f"{Fm}.conv_2", kf, pads=[0, 0])
            hcur = self.n("Mul", [hcur, x_mask], name=f"{fs}/Mul_2")
            x = self.layer_norm(self.n("Add", [x, hcur], name=f"/enc_p/encoder/Add_{2 * l + 1}"), f"enc_p.encoder.norm_layers_2.{l}")
        x = self.n("Mul", [x, x_mask], name="/enc_p/Mul_1")
        stats = self.n("Mul", [self.conv(x, "enc_p.proj", 1), x_mask], name="/enc_p/Mul_2")
        m_p, logs_p = self.n("Split", [stats, self.const([c.inter, c.inter], np.int64)], [attr_int("axis", 1)], name="/enc_p/Split", outs=[self.t("Split"), self.t("Split")])
        # ---- durations → alignment (skeleton: only supported ops, not verified in detail) ----
        if c.dp_present:
            dpn = self.n("RandomNormalLike", [self.n("Slice", [x, self.const([0], np.int64), self.const([2], np.int64), self.const([1], np.int64)], name="/dp/Slice")], name="/dp/RandomNormalLike")
            logw = self.n("Mul", [dpn, self.n("Gather", ["scales", self.const(2, np.int64)], name="/Gather_noise_w")], name="/dp/Mul")
        else:
            logw = self.n("Slice", [x, self.const([0], np.int64), self.const([1], np.int64), self.const([1], np.int64)], name="/Slice_logw")
        w = self.n("Mul", [self.n("Mul", [self.n("Exp", [logw], name="/Exp"), x_mask], name="/Mul"), self.n("Gather", ["scales", self.const(1, np.int64)], name="/Gather_len")], name="/Mul_1")
        w_ceil = self.n("Ceil", [w], name="/Ceil")
        ylen = self.n("Cast", [self.n("Clip", [self.n("ReduceSum", [w_ceil, self.const([1, 2], np.int64)], name="/ReduceSum"), self.const(1.0)], name="/Clip")], [attr_int("to", 7)], name="/Cast")
        y_mask = self.n("Cast", [self.n("Unsqueeze", [self.n("Less", [self.n("Range", [self.const(0, np.int64), ylen, self.const(1, np.int64)], name="/Range"), ylen], name="/Less"),
                                                      self.const([1], np.int64)], name="/Unsqueeze")], [attr_int("to", 1)], name="/Cast_1")
        cum = self.n("CumSum", [w_ceil, self.const(-1, np.int64)], name="/CumSum")
        path = self.n("Cast", [self.n("Less", [self.n("Unsqueeze", [y_mask, self.const([-1], np.int64)], name="/Unsqueeze_1"), cum], name="/Less_1")], [attr_int("to", 1)], name="/Cast_2")
        attn = self.n("Squeeze", [self.n("Sub", [path, self.n("Pad", [path, self.const([0] * 8, np.int64)], [], name="/Pad")], name="/Sub"), self.const([1], np.int64)], name="/Squeeze")

        def expand(tn, tag):
            y = self.n("MatMul", [attn, self.n("Transpose", [tn], [attr_ints("perm", [0, 2, 1])], name=f"/Transpose_{tag}")], name=f"/MatMul_{tag}")
            return self.n("Transpose", [y], [attr_ints("perm", [0, 2, 1])], name=f"/Transpose_{tag}b")
        m_f, logs_f = expand(m_p, "m"), expand(logs_p, "logs")
        nz = self.n("RandomNormalLike", [m_f], name="/RandomNormalLike")
        if self.mut == "extra_rng":
            nz = self.n("Add", [nz, self.n("RandomNormalLike", [m_f], name="/RandomNormalLike_x")], name="/Add_rng")
        z = self.n("Add", [m_f, self.n("Mul", [self.n("Mul", [nz, self.n("Exp", [logs_f], name="/Exp_1")], name="/Mul_2"),
                                               self.n("Gather", ["scales", self.const(0, np.int64)], name="/Gather_noise")], name="/Mul_3")], name="/Add")
        # ---- flow, reverse: Flip, coupling, Flip, coupling … ----
        half = c.inter // 2
        for f in reversed(range(c.n_flows)):
            Fl = f"flow.flows.{2 * f}"
            sc = scope_of(Fl)
            if not (self.mut == "no_flip" and f == 1):
                z = self.n("Slice", [z, self.const([-1], np.int64), self.const([-(2 ** 63) + 1], np.int64), self.const([1], np.int64), self.const([-1], np.int64)],
                           name=f"/flow/flows.{2 * f + 1}/Slice")
            x0, x1 = self.n("Split", [z, self.const([half, half], np.int64)], [attr_int("axis", 1)], name=f"{sc}/Split", outs=[self.t("Split"), self.t("Split")])
            hcur = self.n("Mul", [self.conv(x0, f"{Fl}.pre", 1), y_mask], name=f"{sc}/Mul")
            skip = None
            for i in range(c.wn_layers):
                W = f"{Fl}.enc"
                ws = scope_of(W)
                xin = self.conv(hcur, f"{W}.in_layers.{i}", c.wn_kernel, dil=(2 if self.mut == "wn_dilation" and f == 0 and i == 1 else 1),
                                pads=[(c.wn_kernel - 1) // 2] * 2)
                act = self.n("Add", [xin, self.const(np.zeros((1, 2 * H, 1), np.float32))], name=f"{ws}/Add_g{i}")
                ta = self.n("Slice", [act, self.const([0], np.int64), self.const([H], np.int64), self.const([1], np.int64)], name=f"{ws}/Slice_t{i}")
                sa = self.n("Slice", [act, self.const([H], np.int64), self.const([2 * H], np.int64), self.const([1], np.int64)], name=f"{ws}/Slice_s{i}")
                swap = self.mut == "gate_swap" and f == 2 and i == 0
                acts = self.n("Mul", [self.n("Sigmoid" if swap else "Tanh", [ta], name=f"{ws}/Tanh_{i}"), self.n("Tanh" if swap else "Sigmoid", [sa], name=f"{ws}/Sigmoid_{i}")], name=f"{ws}/Mul_{i}")
                rs = self.conv(acts, f"{W}.res_skip_layers.{i}", 1)
                if i + 1 < c.wn_layers:
                    res = self.n("Slice", [rs, self.const([0], np.int64), self.const([H], np.int64), self.const([1], np.int64)], name=f"{ws}/Slice_r{i}")
                    hcur = self.n("Mul", [self.n("Add", [hcur, res], name=f"{ws}/Add_r{i}"), y_mask], name=f"{ws}/Mul_r{i}")
                    sk = self.n("Slice", [rs, self.const([H], np.int64), self.const([2 * H], np.int64), self.const([1], np.int64)], name=f"{ws}/Slice_k{i}")
                else:
                    sk = rs
                skip = sk if skip is None else self.n("Add", [skip, sk], name=f"{ws}/Add_k{i}")
            m = self.n("Mul", [self.conv(self.n("Mul", [skip, y_mask], name=f"{sc}/Mul_1"), f"{Fl}.post", 1), y_mask], name=f"{sc}/Mul_2")
            x1 = self.n("Mul", [self.n("Sub", [x1, m], name=f"{sc}/Sub"), y_mask], name=f"{sc}/Mul_3")
            z = self.n("Concat", [x0, x1], [attr_int("axis", 1)], name=f"{sc}/Concat")
        # ---- HiFi-GAN generator ----
        g = self.conv(self.n("Mul", [z, y_mask], name="/Mul_4"), "dec.conv_pre", 7)
        for u in range(c.n_ups):
            ku, su = c.up_kernels[u], c.up_rates[u]
            g = self.n("LeakyRelu", [g], [attr_float("alpha", 0.2 if self.mut == "lrelu_alpha" and u == 1 else 0.1)], name=f"/dec/LeakyRelu_{u}")
            g = self.conv(g, f"dec.ups.{u}", ku, pads=[(ku - su) // 2] * 2, stride=su, transpose=True)
            xs = None
            for j in range(c.n_rb):
                rb = u * c.n_rb + j
                R = f"dec.resblocks.{rb}"
                rs_, xr, kk = scope_of(R), g, c.rb_kernels[j]
                for di in range(c.rb_n_dil):
                    dil = c.rb_dilations[j][di]
                    xt = self.n("LeakyRelu", [xr], [attr_float("alpha", 0.1)], name=f"{rs_}/LeakyRelu_{2 * di}")
                    if self.mut == "extra_node" and rb == 1 and di == 0:
                        xt = self.n("Relu", [xt], name=f"{rs_}/Relu_extra")
                    if c.resblock_type == 1:
                        xt = self.conv(xt, f"{R}.convs1.{di}", kk, dil=dil)
                        xt = self.n("LeakyRelu", [xt], [attr_float("alpha", 0.1)], name=f"{rs_}/LeakyRelu_{2 * di + 1}")
                        xt = self.conv(xt, f"{R}.convs2.{di}", kk, dil=1)
                    else:
                        xt = self.conv(xt, f"{R}.convs.{di}", kk, dil=dil, pads=([0, 0] if self.mut == "conv_pads" and rb == 0 and di == 1 else None))
                    xr = self.n("Sub" if self.mut == "res_op" and rb == 2 and di == 0 else "Add", [xt, xr], name=f"{rs_}/Add_{di}")
                xs = xr if xs is None else self.n("Add", [xs, xr], name=f"/dec/Add_{u}_{j}")
            g = self.n("Div", [xs, self.const(float(c.n_rb + (1 if self.mut == "mrf_div" and u == 0 else 0)))], name=f"/dec/Div_{u}")
        g = self.n("LeakyRelu", [g], [attr_float("alpha", 0.01)], name="/dec/LeakyRelu_post")
        g = self.conv(g, "dec.conv_post", 7)
        self.n("Erf" if self.mut == "out_act" else "Tanh", [g], name="/dec/Tanh", outs=["output"])
        if self.mut == "bad_op":
            self.n("Einsum", [g], name="/Einsum_0")
        return self.nodes


def piper_voice_onnx(cfg, blob, layout, weight_norm=(), encodings=("raw", "float_data"), anonymous=(), extra_inits=(), graph="full", mut=None,
                     opset=15, inputs=("input", "input_lengths", "scales"), outputs=("output",)):
    """A Piper-shaped model: every blob tensor as an initializer (alternating encodings), optional weight-norm pairs and anonymous
    (constant-folded) conv weights, and the node list — graph="full": the whole inference graph (GraphBuilder; what the verifier
    accepts), graph="sparse": only the generator's Conv / ConvTranspose nodes plus a few strays (what round 2 wrote; the verifier
    must refuse it). `mut` plants one defect in the full graph."""
    inits = []
    anon = {}  # module → {"weight": anonymous name, "bias": …}: what the exporter leaves of a constant-folded weight-norm conv
    names = set(t["name"] for t in layout)
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

    def wn(module, leaf):  # the name a node uses for a module's weight / bias
        if module in anon and leaf in anon[module]:
            return anon[module][leaf]
        full = f"{module}.{leaf}"
        if full in weight_norm:
            return full + "_v"
        return full if full in names else None

    if graph == "full":
        nodes = GraphBuilder(cfg, wn, mut).build()
        return model(nodes, inits, opset=opset, inputs=inputs, outputs=outputs)
    nodes = []
    for module, nms in anon.items():  # only the scope-named node still says which module these belong to
        if module.startswith("dec."):
            continue  # generator nodes are written below (with their anonymous inputs)
        nodes.append(node("Conv", [module + "_in", nms["weight"], nms["bias"]], [module + "_out"],
                          [attr_ints("dilations", [1]), attr_int("group", 1), attr_ints("strides", [1])], name=scope_of(module) + "/Conv"))
    pad = lambda k, d: (k * d - d) // 2
    nodes.append(node("Gather", ["enc_p.emb.weight", "input"], ["/enc_p/emb/Gather_output_0"]))
    ch = cfg.up_initial
    for u in range(cfg.n_ups):
        k, s = cfg.up_kernels[u], cfg.up_rates[u]
        nodes.append(node("ConvTranspose", [f"x{u}", wn(f"dec.ups.{u}", "weight"), wn(f"dec.ups.{u}", "bias")], [f"y{u}"],
                          [attr_ints("dilations", [1]), attr_int("group", 1), attr_ints("kernel_shape", [k]),
                           attr_ints("pads", [(k - s) // 2] * 2), attr_ints("strides", [s])], name=f"/dec/ups.{u}/ConvTranspose"))
        ch //= 2
        for j in range(cfg.n_rb):
            rb = u * cfg.n_rb + j
            for d in range(cfg.rb_n_dil):
                dil = cfg.rb_dilations[j][d]
                kk = cfg.rb_kernels[j]
                nms = [f"dec.resblocks.{rb}.convs1.{d}", f"dec.resblocks.{rb}.convs2.{d}"] if cfg.resblock_type == 1 \
                    else [f"dec.resblocks.{rb}.convs.{d}"]
                for q, nm in enumerate(nms):
                    dd = dil if q == 0 else 1
                    nodes.append(node("Conv", [f"a{rb}_{d}_{q}", wn(nm, "weight"), wn(nm, "bias")], [f"b{rb}_{d}_{q}"],
                                      [attr_ints("dilations", [dd]), attr_int("group", 1), attr_ints("kernel_shape", [kk]),
                                       attr_ints("pads", [pad(kk, dd)] * 2), attr_ints("strides", [1])],
                                      name=(scope_of(nm) + "/Conv") if nm in anon else ""))
    nodes.append(node("Tanh", ["z"], ["output"]))
    return model(nodes, inits, opset=opset, inputs=inputs, outputs=outputs)
