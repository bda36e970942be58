"""PyTorch-CPU fp32 restatement of the Piper VITS inference graph — TEST INFRASTRUCTURE.

Independent of oracle/piper_oracle.c (different language, library ops instead of loops). Three users:
  * tools/gen_golden.py writes tests/golden/*.npz from it (run in the build container only),
  * tests/test_hf_crosscheck.py pins it against the third-party VITS implementation in `transformers.models.vits`,
  * bench.py's `cpu_baseline` leg times it on the GPU box's host cores (SURVEY.md §8d fallback (ii)).
Nothing under piper-swift_amd/ imports it.
"""
import os

import numpy as np
import torch
import torch.nn.functional as Fn

import katdata as kd
import piper_hip as ph  # host-only helpers: config presets, blob layout


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


# ------------------------------------------------------------------ torch reference model (Piper VITS, infer path)
class Ref:
    def __init__(self, cfg, blob):
        self.cfg = cfg
        self.w = {e["name"]: t(blob[e["offset"]:e["offset"] + e["count"]].reshape(e["shape"])) for e in ph.blob_layout(cfg)}

    def conv(self, name, x, pad=(0, 0), dil=1):
        x = Fn.pad(x, pad)
        return Fn.conv1d(x, self.w[name + ".weight"], self.w.get(name + ".bias"), dilation=dil)

    @staticmethod
    def rel_embeddings(emb, T, window):
        pad_length = max(T - (window + 1), 0)
        start = max((window + 1) - T, 0)
        e = emb.unsqueeze(0)
        if pad_length > 0:
            e = Fn.pad(e, [0, 0, pad_length, pad_length, 0, 0])
        return e[:, start:start + 2 * T - 1]

    @staticmethod
    def rel_to_abs(x):
        b, h, L, _ = x.shape
        x = Fn.pad(x, [0, 1])
        x = x.reshape(b, h, L * 2 * L)
        x = Fn.pad(x, [0, L - 1])
        return x.reshape(b, h, L + 1, 2 * L - 1)[:, :, :L, L - 1:]

    @staticmethod
    def abs_to_rel(x):
        b, h, L, _ = x.shape
        x = Fn.pad(x, [0, L - 1])
        x = x.reshape(b, h, L * (2 * L - 1))
        x = Fn.pad(x, [L, 0])
        return x.reshape(b, h, L, 2 * L)[:, :, :, 1:]

    @classmethod
    def attention(cls, q, k, v, ek, ev, heads, window):
        b, C, T = q.shape
        d = C // heads
        q = q.view(b, heads, d, T).transpose(2, 3)
        k = k.view(b, heads, d, T).transpose(2, 3)
        v = v.view(b, heads, d, T).transpose(2, 3)
        qs = q / (d ** 0.5)
        scores = torch.matmul(qs, k.transpose(-2, -1))
        rk = cls.rel_embeddings(ek, T, window)
        scores = scores + cls.rel_to_abs(torch.matmul(qs, rk.unsqueeze(0).transpose(-2, -1)))
        p = Fn.softmax(scores, dim=-1)
        out = torch.matmul(p, v)
        rv = cls.rel_embeddings(ev, T, window)
        out = out + torch.matmul(cls.abs_to_rel(p), rv.unsqueeze(0))
        return out.transpose(2, 3).contiguous().view(b, C, T)

    @staticmethod
    def layernorm(x, g, b, eps=1e-5):
        return Fn.layer_norm(x.transpose(1, -1), (x.shape[1],), g, b, eps).transpose(1, -1)

    def text_encoder(self, ids):
        c = self.cfg
        x = self.w["enc_p.emb.weight"][t(np.asarray(ids, np.int64))] * (c.hidden ** 0.5)
        x = x.unsqueeze(0).transpose(1, 2)
        for l in range(c.n_layers):
            P = f"enc_p.encoder.attn_layers.{l}."
            q, k, v = (self.conv(P + n, x) for n in ("conv_q", "conv_k", "conv_v"))
            y = self.attention(q, k, v, self.w[P + "emb_rel_k"], self.w[P + "emb_rel_v"], c.n_heads, c.window)
            y = self.conv(P + "conv_o", y)
            x = self.layernorm(x + y, self.w[f"enc_p.encoder.norm_layers_1.{l}.gamma"], self.w[f"enc_p.encoder.norm_layers_1.{l}.beta"])
            kf = c.ffn_kernel
            pad = ((kf - 1) // 2, kf // 2)
            y = torch.relu(self.conv(f"enc_p.encoder.ffn_layers.{l}.conv_1", x, pad))
            y = self.conv(f"enc_p.encoder.ffn_layers.{l}.conv_2", y, pad)
            x = self.layernorm(x + y, self.w[f"enc_p.encoder.norm_layers_2.{l}.gamma"], self.w[f"enc_p.encoder.norm_layers_2.{l}.beta"])
        return x, self.conv("enc_p.proj", x)

    @staticmethod
    def wn_layer(x, skip, w_in, b_in, w_rs, b_rs, K, dil, last):
        C = x.shape[1]
        pad = (K * dil - dil) // 2
        xin = Fn.conv1d(x, w_in, b_in, dilation=dil, padding=pad)
        acts = torch.tanh(xin[:, :C]) * torch.sigmoid(xin[:, C:])
        rs = Fn.conv1d(acts, w_rs, b_rs)
        if skip is None:
            skip = torch.zeros_like(x)
        if last:
            return None, skip + rs
        return x + rs[:, :C], skip + rs[:, C:]

    def flow_reverse(self, x):
        c = self.cfg
        half = c.inter // 2
        for f in reversed(range(c.n_flows)):
            x = torch.flip(x, [1])
            x0, x1 = x[:, :half], x[:, half:]
            P = f"flow.flows.{2 * f}."
            h = self.conv(P + "pre", x0)
            skip = None
            for i in range(c.wn_layers):
                last = i + 1 == c.wn_layers
                hn, skip = self.wn_layer(h, skip, self.w[P + f"enc.in_layers.{i}.weight"], self.w[P + f"enc.in_layers.{i}.bias"],
                                         self.w[P + f"enc.res_skip_layers.{i}.weight"], self.w[P + f"enc.res_skip_layers.{i}.bias"],
                                         c.wn_kernel, 1, last)
                if not last:
                    h = hn
            m = self.conv(P + "post", skip)
            x = torch.cat([x0, x1 - m], 1)
        return x

    @staticmethod
    def resblock(type_, x, K, dils, ws, bs, slope=0.1):
        for i, d in enumerate(dils):
            if type_ == 1:
                xt = Fn.conv1d(Fn.leaky_relu(x, slope), ws[2 * i], bs[2 * i], dilation=d, padding=(K * d - d) // 2)
                xt = Fn.conv1d(Fn.leaky_relu(xt, slope), ws[2 * i + 1], bs[2 * i + 1], padding=(K - 1) // 2)
            else:
                xt = Fn.conv1d(Fn.leaky_relu(x, slope), ws[i], bs[i], dilation=d, padding=(K * d - d) // 2)
            x = xt + x
        return x

    def generator(self, z):
        c = self.cfg
        x = self.conv("dec.conv_pre", z, (3, 3))
        for u in range(c.n_ups):
            x = Fn.leaky_relu(x, 0.1)
            k, s = c.up_kernels[u], c.up_rates[u]
            x = Fn.conv_transpose1d(x, self.w[f"dec.ups.{u}.weight"], self.w[f"dec.ups.{u}.bias"], stride=s, padding=(k - s) // 2)
            xs = None
            for j in range(c.n_rb):
                rb = u * c.n_rb + j
                dils = [c.rb_dilations[j][d] for d in range(c.rb_n_dil)]
                if c.resblock_type == 1:
                    ws, bs = [], []
                    for d in range(c.rb_n_dil):
                        for cn in ("convs1", "convs2"):
                            ws.append(self.w[f"dec.resblocks.{rb}.{cn}.{d}.weight"])
                            bs.append(self.w[f"dec.resblocks.{rb}.{cn}.{d}.bias"])
                else:
                    ws = [self.w[f"dec.resblocks.{rb}.convs.{d}.weight"] for d in range(c.rb_n_dil)]
                    bs = [self.w[f"dec.resblocks.{rb}.convs.{d}.bias"] for d in range(c.rb_n_dil)]
                r = self.resblock(c.resblock_type, x, c.rb_kernels[j], dils, ws, bs)
                xs = r if xs is None else xs + r
            x = xs / c.n_rb
        x = Fn.leaky_relu(x)
        x = Fn.conv1d(x, self.w["dec.conv_post.weight"], None, padding=3)
        return torch.tanh(x)

    # ---- stochastic duration predictor, reverse (VITS models.StochasticDurationPredictor / modules.DDSConv / ConvFlow /
    # ElementwiseAffine / transforms.piecewise_rational_quadratic_transform; Piper infer: w = exp(logw)·mask·length_scale)
    def dds(self, base, x, g=None):
        c = self.cfg
        if g is not None:
            x = x + g
        K = c.dp_kernel
        for i in range(c.dp_dds_layers):
            dil = K ** i
            y = Fn.conv1d(x, self.w[f"{base}.convs.convs_sep.{i}.weight"], self.w[f"{base}.convs.convs_sep.{i}.bias"], dilation=dil,
                          padding=(K * dil - dil) // 2, groups=x.shape[1])
            y = Fn.gelu(self.layernorm(y, self.w[f"{base}.convs.norms_1.{i}.gamma"], self.w[f"{base}.convs.norms_1.{i}.beta"]))
            y = self.conv(f"{base}.convs.convs_1x1.{i}", y)
            y = Fn.gelu(self.layernorm(y, self.w[f"{base}.convs.norms_2.{i}.gamma"], self.w[f"{base}.convs.norms_2.{i}.beta"]))
            x = x + y
        return x

    def spline_inverse(self, x1, h):
        """x1 [1,1,T]; h [1, 3·bins − 1, T] (ConvFlow.proj output) → inverse rational-quadratic spline with linear tails."""
        c = self.cfg
        nb, B, fc = c.dp_bins, float(c.dp_tail_bound), float(c.hidden)
        h = h[0].transpose(0, 1)  # [T, 29]
        uw, uh, ud = h[:, :nb] / (fc ** 0.5), h[:, nb:2 * nb] / (fc ** 0.5), h[:, 2 * nb:]
        mbw = mbh = md = 1e-3
        const = float(np.log(np.exp(1 - md) - 1))
        ud = Fn.pad(ud, (1, 1), value=const)
        x = x1[0, 0]
        inside = (x >= -B) & (x <= B)
        widths = mbw + (1 - mbw * nb) * Fn.softmax(uw, -1)
        cw = Fn.pad(torch.cumsum(widths, -1), (1, 0)) * (2 * B) - B
        cw[:, 0], cw[:, -1] = -B, B
        widths = cw[:, 1:] - cw[:, :-1]
        derivs = md + Fn.softplus(ud)
        heights = mbh + (1 - mbh * nb) * Fn.softmax(uh, -1)
        chh = Fn.pad(torch.cumsum(heights, -1), (1, 0)) * (2 * B) - B
        chh[:, 0], chh[:, -1] = -B, B
        heights = chh[:, 1:] - chh[:, :-1]
        loc = chh.clone()
        loc[:, -1] += 1e-6
        idx = (torch.sum(x[:, None] >= loc, -1) - 1).clamp(0, nb - 1)[:, None]
        g = lambda tns: tns.gather(-1, idx)[:, 0]
        icw, ibw, ich, ih = g(cw), g(widths), g(chh), g(heights)
        idl = g(heights / widths)
        d0, d1 = g(derivs), g(derivs[:, 1:])
        i1 = d0 + d1 - 2 * idl
        i2 = x - ich
        i3 = i2 * i1
        a = ih * (idl - d0) + i3
        b = ih * d0 - i3
        cc = -idl * i2
        root = (2 * cc) / (-b - torch.sqrt(b * b - 4 * a * cc))
        out = torch.where(inside, root * ibw + icw, x)
        return out.reshape(1, 1, -1)

    def duration_logw(self, enc_out, dp_noise, noise_w):
        """enc_out [1,H,T] (text-encoder output), dp_noise [2,T] (the `dp` RandomNormalLike tensor) → logw [T]."""
        c = self.cfg
        x = self.conv("dp.pre", enc_out)
        x = self.dds("dp", x)
        x = self.conv("dp.proj", x)
        z = t(np.ascontiguousarray(dp_noise, np.float32)).reshape(1, 2, -1) * noise_w
        for f in range(2 * c.dp_n_flows - 1, 1, -2):  # ConvFlows 7, 5, 3 — each preceded by a Flip
            z = torch.flip(z, [1])
            z0, z1 = z[:, :1], z[:, 1:]
            h = self.conv(f"dp.flows.{f}.pre", z0)
            h = self.dds(f"dp.flows.{f}", h, g=x)
            h = self.conv(f"dp.flows.{f}.proj", h)
            z = torch.cat([z0, self.spline_inverse(z1, h)], 1)
        z = torch.flip(z, [1])
        z = (z - self.w["dp.flows.0.m"].reshape(1, 2, 1)) * torch.exp(-self.w["dp.flows.0.logs"].reshape(1, 2, 1))
        return z[0, 0]

    def durations(self, ids, dp_noise, noise_w, length_scale):
        enc, _ = self.text_encoder(ids)
        logw = self.duration_logw(enc, dp_noise, noise_w)
        w = torch.exp(logw) * length_scale
        return torch.ceil(w).to(torch.int64).numpy(), logw.numpy()

    def synthesize(self, ids, durations, noise, noise_scale):
        c = self.cfg
        enc, stats = self.text_encoder(ids)
        m_p, logs_p = stats[:, :c.inter], stats[:, c.inter:]
        T, F = len(ids), int(np.sum(durations))
        attn = torch.zeros(1, F, T)
        f = 0
        for i, dcount in enumerate(durations):
            attn[0, f:f + dcount, i] = 1.0
            f += dcount
        m_e = torch.matmul(attn, m_p.transpose(1, 2)).transpose(1, 2)
        l_e = torch.matmul(attn, logs_p.transpose(1, 2)).transpose(1, 2)
        z_p = m_e + t(noise).reshape(1, c.inter, F) * torch.exp(l_e) * noise_scale
        z = self.flow_reverse(z_p)
        o = self.generator(z)
        return dict(enc_out=enc, m_p=m_p, logs_p=logs_p, z_p=z_p, z=z, audio=o.reshape(-1))


# ------------------------------------------------------------------ HF cross-checks (generation time only)
def hf_crosscheck(cfg_m, blob_m, cfg_h, blob_h):
    os.environ["HF_HUB_OFFLINE"] = "1"
    from transformers.models.vits import modeling_vits as mv
    from transformers.models.vits.configuration_vits import VitsConfig
    rep = {}
    hc = VitsConfig()  # defaults = Piper "high" decoder geometry, hidden 192, 2 heads, window 4
    ref_m, ref_h = Ref(cfg_m, blob_m), Ref(cfg_h, blob_h)
    with torch.no_grad():
        # attention + encoder layer 0
        lay = mv.VitsEncoderLayer(hc).eval()
        P = "enc_p.encoder.attn_layers.0."
        for hn, pn in (("q_proj", "conv_q"), ("k_proj", "conv_k"), ("v_proj", "conv_v"), ("out_proj", "conv_o")):
            getattr(lay.attention, hn).weight.copy_(ref_m.w[P + pn + ".weight"][:, :, 0])
            getattr(lay.attention, hn).bias.copy_(ref_m.w[P + pn + ".bias"])
        lay.attention.emb_rel_k.copy_(ref_m.w[P + "emb_rel_k"].unsqueeze(0))
        lay.attention.emb_rel_v.copy_(ref_m.w[P + "emb_rel_v"].unsqueeze(0))
        lay.layer_norm.weight.copy_(ref_m.w["enc_p.encoder.norm_layers_1.0.gamma"])
        lay.layer_norm.bias.copy_(ref_m.w["enc_p.encoder.norm_layers_1.0.beta"])
        lay.final_layer_norm.weight.copy_(ref_m.w["enc_p.encoder.norm_layers_2.0.gamma"])
        lay.final_layer_norm.bias.copy_(ref_m.w["enc_p.encoder.norm_layers_2.0.beta"])
        for hn, pn in (("conv_1", "conv_1"), ("conv_2", "conv_2")):
            getattr(lay.feed_forward, hn).weight.copy_(ref_m.w[f"enc_p.encoder.ffn_layers.0.{pn}.weight"])
            getattr(lay.feed_forward, hn).bias.copy_(ref_m.w[f"enc_p.encoder.ffn_layers.0.{pn}.bias"])
        for T in (3, 14, 40):
            x = t(kd.sym(99 + T, (1, 192, T)))
            q, k, v = (ref_m.conv(P + n, x) for n in ("conv_q", "conv_k", "conv_v"))
            mine = ref_m.conv(P + "conv_o", Ref.attention(q, k, v, ref_m.w[P + "emb_rel_k"], ref_m.w[P + "emb_rel_v"], 2, 4))
            theirs, _ = lay.attention(x.transpose(1, 2))
            rep[f"attention_T{T}"] = float((mine - theirs.transpose(1, 2)).abs().max())
            # whole layer
            one = torch.ones(1, T, 1)
            theirs = lay(x.transpose(1, 2), one)[0].transpose(1, 2)
            y = ref_m.layernorm(x + mine, lay.layer_norm.weight, lay.layer_norm.bias)
            f1 = torch.relu(ref_m.conv("enc_p.encoder.ffn_layers.0.conv_1", y, (1, 1)))
            f2 = ref_m.conv("enc_p.encoder.ffn_layers.0.conv_2", f1, (1, 1))
            mine2 = ref_m.layernorm(y + f2, lay.final_layer_norm.weight, lay.final_layer_norm.bias)
            rep[f"encoder_layer_T{T}"] = float((mine2 - theirs).abs().max())
        # flow
        blk = mv.VitsResidualCouplingBlock(hc).eval()
        for f in range(4):
            L = blk.flows[f]
            P = f"flow.flows.{2 * f}."
            L.conv_pre.weight.copy_(ref_m.w[P + "pre.weight"]); L.conv_pre.bias.copy_(ref_m.w[P + "pre.bias"])
            L.conv_post.weight.copy_(ref_m.w[P + "post.weight"]); L.conv_post.bias.copy_(ref_m.w[P + "post.bias"])
            for i in range(4):
                for lst, nm in ((L.wavenet.in_layers, "in_layers"), (L.wavenet.res_skip_layers, "res_skip_layers")):
                    torch.nn.utils.parametrize.remove_parametrizations(lst[i], "weight")
                    lst[i].weight.copy_(ref_m.w[P + f"enc.{nm}.{i}.weight"])
                    lst[i].bias.copy_(ref_m.w[P + f"enc.{nm}.{i}.bias"])
        zp = t(kd.sym(77, (1, 192, 20)))
        rep["flow_reverse"] = float((ref_m.flow_reverse(zp) - blk(zp, torch.ones(1, 1, 20), reverse=True)).abs().max())
        # HiFi-GAN (high geometry = VitsConfig defaults)
        gen = mv.VitsHifiGan(hc).eval()
        gen.conv_pre.weight.copy_(ref_h.w["dec.conv_pre.weight"]); gen.conv_pre.bias.copy_(ref_h.w["dec.conv_pre.bias"])
        for u in range(4):
            gen.upsampler[u].weight.copy_(ref_h.w[f"dec.ups.{u}.weight"]); gen.upsampler[u].bias.copy_(ref_h.w[f"dec.ups.{u}.bias"])
        for rb in range(12):
            for d in range(3):
                for cn, lst in (("convs1", gen.resblocks[rb].convs1), ("convs2", gen.resblocks[rb].convs2)):
                    lst[d].weight.copy_(ref_h.w[f"dec.resblocks.{rb}.{cn}.{d}.weight"])
                    lst[d].bias.copy_(ref_h.w[f"dec.resblocks.{rb}.{cn}.{d}.bias"])
        gen.conv_post.weight.copy_(ref_h.w["dec.conv_post.weight"])
        z = t(kd.sym(55, (1, 192, 4)))
        rep["hifigan_high"] = float((ref_h.generator(z) - gen(z)).abs().max())
        # stochastic duration predictor, reverse mode (noise injected by patching torch.randn for the call)
        if getattr(cfg_m, "dp_present", 0):
            sdp = mv.VitsStochasticDurationPredictor(hc).eval()
            def cp(dst, name):
                dst.weight.copy_(ref_m.w[name + ".weight"]); dst.bias.copy_(ref_m.w[name + ".bias"])
            def cp_dds(dds, base):
                for i in range(cfg_m.dp_dds_layers):
                    cp(dds.convs_dilated[i], f"{base}.convs.convs_sep.{i}")
                    cp(dds.convs_pointwise[i], f"{base}.convs.convs_1x1.{i}")
                    for j, lst in ((1, dds.norms_1), (2, dds.norms_2)):
                        lst[i].weight.copy_(ref_m.w[f"{base}.convs.norms_{j}.{i}.gamma"]); lst[i].bias.copy_(ref_m.w[f"{base}.convs.norms_{j}.{i}.beta"])
            cp(sdp.conv_pre, "dp.pre"); cp(sdp.conv_proj, "dp.proj"); cp_dds(sdp.conv_dds, "dp")
            sdp.flows[0].translate.copy_(ref_m.w["dp.flows.0.m"]); sdp.flows[0].log_scale.copy_(ref_m.w["dp.flows.0.logs"])
            for j in range(2, cfg_m.dp_n_flows + 1):  # HF flows[j] = VITS dp.flows[2j − 1]
                fl = sdp.flows[j]
                cp(fl.conv_pre, f"dp.flows.{2 * j - 1}.pre"); cp(fl.conv_proj, f"dp.flows.{2 * j - 1}.proj"); cp_dds(fl.conv_dds, f"dp.flows.{2 * j - 1}")
            for T in (5, 14, 50):
                enc = t(kd.sym(300 + T, (1, 192, T)))
                nz = kd.sym(310 + T, (2, T), 1.7320508)
                mine = ref_m.duration_logw(enc, nz, 0.8)
                orig = torch.randn
                torch.randn = lambda *a, **k: t(nz).reshape(1, 2, T)
                try:
                    theirs = sdp(enc, torch.ones(1, 1, T), reverse=True, noise_scale=0.8)[0, 0]
                finally:
                    torch.randn = orig
                rep[f"duration_predictor_T{T}"] = float((mine - theirs).abs().max())
    for k, v in rep.items():
        print(f"  HF cross-check {k}: max|Δ| = {v:.3e}")
        assert v < 2e-4, (k, v)
    return rep
