/*
 * piper_hip_voice_layout.h — order and shapes of the tensors in a packed fp32 voice blob.
 *
 * The blob is the concatenation, in the order walked below, of the VITS initializers the hot path
 * reads, each dense row-major in its ONNX layout (Conv [Cout,Cin/g,K]; ConvTranspose [Cin,Cout/g,K];
 * TensorValue.swift:45-116 decodes them to the same flat float order).  A loader for real Piper
 * `.onnx` files only has to emit initializers in this order.
 *
 * Names follow the Piper/VITS initializer names the reference's tests mention
 * (`enc_p.encoder.attn_layers.0.conv_q.weight`, ONNXParsingTests.swift:32).
 *
 * Header-only, plain C, shared by the library (csrc/) and the test oracle (oracle/): it is part of the
 * boundary, not of either implementation.
 */
#ifndef PIPER_HIP_VOICE_LAYOUT_H
#define PIPER_HIP_VOICE_LAYOUT_H

#include <stdio.h>
#include <string.h>

#include "piper_hip.h"

typedef enum {
  PIPER_T_WEIGHT = 0, /* N(0, 1/sqrt(fan_in)) in the synthetic generator */
  PIPER_T_BIAS = 1,   /* N(0, 0.01) */
  PIPER_T_GAMMA = 2,  /* ones */
  PIPER_T_BETA = 3,   /* zeros */
  PIPER_T_EMB = 4     /* N(0, 1/sqrt(dim)) */
} piper_tensor_kind;

typedef struct {
  char name[96];
  piper_tensor_kind kind;
  int rank;
  long long shape[3];
  long long fan_in; /* for PIPER_T_WEIGHT / PIPER_T_EMB */
  size_t offset;    /* in floats from the blob start */
  size_t count;
} piper_tensor_desc;

typedef void (*piper_layout_visitor)(const piper_tensor_desc* d, void* user);

static inline void piper__emit(piper_layout_visitor fn, void* user, size_t* off, const char* name,
                               piper_tensor_kind kind, int rank, long long d0, long long d1, long long d2,
                               long long fan_in) {
  piper_tensor_desc d;
  memset(&d, 0, sizeof d);
  snprintf(d.name, sizeof d.name, "%s", name);
  d.kind = kind;
  d.rank = rank;
  d.shape[0] = d0;
  d.shape[1] = rank > 1 ? d1 : 1;
  d.shape[2] = rank > 2 ? d2 : 1;
  d.fan_in = fan_in;
  d.offset = *off;
  d.count = (size_t)(d.shape[0] * d.shape[1] * d.shape[2]);
  *off += d.count;
  if (fn) fn(&d, user);
}

static inline void piper__conv(piper_layout_visitor fn, void* user, size_t* off, const char* prefix, long long cout,
                               long long cin, long long k, int has_bias) {
  char nm[96];
  snprintf(nm, sizeof nm, "%s.weight", prefix);
  piper__emit(fn, user, off, nm, PIPER_T_WEIGHT, 3, cout, cin, k, cin * k);
  if (has_bias) {
    snprintf(nm, sizeof nm, "%s.bias", prefix);
    piper__emit(fn, user, off, nm, PIPER_T_BIAS, 1, cout, 1, 1, 0);
  }
}

/* Walks every tensor of the blob in order; returns the total float count. */
static inline size_t piper_hip_layout_walk(const piper_hip_voice_config* c, piper_layout_visitor fn, void* user) {
  size_t off = 0;
  char p[96];
  const long long H = c->hidden, D = c->hidden / c->n_heads, I = c->inter, half = c->inter / 2;
  piper__emit(fn, user, &off, "enc_p.emb.weight", PIPER_T_EMB, 2, c->n_vocab, H, 1, H);
  for (int l = 0; l < c->n_layers; l++) {
    static const char* qkvo[4] = {"conv_q", "conv_k", "conv_v", "conv_o"};
    for (int j = 0; j < 4; j++) {
      snprintf(p, sizeof p, "enc_p.encoder.attn_layers.%d.%s", l, qkvo[j]);
      piper__conv(fn, user, &off, p, H, H, 1, 1);
    }
    snprintf(p, sizeof p, "enc_p.encoder.attn_layers.%d.emb_rel_k", l);
    piper__emit(fn, user, &off, p, PIPER_T_EMB, 2, 2 * c->window + 1, D, 1, D);
    snprintf(p, sizeof p, "enc_p.encoder.attn_layers.%d.emb_rel_v", l);
    piper__emit(fn, user, &off, p, PIPER_T_EMB, 2, 2 * c->window + 1, D, 1, D);
    snprintf(p, sizeof p, "enc_p.encoder.norm_layers_1.%d.gamma", l);
    piper__emit(fn, user, &off, p, PIPER_T_GAMMA, 1, H, 1, 1, 0);
    snprintf(p, sizeof p, "enc_p.encoder.norm_layers_1.%d.beta", l);
    piper__emit(fn, user, &off, p, PIPER_T_BETA, 1, H, 1, 1, 0);
    snprintf(p, sizeof p, "enc_p.encoder.ffn_layers.%d.conv_1", l);
    piper__conv(fn, user, &off, p, c->ffn, H, c->ffn_kernel, 1);
    snprintf(p, sizeof p, "enc_p.encoder.ffn_layers.%d.conv_2", l);
    piper__conv(fn, user, &off, p, H, c->ffn, c->ffn_kernel, 1);
    snprintf(p, sizeof p, "enc_p.encoder.norm_layers_2.%d.gamma", l);
    piper__emit(fn, user, &off, p, PIPER_T_GAMMA, 1, H, 1, 1, 0);
    snprintf(p, sizeof p, "enc_p.encoder.norm_layers_2.%d.beta", l);
    piper__emit(fn, user, &off, p, PIPER_T_BETA, 1, H, 1, 1, 0);
  }
  piper__conv(fn, user, &off, "enc_p.proj", 2 * I, H, 1, 1);
  /* flow: couplings at module indices 0,2,4,.. (Flip modules in between hold no weights) */
  for (int f = 0; f < c->n_flows; f++) {
    snprintf(p, sizeof p, "flow.flows.%d.pre", 2 * f);
    piper__conv(fn, user, &off, p, H, half, 1, 1);
    for (int i = 0; i < c->wn_layers; i++) {
      snprintf(p, sizeof p, "flow.flows.%d.enc.in_layers.%d", 2 * f, i);
      piper__conv(fn, user, &off, p, 2 * H, H, c->wn_kernel, 1);
      snprintf(p, sizeof p, "flow.flows.%d.enc.res_skip_layers.%d", 2 * f, i);
      piper__conv(fn, user, &off, p, (i + 1 < c->wn_layers) ? 2 * H : H, H, 1, 1);
    }
    snprintf(p, sizeof p, "flow.flows.%d.post", 2 * f);
    piper__conv(fn, user, &off, p, half, H, 1, 1); /* mean_only */
  }
  piper__conv(fn, user, &off, "dec.conv_pre", c->up_initial, I, 7, 1);
  long long ch = c->up_initial;
  for (int u = 0; u < c->n_ups; u++) {
    char nm[96];
    snprintf(nm, sizeof nm, "dec.ups.%d.weight", u);
    /* ConvTranspose layout [Cin, Cout, K]; fan_in per output sample = Cin*K/stride */
    piper__emit(fn, user, &off, nm, PIPER_T_WEIGHT, 3, ch, ch / 2, c->up_kernels[u],
                ch * c->up_kernels[u] / c->up_rates[u]);
    snprintf(nm, sizeof nm, "dec.ups.%d.bias", u);
    piper__emit(fn, user, &off, nm, PIPER_T_BIAS, 1, ch / 2, 1, 1, 0);
    ch /= 2;
  }
  ch = c->up_initial;
  for (int u = 0; u < c->n_ups; u++) {
    ch /= 2;
    for (int j = 0; j < c->n_rb; j++) {
      const int rb = u * c->n_rb + j;
      for (int d = 0; d < c->rb_n_dil; d++) {
        if (c->resblock_type == 1) {
          snprintf(p, sizeof p, "dec.resblocks.%d.convs1.%d", rb, d);
          piper__conv(fn, user, &off, p, ch, ch, c->rb_kernels[j], 1);
          snprintf(p, sizeof p, "dec.resblocks.%d.convs2.%d", rb, d);
          piper__conv(fn, user, &off, p, ch, ch, c->rb_kernels[j], 1);
        } else {
          snprintf(p, sizeof p, "dec.resblocks.%d.convs.%d", rb, d);
          piper__conv(fn, user, &off, p, ch, ch, c->rb_kernels[j], 1);
        }
      }
    }
  }
  piper__conv(fn, user, &off, "dec.conv_post", 1, ch, 7, 0);
  /* stochastic duration predictor, appended (earlier offsets do not move). Module indices as in VITS: dp.flows = [ElementwiseAffine,
   * ConvFlow, Flip, ConvFlow, Flip, …]; inference (reverse) uses flows 2·n_flows − 1, …, 3 and flow 0 — flow 1 is the one
   * `flows[:-2] + [flows[-1]]` drops, so an export holds no initializers for it. */
  if (c->dp_present) {
    const long long Kd = c->dp_kernel;
    piper__conv(fn, user, &off, "dp.pre", H, H, 1, 1);
    piper__conv(fn, user, &off, "dp.proj", H, H, 1, 1);
    for (int blk = 0; blk <= c->dp_n_flows - 1; blk++) { /* blk 0: the predictor's own DDSConv; blk ≥ 1: ConvFlow 2·blk + 1 */
      char base[64];
      if (blk == 0) snprintf(base, sizeof base, "dp");
      else snprintf(base, sizeof base, "dp.flows.%d", 2 * blk + 1);
      if (blk > 0) {
        snprintf(p, sizeof p, "%s.pre", base);
        piper__conv(fn, user, &off, p, H, 1, 1, 1);
      }
      for (int i = 0; i < c->dp_dds_layers; i++) {
        char nm[96];
        snprintf(nm, sizeof nm, "%s.convs.convs_sep.%d.weight", base, i); /* depthwise: [H, 1, K] */
        piper__emit(fn, user, &off, nm, PIPER_T_WEIGHT, 3, H, 1, Kd, Kd);
        snprintf(nm, sizeof nm, "%s.convs.convs_sep.%d.bias", base, i);
        piper__emit(fn, user, &off, nm, PIPER_T_BIAS, 1, H, 1, 1, 0);
        snprintf(p, sizeof p, "%s.convs.convs_1x1.%d", base, i);
        piper__conv(fn, user, &off, p, H, H, 1, 1);
        for (int j = 1; j <= 2; j++) {
          snprintf(nm, sizeof nm, "%s.convs.norms_%d.%d.gamma", base, j, i);
          piper__emit(fn, user, &off, nm, PIPER_T_GAMMA, 1, H, 1, 1, 0);
          snprintf(nm, sizeof nm, "%s.convs.norms_%d.%d.beta", base, j, i);
          piper__emit(fn, user, &off, nm, PIPER_T_BETA, 1, H, 1, 1, 0);
        }
      }
      if (blk > 0) {
        snprintf(p, sizeof p, "%s.proj", base);
        piper__conv(fn, user, &off, p, 3 * c->dp_bins - 1, H, 1, 1);
      }
    }
    piper__emit(fn, user, &off, "dp.flows.0.m", PIPER_T_BETA, 2, 2, 1, 1, 0);
    piper__emit(fn, user, &off, "dp.flows.0.logs", PIPER_T_BETA, 2, 2, 1, 1, 0);
  }
  return off;
}

#endif /* PIPER_HIP_VOICE_LAYOUT_H */
