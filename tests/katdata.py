"""Deterministic, version-proof test data (SplitMix64 in numpy) + the known-answer case tables.

The same generator is implemented in C inside the library (`piper_hip_voice_synthetic_blob`,
piper-swift_amd/csrc/voice_blob.cpp) so that Python, the oracle and the GPU path all see identical
synthetic weights without shipping a file (SURVEY.md §8d "Synthetic inputs").
"""
import numpy as np

_G = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_TS = np.uint64(0xD1B54A32D192ED03)


def _mix(z):
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def u01(seed, n):
    """n floats in [0,1) with 24 random bits each: element j = mix(seed + (j+1)*G) >> 40."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = _mix(np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + idx * _G)
    return (z >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)


def sym(seed, shape, scale=1.0):
    """U(-scale, scale) float32 array: (2u-1) exact in fp32, one fp32 multiply by scale."""
    n = int(np.prod(shape)) if len(shape) else 1
    v = (np.float32(2.0) * u01(seed, n) - np.float32(1.0)) * np.float32(scale)
    return v.reshape(shape).astype(np.float32)


def tensor_seed(seed, index):
    with np.errstate(over="ignore"):
        return int(np.uint64(seed & 0xFFFFFFFFFFFFFFFF) ^ (np.uint64(index + 1) * _TS))


def weight(seed, shape, fan_in):
    return sym(seed, shape, np.float32(np.sqrt(3.0 / float(fan_in))))


# --------------------------------------------------------------------------------------------
# Op-level known-answer cases (SURVEY.md §8c "Op KATs"): every distinct hot-path geometry of §8a
# rows a3/a4/a5 (time axis shortened so the golden file stays small) plus the edge cases the
# domain has: L < K*dilation, asymmetric pads, stride 2, depthwise groups, no bias, empty output.
# --------------------------------------------------------------------------------------------
# (name, Cin, Cout, K, dilation, padL, padR, stride, groups, bias, L, N)
CONV1D_CASES = [
    ("enc_qkv_k1", 192, 192, 1, 1, 0, 0, 1, 1, True, 14, 1),
    ("enc_ffn1_k3", 192, 768, 3, 1, 1, 1, 1, 1, True, 14, 1),
    ("enc_ffn2_k3", 768, 192, 3, 1, 1, 1, 1, 1, True, 14, 1),
    ("enc_proj_k1", 192, 384, 1, 1, 0, 0, 1, 1, True, 14, 1),
    ("flow_pre_k1", 96, 192, 1, 1, 0, 0, 1, 1, True, 42, 1),
    ("flow_wn_in_k5", 192, 384, 5, 1, 2, 2, 1, 1, True, 42, 1),
    ("flow_rs_k1", 192, 384, 1, 1, 0, 0, 1, 1, True, 42, 1),
    ("flow_rs_last_k1", 192, 192, 1, 1, 0, 0, 1, 1, True, 42, 1),
    ("flow_post_k1", 192, 96, 1, 1, 0, 0, 1, 1, True, 42, 1),
    ("dec_pre_k7", 192, 256, 7, 1, 3, 3, 1, 1, True, 42, 1),
    ("dec_pre_high_k7", 192, 512, 7, 1, 3, 3, 1, 1, True, 10, 1),
    ("rb2_c128_k3_d1", 128, 128, 3, 1, 1, 1, 1, 1, True, 80, 1),
    ("rb2_c128_k3_d2", 128, 128, 3, 2, 2, 2, 1, 1, True, 80, 1),
    ("rb2_c128_k5_d2", 128, 128, 5, 2, 4, 4, 1, 1, True, 80, 1),
    ("rb2_c128_k5_d6", 128, 128, 5, 6, 12, 12, 1, 1, True, 80, 1),
    ("rb2_c128_k7_d3", 128, 128, 7, 3, 9, 9, 1, 1, True, 80, 1),
    ("rb2_c128_k7_d12", 128, 128, 7, 12, 36, 36, 1, 1, True, 80, 1),
    ("rb2_c64_k5_d6", 64, 64, 5, 6, 12, 12, 1, 1, True, 200, 1),
    ("rb2_c32_k7_d12", 32, 32, 7, 12, 36, 36, 1, 1, True, 300, 1),
    ("rb2_c32_k3_d1", 32, 32, 3, 1, 1, 1, 1, 1, True, 300, 1),
    ("rb1_c256_k11_d5", 256, 256, 11, 5, 25, 25, 1, 1, True, 40, 1),
    ("rb1_c64_k7_d3", 64, 64, 7, 3, 9, 9, 1, 1, True, 130, 1),
    ("dec_post_k7_nobias", 32, 1, 7, 1, 3, 3, 1, 1, False, 300, 1),
    ("edge_L_lt_Kd", 32, 32, 7, 12, 36, 36, 1, 1, True, 5, 1),
    ("edge_asym_pad", 16, 24, 3, 1, 2, 0, 1, 1, True, 17, 1),
    ("edge_stride2", 16, 32, 4, 1, 1, 1, 2, 1, True, 33, 1),
    ("edge_depthwise_d3", 192, 192, 3, 3, 3, 3, 1, 192, True, 30, 1),
    ("edge_groups2", 8, 12, 3, 1, 1, 1, 1, 2, True, 9, 1),
    ("edge_batch2", 32, 64, 3, 2, 2, 2, 1, 1, True, 50, 2),
    ("edge_odd_channels", 5, 7, 3, 1, 1, 1, 1, 1, True, 11, 1),
    ("edge_L1", 32, 32, 3, 1, 1, 1, 1, 1, True, 1, 1),
    ("edge_empty_out", 4, 4, 5, 1, 0, 0, 1, 1, True, 4, 1),
]

# (name, Cin, Cout, K, stride, padL, padR, outPad, dilation, groups, bias, L, N)
CONVT_CASES = [
    ("up0_medium", 256, 128, 16, 8, 4, 4, 0, 1, 1, True, 6, 1),
    ("up1_medium", 128, 64, 16, 8, 4, 4, 0, 1, 1, True, 12, 1),
    ("up2_medium", 64, 32, 8, 4, 2, 2, 0, 1, 1, True, 42, 1),
    ("up0_high", 512, 256, 16, 8, 4, 4, 0, 1, 1, True, 3, 1),
    ("up2_high", 128, 64, 4, 2, 1, 1, 0, 1, 1, True, 42, 1),
    ("up3_high", 64, 32, 4, 2, 1, 1, 0, 1, 1, True, 42, 1),
    ("edge_L1", 64, 32, 8, 4, 2, 2, 0, 1, 1, True, 1, 1),
    ("edge_outpad1", 16, 8, 4, 2, 1, 1, 1, 1, 1, True, 7, 1),
    ("edge_nobias_groups2", 8, 6, 4, 2, 1, 1, 0, 1, 2, False, 5, 1),
    ("edge_dil2", 8, 8, 3, 2, 1, 1, 0, 2, 1, True, 9, 1),
    ("edge_batch2", 32, 16, 8, 4, 2, 2, 0, 1, 1, True, 5, 2),
]

# (name, a_shape, b_shape): rel-attention GEMMs at T=14 and 40 (a5), lead-dim broadcast, path expansion.
MATMUL_CASES = [
    ("qk_T14", (1, 2, 14, 96), (1, 2, 96, 14)),
    ("relk_T14_bcast", (1, 2, 14, 96), (1, 1, 96, 27)),
    ("av_T14", (1, 2, 14, 14), (1, 2, 14, 96)),
    ("relv_T14_bcast", (1, 2, 14, 27), (1, 1, 27, 96)),
    ("qk_T40", (1, 2, 40, 96), (1, 2, 96, 40)),
    ("relk_T40_bcast", (1, 2, 40, 96), (1, 1, 96, 79)),
    ("path_expand", (1, 42, 14), (1, 14, 192)),
    ("rank2", (7, 5), (5, 3)),
    ("a_bcast", (1, 1, 3, 4), (2, 3, 4, 5)),
]

SOFTMAX_CASES = [("attn_T14", (1, 2, 14, 14)), ("attn_T112", (1, 2, 112, 112)), ("wide", (3, 1000)), ("col1", (5, 1))]

# (name, a_shape, b_shape)
BINARY_CASES = [
    ("same", (1, 192, 14), (1, 192, 14)),
    ("mask", (1, 192, 14), (1, 1, 14)),
    ("scalar", (1, 2, 14, 14), (1,)),
    ("gamma", (1, 14, 192), (192,)),
    ("rank_mix", (3, 1, 5), (2, 1, 4, 1)),
]


def case_seed(kind, idx):
    return 1234 + 7919 * idx + {"conv": 0, "convt": 100003, "mm": 200003, "sm": 300007, "bin": 400009, "un": 500009,
                                "mod": 600011, "cfg": 700001, "rng": 800011, "dp": 900001}[kind]


def conv1d_inputs(idx):
    name, cin, cout, k, d, pl, pr, s, g, bias, L, n = CONV1D_CASES[idx]
    sd = case_seed("conv", idx)
    x = sym(sd, (n, cin, L))
    w = weight(sd + 1, (cout, cin // g, k), (cin // g) * k)
    b = sym(sd + 2, (cout,), 0.1) if bias else None
    return x, w, b


def convt_inputs(idx):
    name, cin, cout, k, s, pl, pr, op, d, g, bias, L, n = CONVT_CASES[idx]
    sd = case_seed("convt", idx)
    x = sym(sd, (n, cin, L))
    w = weight(sd + 1, (cin, cout // g, k), max(1, (cin // g) * k // s))
    b = sym(sd + 2, (cout,), 0.1) if bias else None
    return x, w, b


def matmul_inputs(idx):
    name, sa, sb = MATMUL_CASES[idx]
    sd = case_seed("mm", idx)
    return sym(sd, sa), sym(sd + 1, sb)


def softmax_input(idx):
    name, s = SOFTMAX_CASES[idx]
    x = sym(case_seed("sm", idx), s, 8.0)
    flat = x.reshape(-1)
    if flat.size > 8:  # ±large values
        flat[3] = 80.0
        flat[5] = -80.0
    return x


def binary_inputs(idx):
    name, sa, sb = BINARY_CASES[idx]
    sd = case_seed("bin", idx)
    a = sym(sd, sa, 2.0)
    b = sym(sd + 1, sb, 2.0)
    return a, b


def unary_input():
    x = sym(case_seed("un", 0), (4099,), 6.0)
    x[:8] = np.array([0.0, -0.0, 1e-8, -1e-8, 30.0, -30.0, 88.0, -88.0], np.float32)
    return x


FIXTURE_IDS = [1, 20, 0, 120, 0, 61, 0, 24, 0, 59, 0, 100, 0, 2]  # bench/fixtures/test_summary.json:8


def bf16_round(a):
    """fp32 → nearest-even bf16 → fp32 (what v_cvt_pk_bf16_f32 does; no NaN/Inf in the test data)."""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).reshape(np.shape(a))


# bf16-operand conv cases: (Cin, Cout, K, dil, padL, padR, L, N, bias)
CONV_BF16_CASES = [
    (32, 32, 7, 1, 3, 3, 200, 1, True),       # last decoder stage / conv_post-like reach
    (64, 64, 3, 5, 5, 5, 1000, 2, True),      # ResBlock1 dilation 5, batch 2
    (192, 256, 7, 1, 3, 3, 42, 1, True),      # conv_pre medium at factor 1
    (256, 256, 11, 5, 25, 25, 300, 1, True),  # widest reach of the high voice
    (128, 128, 7, 12, 36, 36, 2688, 1, True), # widest reach of the medium voice, stage 0 at factor 8
    (32, 40, 3, 1, 0, 2, 77, 1, False),       # ragged: Cout not a multiple of 32, asymmetric pads, no bias
    (96, 64, 1, 1, 0, 0, 130, 3, True),       # k=1, 3 row groups of channels, batch 3
    (32, 32, 5, 2, 4, 4, 5, 1, True),         # row shorter than the reach
]
# (Cin, Cout, K, stride, L, N)
CONVT_BF16_CASES = [
    (256, 128, 16, 8, 42, 1),
    (64, 32, 8, 4, 500, 2),
    (128, 64, 4, 2, 129, 1),
    (512, 256, 16, 8, 20, 1),
]

# long-row fp32 cases routed to the window kernel (conv_win.hip): (Cin, Cout, K, dil, padL, padR, L, N, bias)
CONV_WIN_CASES = [
    (32, 32, 7, 1, 3, 3, 4096, 1, True),       # C=32: 4 waves side by side along the columns
    (64, 64, 3, 5, 5, 5, 2048, 2, True),       # 2×2 waves, batch 2
    (128, 128, 7, 12, 36, 36, 2688, 1, True),  # medium stage 0 at factor 8: K-split over the block's waves
    (128, 128, 3, 1, 1, 1, 1024, 1, False),    # K-split 4, no bias
    (32, 40, 5, 2, 3, 5, 1300, 1, True),       # ragged Cout, asymmetric pads, L % 4 == 0 but not % 32
    (256, 256, 11, 5, 25, 25, 1344, 1, True),  # high stage 0, widest window (Cin × reach)
    (34, 48, 3, 3, 3, 3, 1028, 1, True),       # odd channel-pair count (17): step padding inside a K range
    # conv_pipe.hip (persistent, chunk-pipelined; Cin % 32 == 0): blocks that walk several tiles, both ring phases, two column
    # tiles per wave, a row group with surplus waves, batch > 1
    (32, 32, 3, 1, 1, 1, 200000, 1, True),     # 1 563 tiles of 128 columns on ≤ 768 resident blocks: 2–3 tiles per block, odd tap count
    (32, 32, 5, 2, 4, 4, 262148, 1, True),     # NTW = 2 (256-column tiles), L % 256 ≠ 0
    (64, 64, 3, 2, 2, 2, 70000, 2, False),     # 2×2 waves × 2 column tiles, batch 2, 2 chunks per tile, no bias
    (64, 96, 3, 2, 2, 2, 3000, 1, True),       # 3 row tiles in row groups of 2: the last group's second wave is surplus
    (96, 32, 2, 1, 0, 1, 5000, 1, True),       # even tap count (ring phase never flips), 3 chunks, asymmetric pads
]
# (Cin, Cout, K, stride, L, N)
CONVT_WIN_CASES = [
    (128, 64, 16, 8, 2688, 1),
    (64, 32, 8, 4, 1024, 2),
    (256, 128, 16, 8, 336, 1),
    (128, 64, 4, 2, 260, 1),
    (64, 32, 8, 4, 60000, 1),   # conv_pipe: 4 row tiles (one per phase) × 1 875 column tiles: several tiles per block
    (32, 32, 4, 2, 9000, 2),    # one chunk per tile, 2 phases × 32 rows, batch 2
    (64, 96, 4, 2, 5000, 1),    # 6 row tiles (3 per phase) in row groups of 2: the middle group spans both phases
    (32, 64, 6, 3, 2052, 2),    # stride 3 (shift-free phase index), 6 row tiles, L % 32 != 0, batch 2
]
