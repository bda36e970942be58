#!/usr/bin/env python3
"""Op-level ConvTranspose cases against the oracle in THIS process's environment — tests/test_gpu_ops.py runs it with
PIPER_HIP_PIPE_CT_MIN_GFLOP=0 so that conv_pipe_kernel's ConvTranspose path (off by default since the window kernel got
faster) stays checked. Prints one line per case: index, max |Δ|."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "piper-swift_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import katdata as kd  # noqa: E402
import oracle as orc  # noqa: E402  (the checker)
import piper_hip as ph  # noqa: E402

SD = 1234 + 600011  # as tests/test_gpu_ops.py
b = ph.HipBackend(0)
for idx in [int(a) for a in sys.argv[1:]] or range(len(kd.CONVT_WIN_CASES)):
    Cin, Cout, K, s, L, N = kd.CONVT_WIN_CASES[idx]
    pad = (K - s) // 2
    x = kd.sym(SD + 1300 + idx, (N, Cin, L))
    w = kd.weight(SD + 1350 + idx, (Cin, Cout, K), Cin * K // s)
    bias = kd.sym(SD + 1390 + idx, (Cout,), 0.1)
    out, shp = b.convTranspose1dF32(b.uploadFloat32(x), list(x.shape), b.uploadFloat32(w), list(w.shape), b.uploadFloat32(bias),
                                    stride=s, padL=pad, padR=pad)
    got = b.downloadFloat32(out).reshape(shp)
    ref = orc.convtranspose1d(x, w, bias, s, 1, pad, pad)
    assert list(ref.shape) == shp
    print(f"case {idx}: max_abs_err {float(np.max(np.abs(got - ref))):.3e} ref_max {float(np.max(np.abs(ref))):.3e}")
