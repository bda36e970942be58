"""CPU: the PyTorch restatement behind tests/golden/ (tests/torch_ref.py) against the third-party VITS implementation in
`transformers.models.vits` (SURVEY.md §8c "independent oracle (2)"), and the C oracle against the torch restatement on a
whole utterance — so an error in the ASSUMED Piper graph structure (op order, skew, coupling direction, MRF mean) would
have to be shared by three independently written implementations to go unnoticed."""
import os

import numpy as np
import pytest

import katdata as kd
import oracle as orc

torch = pytest.importorskip("torch")
pytest.importorskip("transformers")


def test_torch_ref_matches_hf_vits(voices):
    os.environ["HF_HUB_OFFLINE"] = "1"
    import torch_ref
    torch.set_num_threads(4)
    (cfg_m, blob_m), (cfg_h, blob_h) = voices["medium"], voices["high"]
    rep = torch_ref.hf_crosscheck(cfg_m, blob_m, cfg_h, blob_h)  # asserts every entry < 2e-4
    assert set(rep) >= {"attention_T3", "attention_T14", "attention_T40", "encoder_layer_T14", "flow_reverse", "hifigan_high"}
    assert max(rep[k] for k in rep if k.startswith("attention")) < 5e-6


def test_oracle_matches_torch_ref_whole_utterance(voices):
    """C oracle (loops in the reference's accumulation order) vs torch (library kernels) on an utterance that is NOT in
    tests/golden: ragged durations, 20 ids."""
    import torch_ref
    torch.set_num_threads(4)
    cfg, blob = voices["medium"]
    rng = np.random.RandomState(3)
    ids = rng.randint(0, 130, size=20).tolist()
    dur = rng.randint(0, 5, size=20).tolist()
    dur[0] = 2
    F = int(np.sum(dur))
    noise = kd.sym(4711, (cfg.inter, F), 1.7320508)
    with torch.no_grad():
        r = torch_ref.Ref(cfg, blob).synthesize(ids, dur, noise, 0.667)
    audio, taps = orc.synthesize(cfg, blob, ids, dur, noise, 0.667, taps=True)
    for k in ("enc_out", "z_p", "z"):
        np.testing.assert_allclose(taps[k].reshape(-1), r[k].numpy().reshape(-1), atol=1e-4, rtol=0)
    np.testing.assert_allclose(audio, r["audio"].numpy(), atol=1e-3, rtol=0)
