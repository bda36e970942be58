"""CPU, world_size 2 over gloo: the N>1 path's host logic — one-shot weight broadcast and utterance sharding."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import piper_hip as ph
from piper_hip import distributed as phd


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = ph.voice_config("medium")
        n = ph.blob_floats(cfg)
        if rank == 0:
            blob = torch.from_numpy(ph.synthetic_blob(cfg, 1234))
        else:
            blob = torch.zeros(n, dtype=torch.float32)
        phd.broadcast_blob(blob, src=0)
        # every rank ends up with rank 0's bytes
        digest = float(blob.double().sum()) + float(blob[::4097].double().abs().sum())
        # disjoint shards of the batch-32 config, no data-path collective
        factors = phd.batch32_factors()
        shards = phd.shard_utterances([14 * f for f in factors], world)
        mine = shards[rank]
        audio_sec = sum(14 * factors[i] * 3 * 256 / 22050.0 for i in mine)
        total = phd.sum_over_ranks(audio_sec)
        slowest = phd.max_over_ranks(0.001 * (rank + 1))
        ret[rank] = dict(digest=digest, mine=mine, total=total, slowest=slowest)
    finally:
        dist.destroy_process_group()


def test_broadcast_and_sharding_gloo():
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
        r0, r1 = ret[0], ret[1]
    assert r0["digest"] == r1["digest"] != 0.0
    assert sorted(r0["mine"] + r1["mine"]) == list(range(32))
    assert not set(r0["mine"]) & set(r1["mine"])
    assert abs(r0["total"] - 101.4) < 0.1 and r0["total"] == r1["total"]  # 101.4 s of audio in the batch
    assert r0["slowest"] == r1["slowest"] == 0.002


def test_lpt_partition_balance():
    factors = phd.batch32_factors()
    assert sorted(factors) == sorted([1, 2, 3, 4, 6, 8, 12, 16] * 4) and factors != sorted(factors)
    assert sum(14 * f for f in factors) == 2912  # ids in the batch
    for world in (1, 2, 4, 8):
        shards = phd.shard_utterances([14 * f for f in factors], world)
        loads = [sum(14 * factors[i] for i in sh) for sh in shards]
        assert sum(loads) == 2912 and max(loads) - min(loads) <= 14 * 16
        assert sorted(i for sh in shards for i in sh) == list(range(32))
    # 8 GPUs: every rank gets 364 ids (one utterance of each factor group)
    assert set(sum(14 * factors[i] for i in sh) for sh in phd.shard_utterances([14 * f for f in factors], 8)) == {364}


def test_bench_launcher_spawns_ranks_dry():
    """`python bench.py --gpus 2` WITHOUT torchrun must start two ranks itself and print ONE line with n_gpus = 2
    (round-1 defect: it silently measured one GPU). PIPER_BENCH_DRY=1 swaps the GPU work for a stand-in and RCCL for
    gloo, so the launcher, the rendezvous, the broadcast, the sharding and the aggregation run here without a GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PIPER_BENCH_DRY="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["world_size_reported"] == 2 and out["dry_run"] is True
    assert out["ms_per_step"] >= 2.0  # MAX over ranks: rank 1's stand-in takes 2 ms per step
    rows = out["batch32_per_rank"]
    assert [r_["rank"] for r_ in rows] == [0, 1] and sum(r_["utterances"] for r_ in rows) == 32
    assert rows[0]["blob_digest"] == rows[1]["blob_digest"] != 0.0  # rank 1 received rank 0's weights
    # a mismatching WORLD_SIZE is an error, not a silent single-GPU run
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], env=dict(env, WORLD_SIZE="1", RANK="0"),
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in bad.stderr


def test_bench_launcher_stops_everyone_when_one_rank_dies():
    """a rank other than 0 that dies before the rendezvous used to leave rank 0 waiting in it until a store
    timeout. The launcher supervises every child: the first non-zero exit stops the siblings and the launcher returns non-zero
    within seconds, naming the rank and showing its stderr."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PIPER_BENCH_DRY="1", PIPER_BENCH_DRY_FAIL_RANK="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=120)
    took = time.monotonic() - t0
    assert r.returncode != 0
    assert "rank 1 exited" in r.stderr and "injected failure on rank 1" in r.stderr, r.stderr[-2000:]
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]  # no half-baked result line
    assert took < 60, f"launcher needed {took:.0f} s to notice a dead rank"
    # the overall deadline bounds a hang that never produces an exit code
    env2 = dict(env, PIPER_BENCH_DEADLINE_S="0.5")
    env2.pop("PIPER_BENCH_DRY_FAIL_RANK")
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                        env=env2, capture_output=True, text=True, timeout=120)
    assert r2.returncode != 0 and "deadline" in r2.stderr
