#!/usr/bin/env python3
"""bench.py — the reference's scale bench (PiperCLI.swift:381-551 `--scale-bench`; ORT twin bench/benchmark_onnxruntime.py)
on the MI355X-native path.

A "step" is one utterance through the whole hot path (text encoder → flow → HiFi-GAN), inputs (ids, durations, noise)
already resident in HBM when the timed region starts, waveform copied back to the host inside the step.
N = 1 workload: BASELINE.json configs[1] — en_GB medium geometry, factor 8 (112 ids, 336 frames, 86 016 samples, 3.901 s
of 22 050 Hz audio), fp32, synthetic weights (seed 1234), durations pinned to 3 frames/id, injected noise.
N > 1: one process per GPU (torchrun), weights broadcast ONCE over RCCL/xGMI from rank 0, then each rank synthesises its
own utterances with no data-path collective (weak scaling).

Prints ONE JSON line (driver contract) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# CPU share of a 1-GPU box is 16 cores; the oracle's OpenMP team must not oversubscribe the host
_CORES = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
os.environ.setdefault("OMP_NUM_THREADS", str(_CORES))

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "piper-swift_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

FIXTURE_IDS = [1, 20, 0, 120, 0, 61, 0, 24, 0, 59, 0, 100, 0, 2]  # bench/fixtures/test_summary.json:8
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA (no sparsity)
HBM_PEAK_GBS = 8000.0
PROFILE_F32 = "r3_rocprof_summary.json"  # committed PMC passes of `python bench.py` (see profiles/README.md)
PROFILE_BF16 = "r3_high_bf16_rocprof_summary.json"


def utterance(factor, seed, inter=192):
    import katdata as kd
    ids = (FIXTURE_IDS * factor)[:4096]  # PiperCLI.swift:467-473
    dur = [3] * len(ids)
    noise = kd.sym(seed, (inter, 3 * len(ids)), 1.7320508)  # unit-variance, seeded
    return ids, dur, noise


def percentile(xs, p):  # PiperCLI.swift:425-436
    s = sorted(xs)
    k = (len(s) - 1) * (p / 100.0)
    f, c = int(np.floor(k)), int(np.ceil(k))
    return s[f] if f == c else s[f] + (s[c] - s[f]) * (k - f)


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without torchrun: this process never touches the GPU (no HIP call, no torch.cuda, not even
    an `import torch`); it starts N FRESH children — one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set — relays rank
    0's single JSON line and fails if any rank fails. (A process that has initialised the GPU must not exec or fork workers.)

    Every child is supervised, not only rank 0: the first rank to exit non-zero (import error, hipSetDevice failure, an RCCL
    error …) takes the others down at once — they would otherwise sit in the rendezvous or in a collective until a store /
    watchdog timeout — its stderr tail is printed and the launcher exits 1. An overall deadline (PIPER_BENCH_DEADLINE_S,
    default 900 s) bounds a hang that produces no exit code at all."""
    import subprocess
    import tempfile
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    deadline = time.monotonic() + float(os.environ.get("PIPER_BENCH_DEADLINE_S", "900"))
    procs, logs = [], []
    out0 = tempfile.TemporaryFile()
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        err = tempfile.TemporaryFile()
        logs.append(err)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL, stderr=err))

    def tail(f, nbytes=2000):
        f.seek(0, 2)
        size = f.tell()
        f.seek(max(0, size - nbytes))
        return f.read().decode("utf-8", "replace")

    def stop_all():
        for q in procs:
            if q.poll() is None:
                q.terminate()
        t_end = time.monotonic() + 5.0
        for q in procs:
            try:
                q.wait(timeout=max(0.1, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                q.kill()
                q.wait()

    failed = None
    while True:
        rcs = [q.poll() for q in procs]
        bad = [i for i, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            failed = (bad[0], rcs[bad[0]], "exited")
            break
        if all(rc == 0 for rc in rcs):
            break
        if time.monotonic() > deadline:
            failed = (next(i for i, rc in enumerate(rcs) if rc is None), None, "still running at the deadline")
            break
        time.sleep(0.05)
    if failed:
        stop_all()
        r, rc, what = failed
        sys.stderr.write(f"bench.py: rank {r} {what} (exit code {rc}); the other ranks were stopped. Its stderr tail:\n{tail(logs[r])}\n")
        raise SystemExit(1)
    for r, f in enumerate(logs):  # pass the ranks' diagnostics on (warnings, the one-rank notes)
        t = tail(f, 4000)
        if t.strip():
            sys.stderr.write(t if r == 0 else "".join(f"[rank {r}] {ln}\n" for ln in t.splitlines()))
    out0.seek(0)
    lines = [ln for ln in out0.read().decode("utf-8", "replace").splitlines() if ln.startswith("{")]
    if len(lines) != 1:
        sys.stderr.write(f"bench.py: all {n} ranks exited 0 but rank 0 printed {len(lines)} JSON line(s)\n")
        raise SystemExit(1)
    print(lines[0], flush=True)


def dry_run(args, rank, world, saved_stdout):
    """PIPER_BENCH_DRY=1: the multi-rank control flow on CPUs over gloo — rendezvous, one-shot blob broadcast, LPT shard of
    the batch-32 config, barrier-bracketed timed region, MAX over ranks, per-rank gather, ONE line from rank 0 — with a
    stand-in for the GPU work (no HIP, no kernels, `value` meaningless). Exists so tests/ can exercise `--gpus N`'s
    launcher and aggregation on a box without GPUs; never a measurement."""
    import torch
    import torch.distributed as dist
    import piper_hip as ph
    from piper_hip import distributed as phd
    if os.environ.get("PIPER_BENCH_DRY_FAIL_RANK") == str(rank):  # fault injection for tests/: this rank dies before the rendezvous
        sys.stderr.write(f"injected failure on rank {rank}\n")
        raise SystemExit(3)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = ph.voice_config(args.quality)
    n = ph.blob_floats(cfg)
    blob = torch.from_numpy(ph.synthetic_blob(cfg, 1234)) if rank == 0 else torch.zeros(n, dtype=torch.float32)
    dist.barrier()
    t0 = time.perf_counter()
    phd.broadcast_blob(blob, src=0)
    bcast_ms = (time.perf_counter() - t0) * 1e3
    digest = float(blob[::4099].double().sum())
    factors = phd.batch32_factors()
    mine = phd.shard_utterances([14 * f for f in factors], world)[rank]
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (rank + 1))  # stand-in: rank r "takes" r+1 ms per step
    dist.barrier()
    elapsed = phd.max_over_ranks(time.perf_counter() - t0)
    per_rank = [None] * world
    dist.all_gather_object(per_rank, {"rank": rank, "utterances": len(mine), "ids": int(sum(14 * factors[i] for i in mine)),
                                      "blob_digest": digest})
    if rank == 0:
        out = {"metric": "DRY RUN (gloo, no GPU work) — launcher / aggregation rehearsal only", "value": 0.0,
               "unit": "audio-sec/wall-sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(elapsed * 1e3 / args.steps, 4), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32", "data": "none", "dry_run": True, "world_size_reported": dist.get_world_size(),
               "weight_broadcast_ms": round(bcast_ms, 3), "batch32_per_rank": per_rank,
               "config": {"workload": "dry run", "parallelism": f"utterance-replicas x{world}"}}
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--factor", type=int, default=8)
    ap.add_argument("--quality", default="medium", choices=["medium", "high"])
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16"],
                    help="bf16 = BASELINE configs[4]: generator convs on bf16 operands with fp32 accumulate (use with --quality high)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-scale-bench", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--slots", type=int, default=8, help="utterances in flight for the batch-throughput side metric")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under torchrun: be the launcher (before anything in this process touches the GPU)
        return launch_ranks(args.gpus, sys.argv[1:])

    # stdout carries exactly ONE line (the JSON): anything libraries print (RCCL's version banner, …) goes to stderr
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torchrun --nproc-per-node {args.gpus}, "
                         f"or plain `python bench.py --gpus {args.gpus}` (it starts the ranks itself)")
    if os.environ.get("PIPER_BENCH_DRY") == "1":
        return dry_run(args, rank, world, saved_stdout)
    # PIPER_BENCH_FORCE_DIST=1 runs the multi-GPU code path (RCCL init, device-resident broadcast blob, MAX all-reduce)
    # with a single rank — the only way to rehearse it on a one-GPU box
    distributed = world > 1 or os.environ.get("PIPER_BENCH_FORCE_DIST") == "1"

    import piper_hip as ph
    if distributed:
        import torch  # noqa: F401  (first, so that piper_hip binds to the HIP runtime PyTorch/RCCL already use)
        ph.set_runtime("torch")
    cfg = ph.voice_config(args.quality)
    n_floats = ph.blob_floats(cfg)
    bcast_ms = None
    comm_info = None
    keep = []
    if distributed:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        # one-shot weight broadcast over RCCL/xGMI: rank 0 owns the blob, everyone else receives it in HBM
        wbuf = torch.empty(n_floats, dtype=torch.float32, device="cuda")
        if rank == 0:
            wbuf.copy_(torch.from_numpy(ph.synthetic_blob(cfg, 1234)))
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        from piper_hip import distributed as phd
        phd.broadcast_blob(wbuf, src=0)
        torch.cuda.synchronize()
        bcast_ms = (time.perf_counter() - t0) * 1e3
        backend = ph.HipBackend(local_rank)
        # the same broadcast once more through the library's own C-ABI (piper_hip_comm_* over rccl.h — what a host without
        # torch.distributed would call); its result must equal what torch.distributed delivered
        # Every rank must take the SAME path through the collectives below: a rank that skipped one (librccl not loadable by the
        # library, id creation failed on rank 0) while the others entered it would hang the job, not report an error. So the ranks
        # first agree — MIN over ranks of "RCCL is usable through the C-ABI here" (rank 0's flag includes creating the id) — and
        # the rehearsal runs on all ranks or on none.
        uid, why = None, ""
        try:
            usable = 1 if ph.comm_available() else 0
            if usable and rank == 0:
                uid = ph.comm_unique_id()
        except Exception as e:
            usable, why = 0, repr(e)
        flag = torch.tensor([usable], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            idt = torch.zeros(ph.COMM_ID_BYTES, dtype=torch.uint8, device="cuda")
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(uid), dtype=torch.uint8))
            dist.broadcast(idt, 0)
            # from here on an error is fatal for the whole job (the launcher stops the other ranks): the collectives are matched
            comm = ph.Comm(backend, idt.cpu().numpy().tobytes(), rank, world)
            w2 = wbuf.clone() if rank == 0 else torch.zeros_like(wbuf)
            torch.cuda.synchronize()
            comm.barrier()
            t0 = time.perf_counter()
            comm.broadcast_f32(w2.data_ptr(), n_floats, 0)
            c_ms = comm.max((time.perf_counter() - t0) * 1e3)
            comm_info = {"world": comm.world, "broadcast_ms": round(c_ms, 3), "equals_torch_broadcast": bool(torch.equal(w2, wbuf))}
            comm.close()
            del w2
        else:
            comm_info = {"skipped": "piper_hip_comm_* not usable on every rank" + (f" (this rank: {why})" if why else "")}
        rt = ph.HipRuntime(backend, cfg, wbuf.data_ptr(), on_device=True)
        keep.append(wbuf)
    else:
        backend = ph.HipBackend(0)
        rt = ph.HipRuntime(backend, cfg, ph.synthetic_blob(cfg, 1234))

    def barrier():
        if distributed:
            import torch
            import torch.distributed as dist
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    bf16 = args.precision == "bf16"
    if bf16:
        rt.set_precision("bf16")
    hop, sr = cfg.hop, cfg.sample_rate
    ids, dur, noise = utterance(args.factor, 1234 + rank, cfg.inter)
    n_samples = rt.num_samples(ids, dur)
    audio_sec = n_samples / sr
    # host-side cost of a request: cold = first prepare of this bucket (schedule build + graph capture +
    # instantiate), warm = the same bucket again (H2D of ids / durations / noise into the cached plan, nothing rebuilt)
    a = time.perf_counter()
    rt.prepare(0, ids, dur, noise, 0.667)
    prepare_cold_ms = (time.perf_counter() - a) * 1e3
    cold_breakdown = rt.last_build_breakdown()
    # the first request of a bucket runs its schedule eagerly (that run is the answer) and captures the graph behind it
    rt.launch(0)
    rt.collect(0)
    first_request_ms = (time.perf_counter() - a) * 1e3
    cold_breakdown.update({k: v for k, v in rt.last_build_breakdown().items() if k in ("capture", "instantiate")})
    warm = []
    for k in range(5):
        i2 = ids[:len(ids) - k] if k else ids  # k ≠ 0: another true length inside the same (T, F) bucket
        d2 = dur[:len(i2)]
        a = time.perf_counter()
        rt.prepare(0, i2, d2, noise[:, :sum(d2)], 0.667)
        warm.append((time.perf_counter() - a) * 1e3)
    plan0 = rt.plan_info(0)
    rt.prepare(0, ids, dur, noise, 0.667)

    out_buf = rt.pinned_empty(n_samples)  # page-locked destination (piper_hip_host_alloc): the D2H of the waveform is one DMA

    def step():
        rt.launch(0)
        return rt.collect(0, out=out_buf)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    gpu_ms = []
    for _ in range(args.steps):
        step()
        gpu_ms.append(rt.last_gpu_ms(0))
    barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        from piper_hip import distributed as phd
        elapsed = phd.max_over_ranks(elapsed, device="cuda")
    ms_per_step = elapsed * 1e3 / args.steps
    value = world * args.steps * audio_sec / elapsed  # whole-job audio-seconds per wall-second

    out = {
        "metric": "audio-sec/wall-sec (RTF^-1) with ms/utterance, Piper VITS en_GB-medium geometry, scale-bench factor",
        "value": round(value, 3),
        "unit": "audio-sec/wall-sec",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "ms_per_utterance": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16" if bf16 else "f32",
        "data": "synthetic (seeded weights of Piper-medium geometry, fixture phoneme ids tiled, pinned 3 frames/id, injected noise)",
        "config": {"workload": f"{args.quality} factor={args.factor}: {len(ids)} ids, {sum(dur)} frames, {n_samples} samples "
                               f"({audio_sec:.3f} s audio) per utterance, 1 utterance per step per GPU",
                   "factor": args.factor, "phoneme_count": len(ids), "frames": int(sum(dur)), "samples": int(n_samples),
                   "sample_rate": sr, "parallelism": f"utterance-replicas x{world} (one-shot RCCL weight broadcast)"},
        "gpu_ms_mean": round(float(np.mean(gpu_ms)), 4),
        "tuning_switches": ph.config_string(),  # PIPER_HIP_* A/B switches honoured in this process ("" = the shipped defaults)
        "prepare_ms_cold": round(prepare_cold_ms, 3),
        "prepare_ms_cold_breakdown": dict(cold_breakdown, note="first plan build of the process (it also creates the process's first HIP stream); capture + instantiate run "
                                          "AFTER the eager first launch has been enqueued, overlapping it"),
        "first_request_ms": round(first_request_ms, 3),
        "prepare_ms_warm": round(float(np.median(warm)), 4),
        "plan": {"bucket_ids": plan0["bucket_t"], "bucket_frames": plan0["bucket_f"]},
    }
    # end to end for one request on a cached plan: ids / durations / noise on the HOST → audio on the HOST
    e2e = []
    for _ in range(min(20, args.steps)):
        a = time.perf_counter()
        rt.prepare(0, ids, dur, noise, 0.667)
        rt.launch(0)
        rt.collect(0)
        e2e.append((time.perf_counter() - a) * 1e3)
    out["end_to_end_ms"] = round(float(np.median(e2e)), 4)
    if cfg.dp_present:  # the same with the frames per id PREDICTED on the device (duration predictor + one D2H of T int32)
        dpn = np.zeros((2, len(ids)), np.float32)
        rt.prepare(1, ids, None, None, 0.667, noise_mode="device", dp_noise=dpn)
        rt.launch(1); rt.collect(1)
        e2p = []
        for _ in range(min(20, args.steps)):
            a = time.perf_counter()
            rt.prepare(1, ids, None, None, 0.667, noise_mode="device", dp_noise=dpn)
            rt.launch(1)
            au = rt.collect(1)
            e2p.append((time.perf_counter() - a) * 1e3)
        out["end_to_end_predicted_durations"] = {"ms": round(float(np.median(e2p)), 4), "samples": int(au.size),
                                                 "note": "duration predictor + device RandomNormalLike; no host-supplied tensors but the ids"}
        # the same without the host round trip between the predictor and the flow: the caller bounds the frames (here 3 per id, the
        # plan the headline runs), generate_path runs on the device, the lengths come back with the waveform
        bound = 3 * len(ids)
        rt.prepare(1, ids, None, None, 0.667, noise_mode="device", dp_noise=dpn, max_frames=bound)
        rt.launch(1); rt.collect(1)
        e2b = []
        for _ in range(min(20, args.steps)):
            a = time.perf_counter()
            rt.prepare(1, ids, None, None, 0.667, noise_mode="device", dp_noise=dpn, max_frames=bound)
            rt.launch(1)
            ab = rt.collect(1)
            e2b.append((time.perf_counter() - a) * 1e3)
        out["end_to_end_predicted_durations"]["bounded"] = {"ms": round(float(np.median(e2b)), 4), "samples": int(ab.size), "max_frames": bound,
            "note": "piper_hip_voice_prepare_batch_bounded: one stream, no synchronisation before collect; plan = bucket(max_frames)"}
    # ---- a server's view: 200 requests of mixed length, one after the other on one slot id — ids, durations and noise
    # on the HOST → audio on the HOST. Lengths are seeded draws (14 … 400 ids, 1 … 5 frames per id), so requests land in many (T, F)
    # buckets: a bucket's first request builds its plan (schedule + arena, ≈ 0.5 ms) and runs eagerly, later ones replay the graph.
    if not args.no_scale_bench and rank == 0:
        import katdata as kd
        rs = np.random.RandomState(20240607)
        lat, buckets, built, detail = [], set(), 0, []
        for i in range(200):
            Tn = int(np.clip(np.exp(rs.normal(np.log(90.0), 0.6)), 14, 400))
            rid = [FIXTURE_IDS[j % 14] for j in range(Tn)]
            rdur = [int(x) for x in rs.randint(1, 6, size=Tn)]
            rnz = kd.sym(5000 + i, (cfg.inter, int(sum(rdur))), 1.7320508)
            a = time.perf_counter()
            rt.prepare(2, rid, rdur, rnz, 0.667)
            b = time.perf_counter()
            rt.launch(2)
            c = time.perf_counter()
            rt.collect(2)
            d = time.perf_counter()
            lat.append((d - a) * 1e3)
            pi = rt.plan_info(2)
            bk = (pi["bucket_t"], pi["bucket_f"])
            detail.append({"ms": round(lat[-1], 3), "request": i, "ids": Tn, "frames": int(sum(rdur)), "bucket": list(bk), "new_bucket": bk not in buckets,
                           "prepare_launch_collect_ms": [round((b - a) * 1e3, 3), round((c - b) * 1e3, 3), round((d - c) * 1e3, 3)],
                           "build": rt.last_build_breakdown() if bk not in buckets else None})
            buckets.add(bk)
        tail = lat[20:]
        out["request_stream"] = {"requests": len(lat), "distinct_buckets": len(buckets), "cached_plans": rt.plan_info(2)["cached_plans"],
                                 "ms_p50": round(percentile(lat, 50), 3), "ms_p95": round(percentile(lat, 95), 3), "ms_max": round(max(lat), 3),
                                 "ms_p50_after_20": round(percentile(tail, 50), 3), "ms_p95_after_20": round(percentile(tail, 95), 3),
                                 "ms_max_after_20": round(max(tail), 3), "ms_first": round(lat[0], 3),
                                 "slowest": sorted(detail, key=lambda d: -d["ms"])[:3],
                                 "note": "ids-to-audio per request incl. H2D of ids / durations / noise and D2H of the waveform; log-normal lengths (median 90 ids), seeded"}
    if bcast_ms is not None:
        import torch.distributed as dist
        out["weight_broadcast_ms"] = round(bcast_ms, 3)
        out["weight_blob_mb"] = round(n_floats * 4 / 1e6, 1)
        out["world_size_reported"] = dist.get_world_size()  # what RCCL's communicator says, not what --gpus asked for
        out["piper_hip_comm"] = comm_info

    # ---- BASELINE configs[3]: the 32 mixed-length utterances, LPT-sharded over the ranks (all 32 on one GPU when N = 1).
    # EVERY rank runs its shard inside the same barrier bracket; the batch time is the MAX over ranks.
    batch32 = None
    if not args.no_scale_bench:
        from piper_hip import distributed as phd
        factors = phd.batch32_factors()
        mine = phd.shard_utterances([14 * f for f in factors], world)[rank]
        by_factor = {}
        for i in mine:
            by_factor.setdefault(factors[i], []).append(i)
        groups = sorted(by_factor.items())
        for sl, (f, idxs) in enumerate(groups):  # one schedule per distinct shape, batch = utterances of that shape
            rt.prepare_batch(sl, [utterance(f, 3000 + 17 * f + k, cfg.inter) for k in range(len(idxs))], 0.667)

        def run_bucketed():
            for sl in range(len(groups)):
                rt.launch(sl)
            for sl in range(len(groups)):
                rt.collect(sl, want_audio=False)
        run_bucketed()
        reps = 5
        barrier()
        a = time.perf_counter()
        for _ in range(reps):
            run_bucketed()
        barrier()
        dt = (time.perf_counter() - a) / reps
        my_audio = sum(14 * factors[i] * 3 * hop / sr for i in mine)
        mine_row = {"rank": rank, "utterances": len(mine), "ids": int(sum(14 * factors[i] for i in mine)), "launches": len(groups),
                    "audio_sec": round(my_audio, 2), "ms_per_batch": round(dt * 1e3, 3)}
        rows = [mine_row]
        if distributed:
            dt = phd.max_over_ranks(dt, device="cuda")
            rows = phd.gather_rows(mine_row)
        tot_audio = sum(14 * f * 3 * hop / sr for f in factors)
        batch32 = {"utterances": 32, "audio_sec": round(tot_audio, 2), "ms_per_batch": round(dt * 1e3, 3),
                   "utterances_per_sec": round(32 / dt, 1), "audio_sec_per_wall_sec": round(tot_audio / dt, 1),
                   "per_rank": rows,
                   "note": "32 mixed-length utterances (factors 1..16), LPT-sharded over ranks, one prepare_batch schedule per shape "
                           "per rank, all ranks inside one barrier bracket, MAX over ranks"}
        # the same shard as ≤ 2 RAGGED launches per rank (longest half / shortest half; per-item lengths on the device)
        order = sorted(mine, key=lambda i: -factors[i])
        halves = [h for h in (order[:(len(order) + 1) // 2], order[(len(order) + 1) // 2:]) if h]
        for sl, idxs in enumerate(halves):
            rt.prepare_batch(8 + sl, [utterance(factors[i], 3000 + 17 * factors[i] + i, cfg.inter) for i in idxs], 0.667)

        def run_ragged():
            for sl in range(len(halves)):
                rt.launch(8 + sl)
            for sl in range(len(halves)):
                rt.collect(8 + sl, want_audio=False)
        run_ragged()
        barrier()
        a = time.perf_counter()
        for _ in range(reps):
            run_ragged()
        barrier()
        dt2 = (time.perf_counter() - a) / reps
        if distributed:
            dt2 = phd.max_over_ranks(dt2, device="cuda")
        batch32["ragged_2_launches"] = {"launches_per_rank": len(halves), "ms_per_batch": round(dt2 * 1e3, 3),
                                        "utterances_per_sec": round(32 / dt2, 1), "audio_sec_per_wall_sec": round(tot_audio / dt2, 1),
                                        "note": "each rank's shard as two ragged prepare_batch launches (bucket = longest item; "
                                                "shorter items are masked by their true lengths on the device)"}
        rt.prepare(0, ids, dur, noise, 0.667)  # slot 0 back to the headline utterance

    if rank == 0:
        # ---- roofline of the dominant kernel, measured live with HIP events on the slot's stream
        if not args.no_profile:
            stats = rt.profile(0, iters=10)
            floor = next((s["avg_us"] for s in stats if s["name"].startswith("(event floor")), 0.0)
            stats = [dict(s, raw_us=s["avg_us"], avg_us=max(s["avg_us"] - floor, 0.05)) for s in stats
                     if not s["name"].startswith("(event floor") and not s["name"].endswith((".fork", ".join"))]
            conv = [s for s in stats if s["flops"] > 0 and "rel_attention" not in s["name"] and s["name"] != "expand_noise"]
            mfma = [s for s in conv if "conv_post" not in s["name"]]  # conv_post is the HBM-bound small-Cout kernel
            tot_us = sum(s["avg_us"] for s in stats)
            # the dominant kernel timed the way it runs in production: its 87 launches replayed as their own HIP graph
            # between two events on the slot stream (kernel time + dispatch boundary, no per-kernel event overhead)
            avg_us, n_l, m_fl, m_by = rt.time_subset(0, "conv_bf16" if bf16 else "conv_mfma", iters=30)
            m_us = avg_us * n_l
            peak_tf = BF16_MFMA_PEAK_TFLOPS if bf16 else FP32_MFMA_PEAK_TFLOPS
            achieved = m_fl / (m_us * 1e-6) / 1e12 if m_us > 0 else 0.0
            traffic, traffic_commit = None, None
            try:  # HBM bytes per launch of these kernel families from the committed PMC passes of the same command
                def per_launch(path, families):
                    pj = json.load(open(os.path.join(ROOT, "profiles", path)))
                    fam = [pj["traffic"][f] for f in families if f in pj["traffic"]]
                    n = sum(f["launches"] for f in fam)
                    return (sum(f["launches"] * f["hbm_mb_per_launch"] for f in fam) / n * 1e6 if n else None), pj.get("commit")
                if args.factor == 8 and args.quality == "medium" and not bf16:
                    traffic, traffic_commit = per_launch(PROFILE_F32, ("conv_k1_ln_kernel", "conv_k3_ln_kernel", "conv_k3_r8_kernel", "conv_k1_kernel", "conv_gate_kernel", "conv_short_kernel", "conv_stream_kernel", "conv_win_kernel", "conv_pipe_kernel", "rb_pair_kernel"))
                elif args.factor == 8 and args.quality == "high" and bf16:
                    traffic, traffic_commit = per_launch(PROFILE_BF16, ("conv_bf16_kernel", "rb_pair_bf16_kernel"))
            except Exception:
                pass
            out["roofline"] = {
                "kernel": ("conv_bf16_kernel (bf16-operand MFMA Conv1d/ConvTranspose1d of the generator, LDS-resident input window; all of its launches in one utterance)"
                           if bf16 else
                           "fp32 MFMA Conv1d/ConvTranspose1d kernels (conv_k1 / conv_k1_ln / conv_k3_ln / conv_k3_r8 / conv_gate / conv_short / conv_stream kernels for short rows, conv_win_kernel / rb_pair_kernel for the generator long rows; all of their launches in one utterance)"),
                "bound": "mfma", "achieved": round(achieved, 3), "peak": peak_tf, "unit": "TFLOP/s",
                "frac": round(achieved / peak_tf, 4), "traffic": traffic,
                "traffic_profile_commit": traffic_commit,  # the tree the PMC passes were taken on (tools/collect_profiles.sh stamps it)
                "traffic_note": f"HBM bytes per launch from profiles/{PROFILE_BF16 if bf16 else PROFILE_F32}: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                "passes of this command, (2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md; algorithmic bytes per launch "
                                f"= {m_by / max(1, n_l) / 1e6:.2f} MB",
                "timing": "HIP events around a graph replay of only these launches (30 replays)",
                "launches": n_l, "avg_launch_us": round(avg_us, 3),
                "algorithmic_gflop_per_utterance": round(m_fl / 1e9, 3),
                "hbm_side": {"algorithmic_GBps": round(m_by / (m_us * 1e-6) / 1e9, 1) if m_us > 0 else 0.0, "peak": HBM_PEAK_GBS},
                "share_of_utterance_gpu_time": round(m_us / (float(np.mean(gpu_ms)) * 1e3), 3),
            }
            # the same measurement per kernel family (the family's launches replayed as one graph): the lumped figure above
            # is dominated by the ≈ 60 short-row launches of the encoder and the flow, which sit on the launch floor
            def fam_of(name):
                if "pair_x3" in name or "ab_lrelu_conv" in name:
                    return "rb_pair_kernel (generator ResBlock conv pairs)"
                if "convT" in name:
                    return "conv_win/conv_pipe (generator ConvTranspose)"
                if "rb" in name and name.startswith("dec."):
                    return "conv_win/conv_pipe (generator ResBlock convs, one conv per launch)"
                if name.startswith("dec."):
                    return None
                if ".in_gate" in name:
                    return "conv_gate_kernel (the flow's gated convs)"
                if name.startswith("enc"):
                    return "encoder convs: conv_k1 / conv_k1_ln / conv_k3_ln / conv_k3_r8 kernels (LayerNorm computed by its consumer)"
                if ".res_skip" in name or name.endswith((".pre", "post_sub")):
                    return "conv_k1_kernel (the flow's k = 1 convs)"
                return "conv_short / conv_stream kernels (the remaining short-row convs)"
            names = {}
            for st in conv:
                f = fam_of(st["name"])
                if f is not None and not bf16:
                    names.setdefault(f, []).append(st["name"])
            fams = {}
            for f, ns in names.items():  # ONE replayed graph per family ("a|b|c" = exactly these launches): a graph of a single
                us1, n1, fl1, _ = rt.time_subset(0, "|".join(ns) + "|", iters=20)  # short launch would time the replay, not the kernel
                fams[f] = {"launches": n1, "us": us1 * n1, "flops": fl1}
            out["roofline_by_kernel"] = [
                {"kernel": f, "launches": e["launches"], "avg_launch_us": round(e["us"] / max(1, e["launches"]), 2), "gflop": round(e["flops"] / 1e9, 3),
                 "achieved": round(e["flops"] / (e["us"] * 1e-6) / 1e12, 2) if e["us"] > 0 else 0.0, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                 "frac": round(e["flops"] / (e["us"] * 1e-6) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4) if e["us"] > 0 else 0.0}
                for f, e in sorted(fams.items(), key=lambda kv: -kv[1]["us"])]
            all_fl = sum(s["flops"] for s in stats)
            out["utterance_roofline"] = {
                "algorithmic_gflop": round(all_fl / 1e9, 3), "t_min_ms_at_fp32_mfma_peak": round(all_fl / (FP32_MFMA_PEAK_TFLOPS * 1e12) * 1e3, 4),
                "frac_of_peak_at_measured_latency": round(all_fl / (FP32_MFMA_PEAK_TFLOPS * 1e12) / (ms_per_step * 1e-3), 4),
                "n_launches": len(stats),
            }
            top = sorted(stats, key=lambda s: -s["avg_us"])[:8]
            out["top_launches_note"] = "per-launch HIP-event deltas minus the empty-kernel event floor (estimate; see profiles/ for rocprofv3)"
            out["top_launches"] = [{"name": s["name"], "us": round(s["avg_us"], 2),
                                    "tflops": round(s["flops"] / (s["avg_us"] * 1e-6) / 1e12, 2) if s["avg_us"] > 0 else 0} for s in top]
        # ---- the reference's scale bench: factor 1,2,4,8 latency (warmup 3, iters 20)
        if not args.no_scale_bench and world == 1:  # single-GPU side metric: other ranks would only wait
            rows = []
            for f in (1, 2, 4, 8):
                i2, d2, n2 = utterance(f, 99 + f, cfg.inter)
                rt.prepare(1, i2, d2, n2, 0.667)
                for _ in range(3):
                    rt.launch(1); rt.collect(1)
                wall, g = [], []
                for _ in range(20):
                    a = time.perf_counter()
                    rt.launch(1); rt.collect(1)
                    wall.append((time.perf_counter() - a) * 1e3)
                    g.append(rt.last_gpu_ms(1))
                asec = rt.num_samples(i2, d2) / sr
                rows.append({"factor": f, "phoneme_count": len(i2), "ms_mean": round(float(np.mean(wall)), 4),
                             "ms_p50": round(percentile(wall, 50), 4), "ms_p95": round(percentile(wall, 95), 4),
                             "ms_max": round(max(wall), 4), "audio_sec": round(asec, 3),
                             "rtf_inv": round(asec / (float(np.mean(wall)) * 1e-3), 1), "gpu_ms_mean": round(float(np.mean(g)), 4)})
            out["scale_bench"] = rows
            # side metric: several utterances in flight on independent slots (streams) of one GPU
            S = max(1, min(args.slots, 12))
            for s in range(S):
                i2, d2, n2 = utterance(args.factor, 500 + s, cfg.inter)
                rt.prepare(2 + s, i2, d2, n2, 0.667)
            for _ in range(2):
                for s in range(S):
                    rt.launch(2 + s)
                for s in range(S):
                    rt.collect(2 + s)
            reps = 10
            a = time.perf_counter()
            for _ in range(reps):
                for s in range(S):
                    rt.launch(2 + s)
                for s in range(S):
                    rt.collect(2 + s)
            dt = time.perf_counter() - a
            out["batch_throughput"] = {"slots_in_flight": S, "utterances_per_sec": round(reps * S / dt, 1),
                                       "audio_sec_per_wall_sec": round(reps * S * audio_sec / dt, 1)}
        # ---- same-shape batching: N utterances per launch share one schedule (batch dimension in every kernel)
        if not args.no_scale_bench and world == 1:  # single-GPU side metric: other ranks would only wait
            NBATCH = 8
            bu = [utterance(args.factor, 700 + b, cfg.inter) for b in range(NBATCH)]
            for sl in (14, 15):
                rt.prepare_batch(sl, bu, 0.667)
            for sl in (14, 15):
                rt.launch(sl)
            for sl in (14, 15):
                rt.collect(sl, want_audio=False)
            reps = 20
            a = time.perf_counter()
            for _ in range(reps):
                rt.launch(14); rt.launch(15)
                rt.collect(14, want_audio=False); rt.collect(15, want_audio=False)
            dt = time.perf_counter() - a
            us_b, n_b, fl_b, _ = rt.time_subset(14, "conv_bf16" if bf16 else "conv_mfma", iters=10)
            out["batched_same_shape"] = {"batch": NBATCH, "slots_in_flight": 2, "utterances_per_sec": round(reps * 2 * NBATCH / dt, 1),
                                         "audio_sec_per_wall_sec": round(reps * 2 * NBATCH * audio_sec / dt, 1),
                                         "conv_kernel_tflops": round(fl_b / (us_b * n_b * 1e-6) / 1e12, 2) if us_b > 0 else None}
        # ---- streaming (synthesizeStream): time to the first audio chunk vs the whole utterance, long-form input
        if not args.no_scale_bench and world == 1:  # single-GPU side metric: other ranks would only wait
            sf, chunk = 64, 64
            si, sd, sn = utterance(sf, 4242, cfg.inter)
            for _ in range(2):  # builds and caches the schedules / graphs of the three window widths
                for _c in rt.synthesize_stream(si, sd, sn, 0.667, chunkFrames=chunk, slot=13):
                    pass
            firsts, totals = [], []
            for _ in range(10):
                a = time.perf_counter()
                g = rt.synthesize_stream(si, sd, sn, 0.667, chunkFrames=chunk, slot=13)
                next(g)
                firsts.append(time.perf_counter() - a)
                for _c in g:
                    pass
                totals.append(time.perf_counter() - a)
            rt.prepare(13, si, sd, sn, 0.667)
            whole = []
            for _ in range(10):
                a = time.perf_counter()
                rt.launch(13)
                rt.collect(13)
                whole.append(time.perf_counter() - a)
            out["streaming"] = {"factor": sf, "frames": int(sum(sd)), "chunk_frames": chunk,
                                "chunk_audio_sec": round(chunk * hop / sr, 3),
                                "first_audio_ms": round(float(np.mean(firsts)) * 1e3, 3),
                                "all_chunks_ms": round(float(np.mean(totals)) * 1e3, 3),
                                "whole_utterance_ms": round(float(np.mean(whole)) * 1e3, 3),
                                "note": "encoder + flow once, generator per window of chunk + receptive-field halo"}
        # ---- CPU baseline: the oracle (C restatement, OpenMP) on the same workload, bounded sample
        if not args.no_cpu_baseline and world == 1:
            import oracle as orc
            blob = ph.synthetic_blob(cfg, 1234)
            cores = int(os.environ.get("OMP_NUM_THREADS", _CORES))
            orc.synthesize(cfg, blob, FIXTURE_IDS, [3] * 14, None, 0.667)  # warm the library
            reps = 1
            a = time.perf_counter()
            for _ in range(reps):
                orc.synthesize(cfg, blob, ids, dur, noise, 0.667)
            dt = (time.perf_counter() - a) / reps
            port = {"value": round(audio_sec / dt, 3), "unit": "audio-sec/wall-sec", "ms_per_utterance": round(dt * 1e3, 1),
                    "cores": cores, "kind": "port",
                    "sample": f"{reps} utterance of the same factor-{args.factor} workload (oracle/piper_oracle.c, OpenMP over output channels)"}
            # SURVEY.md §8d fallback (ii): PyTorch-CPU eager fp32 on the same synthetic graph (tests/torch_ref.py, library conv /
            # matmul kernels) — the closest stand-in for the ORT-CPU baseline the north star names (onnxruntime and a Piper
            # .onnx are not available offline). It is the meaningful CPU figure, so it is `cpu_baseline`; the naive port (the
            # parity oracle, written for clarity, not speed) stays beside it as `cpu_baseline_port`.
            cpu_model = ""
            try:
                cpu_model = next(ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name"))
            except Exception:
                pass
            port["cpu"] = cpu_model
            try:
                import torch
                import torch_ref
                torch.set_num_threads(cores)
                tref = torch_ref.Ref(cfg, blob)
                with torch.no_grad():
                    tref.synthesize(ids, dur, noise, 0.667)  # warm-up (thread pool, oneDNN primitive cache)
                    treps = 5
                    a = time.perf_counter()
                    for _ in range(treps):
                        tref.synthesize(ids, dur, noise, 0.667)
                    tdt = (time.perf_counter() - a) / treps
                out["cpu_baseline"] = {"value": round(audio_sec / tdt, 3), "unit": "audio-sec/wall-sec", "ms_per_utterance": round(tdt * 1e3, 2),
                                       "cores": cores, "kind": "torch-cpu-eager-fp32", "torch": torch.__version__, "cpu": cpu_model,
                                       "sample": f"{treps} utterances of the same factor-{args.factor} workload after 1 warm-up (tests/torch_ref.py)",
                                       "note": "stands in for the ORT-CPU provider the north star names (onnxruntime / a Piper .onnx are not available offline)"}
                out["cpu_baseline_port"] = port
            except Exception as e:  # torch missing on the box: the port is the baseline, and the line says why
                out["cpu_baseline"] = dict(port, torch_unavailable=repr(e))
        if batch32 is not None:
            out["batch32"] = batch32
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    rt.close()
    backend.close()
    if distributed:
        import torch.distributed as dist
        dist.barrier()  # rank 0 may still have been collecting its side metrics
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
