#!/usr/bin/env python3
"""bench.py -- learner transitions/s of the V-RACER / ReF-ER update on MI355X.

Workload (BASELINE.json `metric`, configs[1] shape): synthetic replay of 1 000 000
transitions (5000 episodes x 201 states, state_dim 17, act_dim 6 all bounded), VRACER, 2x256
SoftSign MLP (72 976 padded fp32 parameters), batch 256, clipImpWeight 4, gamma .995, lambda 1.
A "step" = one gradient step = device-side sampling + minibatch gather + MLP forward/backward +
V-RACER head + ReF-ER bookkeeping + (all-reduce) + Adam, including the 1000-step whole-buffer
Retrace / statistics sweeps that fall into the timed region.  The replay is resident in HBM
before the timed region starts.

N > 1 (launched by torch.distributed.run, one rank per GPU): the reference's multi-learner mode
(Settings/HyperParameters.cpp:186-197): batch 256 and the 1M replay are SPLIT over the replicas
(strong scaling), one fp32 gradient all-reduce (RCCL over xGMI) per step.

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events on the library's
stream around every launch of the dominant kernel (second, eager pass); `cpu_baseline` times the
compiled reference (oracle/_ref, kind "reference") -- or the single-threaded CPU oracle
(kind "port") when the reference binary is absent -- on a bounded sample of the same workload.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec peak

N_EPISODES, EP_STATES = 5000, 201
CFG = dict(dimS=17, dimA=6, hidden=(256, 256), nnFunc="SoftSign", batchSize=256, maxTotObsNum=1000000,
           clipImpWeight=4.0, penalTol=0.1, epsAnneal=0.0, gamma=0.995, lambda_=1.0, learnrate=1e-4,
           explNoise=0.4472135955, outWeightsPrefac=0.1, nnLambda=0.0, randSeed=42)


def kernel_work(name, B, dS, dA, hidden, nParams):
    """Algorithmic FLOPs / bytes of one launch of the named kernel (kernel table)."""
    dims = [dS] + list(hidden)
    nDense = 1 + dA
    if name.startswith("gemm16_fwd"):
        j = int(name[len("gemm16_fwd"):])
        return "mfma", 2.0 * B * dims[j] * dims[j + 1]
    if name.startswith("gemm16_dx"):
        j = int(name[len("gemm16_dx"):])
        return "mfma", 2.0 * B * dims[j + 1] * dims[j]
    if name == "gemm16_dw":
        fl = sum(2.0 * B * (dims[j] + 1) * dims[j + 1] for j in range(len(hidden)))
        fl += 2.0 * B * (dims[-1] + 1) * nDense
        return "mfma", fl
    if name == "adam_kernel":
        return "hbm", 7.0 * nParams * 4        # read W,M1,M2,G ; write W,M1,M2
    if name == "head_kernel":
        # per sample: read Y[H] + W_out[H*8] (L2 resident) + a,mu (f64) + write deltas 2x H
        H = dims[-1]
        return "hbm", B * (H * 4 + 3 * dA * 8 + 2 * H * 4 + 13 * 8 * 2)
    if name == "sample_kernel":
        return "hbm", B * (2 * dS * 4 + 8 * 6)
    return "hbm", 0.0


def cpu_baseline(steps_budget_s=20.0):
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_driver_fast")
    ncpu = os.cpu_count() or 1
    if os.path.exists(ref):
        best = None
        tried = []
        with tempfile.TemporaryDirectory() as td:
            for thr in [t for t in (8, 16, 32, 64) if t <= ncpu] or [ncpu]:
                env = dict(os.environ, OMP_NUM_THREADS=str(thr), OMP_PROC_BIND="close", OMP_PLACES="cores")
                try:
                    out = subprocess.run([ref, "bench", "threads=%d" % thr, "nObs=1000000", "nSteps=400", "warmup=20"],
                                         cwd=td, env=env, capture_output=True, text=True, timeout=180).stdout
                    line = [l for l in out.splitlines() if l.startswith("{\"kind\"")][-1]
                    r = json.loads(line)
                    tried.append((thr, r["transitions_per_s"]))
                    if best is None or r["transitions_per_s"] > best["transitions_per_s"]:
                        best = r
                except Exception as e:  # noqa: BLE001
                    tried.append((thr, "failed: %s" % e))
        if best is not None:
            return {"value": best["transitions_per_s"], "unit": "transitions/s", "cores": int(best["threads"]),
                    "kind": "reference",
                    "sample": "compiled reference (oracle/_ref, -O3 -ffast-math, OpenMP), same 1M-transition "
                              "synthetic replay, 400 gradient steps after 20 warm-up; best of threads=%s on %d host CPUs"
                              % ([t for t, _ in tried], ncpu),
                    "tried": tried}
    # fallback: single-threaded CPU oracle (port)
    from oracle_api import oracle_learner, fill_synth, synth_cfg
    from smarties_amd import capi
    L = oracle_learner(capi.make_config(**CFG))
    L.init_weights()
    fill_synth(L, synth_cfg(seed=7, dimS=17, dimA=6, lenMin=EP_STATES, lenMax=EP_STATES, pTerm=0.0), N_EPISODES)
    L.initialize()
    L.step(5)
    t0 = time.time(); n = 0
    while time.time() - t0 < 10.0:
        L.step(10); n += 10
    dt = time.time() - t0
    return {"value": 256.0 * n / dt, "unit": "transitions/s", "cores": 1, "kind": "port",
            "sample": "CPU oracle (oracle/port, single thread), same 1M-transition replay, %d steps" % n}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--roofline-steps", type=int, default=300)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    n_ranks = world

    import numpy as np
    import torch
    import torch.distributed as dist
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    if n_ranks > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=n_ranks)

    from smarties_amd import capi, load_hip
    from oracle_api import synth_cfg, synth_episode   # synthetic replay generator only (oracle/synth.h)

    api = load_hip()
    cfg = capi.make_config(n_ranks=n_ranks, rank=rank, device_id=local_rank, **CFG)
    L = capi.Learner(api, cfg)
    L.init_weights()
    sc = synth_cfg(seed=7, dimS=17, dimA=6, lenMin=EP_STATES, lenMax=EP_STATES, pTerm=0.0)
    per = N_EPISODES // n_ranks
    t_fill = time.time()
    for e in range(rank * per, (rank + 1) * per):
        L.append_episode(**synth_episode(sc, e))
    t_fill = time.time() - t_fill
    if n_ranks > 1:
        idbuf = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            import ctypes as C
            raw = (C.c_uint8 * 128)()
            assert api.fn("comm_unique_id")(raw) == 0
            idbuf = torch.tensor(list(raw), dtype=torch.uint8, device="cuda")
        dist.broadcast(idbuf, 0)
        L.comm_init(bytes(idbuf.cpu().tolist()))
    L.initialize()

    def barrier():
        if n_ranks > 1:
            dist.barrier()
        torch.cuda.synchronize()
        L.sync()

    L.step(args.warmup)
    barrier()
    t0 = time.perf_counter()
    L.step(args.steps)
    L.sync()
    barrier()
    dt = time.perf_counter() - t0
    if n_ranks > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    B_global = CFG["batchSize"]
    value = B_global * args.steps / dt

    # ---- roofline of the dominant kernel: HIP events around every launch, eager pass ------------------
    roof = None
    if rank == 0:
        L.timing_enable(True)
        L.step(args.roofline_steps)
        L.sync()
        names = ["step_tail_kernel"] + ["gemm16_fwd%d" % j for j in range(len(CFG["hidden"]))] + ["head_kernel"] + \
                ["gemm16_dx%d" % j for j in range(1, len(CFG["hidden"]))] + ["gemm16_dw", "adam_kernel", "post_kernel"]
        times = {n: L.timing_get(n) for n in names}
        L.timing_enable(False)
        tot = {n: ms * cnt for n, (ms, cnt) in times.items()}
        dom = max(tot, key=tot.get)
        ms, cnt = times[dom]
        kind, work = kernel_work(dom, L.B, 17, 6, CFG["hidden"], L.nParams)
        if kind == "mfma":
            achieved = work / (ms * 1e-3) / 1e12
            roof = {"kernel": dom, "bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS,
                    "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": None}
        else:
            achieved = work / (ms * 1e-3) / 1e9
            roof = {"kernel": dom, "bound": "hbm", "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": achieved / PEAK_HBM_GBS, "traffic": None}
        roof["avg_launch_us"] = ms * 1e3
        roof["launches"] = cnt
        roof["per_kernel_avg_us"] = {n: round(v[0] * 1e3, 3) for n, v in times.items()}

    out = None
    if rank == 0:
        out = {
            "metric": "learner transitions/sec, VRACER batch=256 over 1M replay",
            "value": value, "unit": "transitions/s", "n_gpus": n_ranks, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "VRACER learner update, synthetic replay 5000 episodes x 200 transitions = 1M "
                                   "(state_dim 17, act_dim 6 bounded), 2x256 SoftSign MLP (72976 padded fp32 params), "
                                   "global batch 256 split over %d replica(s), replay split likewise, "
                                   "device-side mt19937 sampler" % n_ranks,
                       "global_batch": B_global, "replay_transitions": 1000000, "parallelism": "dp%d" % n_ranks},
            "roofline": roof,
            "fill_seconds": t_fill,
        }
        if not args.no_cpu_baseline and n_ranks == 1:
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as e:  # noqa: BLE001
                out["cpu_baseline"] = {"value": None, "unit": "transitions/s", "cores": 0, "kind": "port",
                                       "sample": "failed: %s" % e}
        print(json.dumps(out), flush=True)
    L.close()
    if n_ranks > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
