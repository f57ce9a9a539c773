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
(strong scaling: `value`), one fp32 gradient sum per step (peer windows over xGMI, or RCCL); the same
steps with batch 256 and 1M transitions PER replica follow as `weak_scaling_row` (outside `value`).

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events on the library's
stream (hl_kernel_profile: graph-replayed launches of each kernel of the step; these passes run AFTER the timed region);
`cpu_baseline` times the
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
PROBE_SECONDS = int(os.environ.get("SMARTIES_BENCH_PROBE_SECONDS", "300"))
CFG = dict(dimS=17, dimA=6, hidden=(256, 256), nnFunc="SoftSign", batchSize=256, maxTotObsNum=1000000,
           clipImpWeight=4.0, penalTol=0.1, epsAnneal=0.0, gamma=0.995, lambda_=1.0, learnrate=1e-4,
           explNoise=0.4472135955, outWeightsPrefac=0.1, nnLambda=0.0, randSeed=42)


def step_kernels(B, dS, dA, hidden, nParams, fused):
    """The launches of one replayed step: (profile id, name in a kernel trace, bound, algorithmic
    work per launch).  Work = FLOPs for the MFMA GEMMs, bytes for the HBM/latency-bound head
    (kernel table, states the same figures)."""
    dims = [dS] + list(hidden)
    H = dims[-1]
    nDense = 1 + dA
    nOut = nDense + dA
    L = len(hidden)
    dw_flops = sum(2.0 * B * (dims[j] + 1) * dims[j + 1] for j in range(L)) + 2.0 * B * (H + 1) * nDense
    # head: read last hidden activations + output weights, the minibatch's action / mu / reward rows,
    # write the two delta tiles (pre-residual and pre-activation) and the output gradients; the
    # rider gathers the NEXT minibatch: 2 state rows per sample (row t and t+1) plus its scalars
    head_bytes = B * (H * 4 + 2 * H * 4 + 2 * dA * 8 + 8 * 8 + nOut * 4) + H * 8 * 4 + \
        B * (2 * dS * 4 + 2 * dA * 8 + 6 * 8)
    if fused:
        # two launches per step: forward + head + dX (one kernel, unique FLOPs counted once although
        # the first layer is recomputed by every workgroup of a row panel), then dW + Adam
        fl = 2.0 * B * dims[0] * dims[1] + 2.0 * B * dims[1] * dims[2] + 2 * (2.0 * B * H * nDense) + 2.0 * B * dims[2] * dims[1]
        return [(26, "fused_fwd_head_dx_kernel", "mfma", fl), (27, "dw_table_kernel", "mfma", dw_flops)]
    ks = [(21, "gemm16_kernel<0>", "mfma", 2.0 * B * dims[0] * dims[1])]
    if L > 1:
        ks.append((22, "gemm16_kernel<1>", "mfma", 2.0 * B * dims[L - 1] * dims[L]))
    ks.append((23, "head_kernel_t", "hbm", float(head_bytes)))
    if L > 1:
        ks.append((24, "gemm16_kernel<2>", "mfma", 2.0 * B * dims[L] * dims[L - 1]))
    ks.append((25, "gemm16_kernel<3>", "mfma", dw_flops))
    return ks


def pmc_traffic(kernel, profiles_dir=None):
    """HBM-side bytes per launch of `kernel` from the newest committed rocprofv3 --pmc summary under profiles/ that lists it
    (PMC counters cannot be read from inside this process).  Two layouts are understood: the curated one
    ({"kernels": {name: {"traffic_bytes": ...}}}) and the raw summary tools/pmc3.sh writes ({name: {"FETCH_SIZE": [mean KB,
    dispatches, max KB], "WRITE_SIZE": [...]}}).  For the raw one the corrections of MI355X_MICROARCH.md (HBM) are applied here:
    FETCH_SIZE reports half the bytes of wide coalesced reads on gfx950 -> doubled; WRITE_SIZE as reported (uncalibrated).
    Returns (bytes or None, file name or None)."""
    import glob
    import re
    d = profiles_dir or os.path.join(ROOT, "profiles")
    base = kernel.split("<")[0].replace("void ", "").replace("hl::", "")

    def rnd(f):
        m = re.match(r"r(\d+)", os.path.basename(f))
        return int(m.group(1)) if m else -1
    for f in sorted(glob.glob(os.path.join(d, "r*_pmc.json")), key=rnd, reverse=True):
        try:
            with open(f) as fh:
                j = json.load(fh)
        except Exception:  # noqa: BLE001
            continue
        k = j.get("kernels", {}).get(base) if isinstance(j.get("kernels"), dict) else None
        if isinstance(k, dict) and k.get("traffic_bytes") is not None:
            return float(k["traffic_bytes"]), os.path.basename(f)
        raw = j.get(base)
        if isinstance(raw, dict) and "FETCH_SIZE" in raw and "WRITE_SIZE" in raw:
            return (2.0 * raw["FETCH_SIZE"][0] + raw["WRITE_SIZE"][0]) * 1024.0, os.path.basename(f)
    return None, None


def synthetic_episode(np, e, dS=17, dA=6, N=EP_STATES):
    """Episode `e` of the synthetic replay: the distributions of oracle/synth.h (the generator the
    CPU baseline's harness uses), drawn from a numpy Generator seeded by the episode index."""
    g = np.random.default_rng(1000003 * 7 + e)
    i = np.arange(dS)
    S = (g.standard_normal((N, dS)) * (0.5 + 0.1 * i) + (0.2 * i - 1.0)).astype(np.float32)
    R = g.standard_normal(N) + 0.1
    R[0] = 0.0
    mean = 0.5 * g.standard_normal((N, dA))
    std = 0.3 + 0.4 * g.random((N, dA))
    A = mean + std * g.standard_normal((N, dA))
    MU = np.concatenate([mean, std], axis=1)
    A[-1] = 0.0
    MU[-1] = 0.0
    V = (0.5 * g.standard_normal(N)).astype(np.float32)
    return dict(states=S, actions=A, mu=MU, rewards=R, values=V, terminated=0, tag=e)


def other_configs(api, steps=1000, only=None):
    """us per gradient step of the other BASELINE.json configurations (parity-tested shapes, not the bench workload): small
    synthetic replays resident in HBM, `steps` replayed steps after a warm-up.  Reported next to the bench line."""
    import numpy as np
    from smarties_amd import capi
    conv = [(84, 84, 4, 8, 8, 4), (20, 20, 8, 16, 6, 2), (8, 8, 16, 32, 4, 1), (5, 5, 32, 64, 3, 1)]
    cases = {
        "cfg1_cart_pole_vracer_2x128_b256": (dict(dimS=5, dimA=1, bounded=[1], hidden=(128, 128), nnFunc="Tanh", batchSize=256, maxTotObsNum=40131,
                                                  clipImpWeight=0.5 ** 0.5, explNoise=0.2 ** 0.5), 300, 120, steps),
        "cfg3_humanoid_replica_2x256_b32": (dict(dimS=257, dimA=17, bounded=[0] * 17, hidden=(256, 256), batchSize=32, maxTotObsNum=131072,
                                                 clipImpWeight=(17 / 2.0) ** 0.5), 300, 200, steps),
        "cfg4_racer_rnn_lstm_2x32_b128_bptt16": (dict(dimS=4, dimA=1, bounded=[1], hidden=(32, 32), nnFunc="Tanh", batchSize=128, maxTotObsNum=262144,
                                                      gamma=0.99, adv_kind=capi.ADV_GAUSSIAN, nn_type=capi.NN_LSTM, nnLambda=1e-6, explNoise=0.1), 300, 200, steps),
        "cfg4_mgu_2x32_b128_bptt16": (dict(dimS=4, dimA=1, bounded=[1], hidden=(32, 32), nnFunc="Tanh", batchSize=128, maxTotObsNum=262144,
                                           gamma=0.99, adv_kind=capi.ADV_GAUSSIAN, nn_type=capi.NN_MGU, nnLambda=1e-6, explNoise=0.1), 300, 200, steps),
        # recurrent layers wider than the 32 cells of RACER_RNN.json: time-step-major launches on the MFMA (rectm.hip)
        "lstm_2x128_b128_bptt16": (dict(dimS=4, dimA=1, bounded=[1], hidden=(128, 128), nnFunc="Tanh", batchSize=128, maxTotObsNum=262144,
                                        gamma=0.99, adv_kind=capi.ADV_GAUSSIAN, nn_type=capi.NN_LSTM, nnLambda=1e-6, explNoise=0.1), 300, 200, max(50, steps // 10)),
        "lstm_2x256_b128_bptt16": (dict(dimS=4, dimA=1, bounded=[1], hidden=(256, 256), nnFunc="Tanh", batchSize=128, maxTotObsNum=262144,
                                        gamma=0.99, adv_kind=capi.ADV_GAUSSIAN, nn_type=capi.NN_LSTM, nnLambda=1e-6, explNoise=0.1), 300, 200, max(50, steps // 20)),
        "mgu_2x256_b128_bptt16": (dict(dimS=4, dimA=1, bounded=[1], hidden=(256, 256), nnFunc="Tanh", batchSize=128, maxTotObsNum=262144,
                                       gamma=0.99, adv_kind=capi.ADV_GAUSSIAN, nn_type=capi.NN_MGU, nnLambda=1e-6, explNoise=0.1), 300, 200, max(50, steps // 20)),
        # the bench network at larger batches: where the step stops being a latency chain (fraction of the fp32 MFMA peak below)
        # (replays of 500 000 transitions: the sampler redraws until the minibatch is unique (Sampling.cpp:86-93) -- at 80 000 stored
        #  transitions a batch of 1024 needs several rounds (62 us per step, the sampler riding the step's kernels their longest
        #  workgroup; 51 on this replay), at five times the batch seven)
        "cfgNS_2x256_b1024": (dict(dimS=17, dimA=6, hidden=(256, 256), batchSize=1024, maxTotObsNum=1048576), 2500, 200, max(100, steps // 2)),
        "cfgNS_2x256_b2048": (dict(dimS=17, dimA=6, hidden=(256, 256), batchSize=2048, maxTotObsNum=1048576), 2500, 200, max(100, steps // 4)),      # (first batch of the large-batch launches: bigmm.hip)
        "cfgNS_2x256_b4096": (dict(dimS=17, dimA=6, hidden=(256, 256), batchSize=4096, maxTotObsNum=1048576), 2500, 200, max(100, steps // 4)),
        "cfgNS_2x256_b16384": (dict(dimS=17, dimA=6, hidden=(256, 256), batchSize=16384, maxTotObsNum=1048576), 2500, 200, max(50, steps // 10)),
        # settings/RACER_glider.json (a shipped preset): RACER with the Gaussian advantage, three hidden layers of 128 -- the generic launches
        "glider_racer_3x128_gauss_b256": (dict(dimS=10, dimA=1, bounded=[1], hidden=(128, 128, 128), nnFunc="Tanh", batchSize=256, maxTotObsNum=524288,
                                               gamma=1.0, adv_kind=capi.ADV_GAUSSIAN, epsAnneal=2e-7, nnLambda=1e-6, penalTol=0.05, clipImpWeight=1.0), 400, 200, steps),
        # combinations no settings file builds, replayed as captured launch lists since round 5: an LSTM layer behind two convolutions on
        # 1 + 3 stacked frames; an RNN encoder under MGU layers (a partially observable MDP with nnType left at its default)
        "conv2_lstm32_b64_bptt4": (dict(dimS=256, dimA=1, adv_kind=capi.ADV_DISCRETE, n_options=5, nAppendedObs=3, conv=[(8, 8, 16, 32, 4, 1), (5, 5, 32, 64, 3, 1)],
                                        hidden=(32,), nnFunc="Tanh", batchSize=64, maxTotObsNum=65536, nn_type=capi.NN_LSTM, nnBPTTseq=4), 300, 60, max(100, steps // 4)),
        "pomdp_rnn24_mgu2x16_b128_bptt5": (dict(dimS=5, dimA=2, bounded=[1, 0], hidden=(16, 16), encoder=[24], encoder_rnn=1, nn_type=capi.NN_MGU, nnFunc="Tanh",
                                                batchSize=128, maxTotObsNum=131072, adv_kind=capi.ADV_GAUSSIAN, nnBPTTseq=5), 300, 100, max(100, steps // 4)),
        "cfg5_racer_atari_conv4_512_b128": (dict(dimS=7056, dimA=1, adv_kind=capi.ADV_DISCRETE, n_options=6, nAppendedObs=3, conv=conv, hidden=(512,),
                                                 nnFunc="Tanh", batchSize=128, maxTotObsNum=20000, gamma=0.99, explNoise=0.05), 120, 60, max(100, steps // 5)),
    }
    res = {}
    for name, (kw, nEp, N, n) in cases.items():
        if only and not any(o in name for o in only):
            continue
        g = np.random.default_rng(5)
        try:
            L = capi.Learner(api, capi.make_config(randSeed=7, **kw))
        except capi.HlError as e:
            res[name] = {"error": str(e)}
            continue
        L.init_weights()
        dS, dA = kw["dimS"], kw["dimA"]
        nopt = kw.get("n_options", 0)
        for e in range(nEp):
            S = g.standard_normal((N, dS)).astype(np.float32)
            if nopt:
                A = g.integers(0, nopt, size=(N, 1)).astype(np.float64) + 0.1
                MU = g.random((N, nopt)) + 0.2
                MU /= MU.sum(1, keepdims=True)
            else:
                mean = 0.5 * g.standard_normal((N, dA)); std = 0.3 + 0.4 * g.random((N, dA))
                A = mean + std * g.standard_normal((N, dA)); MU = np.concatenate([mean, std], axis=1)
            R = g.standard_normal(N); R[0] = 0
            A[-1] = 0; MU[-1] = 0
            L.append_episode(states=S, actions=A, mu=MU, rewards=R, values=(0.5 * g.standard_normal(N)).astype(np.float32),
                             terminated=int(e % 3 == 0), tag=e)
        L.initialize()
        L.step(64); L.sync()
        t0 = time.perf_counter(); L.step(n); L.sync(); dt = time.perf_counter() - t0
        res[name] = {"us_per_step": round(dt / n * 1e6, 2), "transitions_per_s": round(kw["batchSize"] * n / dt), "steps": n}
        if name.startswith("cfgNS_"):       # 421 376 FLOP per transition (forward, dX, dW of the 17-256-256-7 network)
            tf = 421376.0 * kw["batchSize"] / (dt / n) / 1e12
            res[name].update({"tflops": round(tf, 2), "frac_of_fp32_mfma_peak": round(tf / PEAK_FP32_MFMA_TFLOPS, 4)})
        L.close()
    return res


def cpu_baseline(steps_budget_s=20.0):
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_driver_fast")
    ncpu = os.cpu_count() or 1
    if os.path.exists(ref):
        best = None
        tried = []
        with tempfile.TemporaryDirectory() as td:
            # (probes up to the box's CPU count -- VERDICT r05: they stopped at 64 of 256; a probe is 400 steps, a few seconds each)
            probes = [t for t in (8, 16, 32, 64, 128, 256) if t <= ncpu] or [ncpu]
            if ncpu > probes[-1] and ncpu <= 512:
                probes.append(ncpu)
            for thr in probes:
                env = dict(os.environ, OMP_NUM_THREADS=str(thr), OMP_PROC_BIND="close", OMP_PLACES="cores")
                try:
                    out = subprocess.run([ref, "bench", "threads=%d" % thr, "nObs=1000000", "nSteps=400", "warmup=20"],
                                         cwd=td, env=env, capture_output=True, text=True, timeout=180).stdout
                    line = [l for l in out.splitlines() if l.startswith("{\"kind\"")][-1]
                    r = json.loads(line)
                    tried.append((thr, r["transitions_per_s"]))
                    if best is None or r["transitions_per_s"] > best["transitions_per_s"]:
                        best = r
                    elif r["transitions_per_s"] < 0.5 * best["transitions_per_s"]:
                        break      # (past the knee: the next probes only get slower -- 256 threads took more than three minutes for 400 steps)
                except Exception as e:  # noqa: BLE001
                    tried.append((thr, "failed: %s" % e))
            if best is not None:      # the quoted value: a longer run at the best thread count (the 400-step probes are noisy)
                thr = int(best["threads"])
                env = dict(os.environ, OMP_NUM_THREADS=str(thr), OMP_PROC_BIND="close", OMP_PLACES="cores")
                try:
                    out = subprocess.run([ref, "bench", "threads=%d" % thr, "nObs=1000000", "nSteps=6000", "warmup=100"],
                                         cwd=td, env=env, capture_output=True, text=True, timeout=300).stdout
                    best = json.loads([l for l in out.splitlines() if l.startswith("{\"kind\"")][-1])
                    long_run = True
                except Exception:  # noqa: BLE001
                    long_run = False
        if best is not None:
            return {"value": best["transitions_per_s"], "unit": "transitions/s", "cores": int(best["threads"]),
                    "kind": "reference", "blas": "none (the reference's own OpenMP-SIMD loops: neither USE_MKL nor USE_OPENBLAS, as in its CMake build; "
                                                 "a CBLAS flavour cannot be built here: the image holds libmkl_rt.so but no cblas / mkl header, and a hand-written "
                                                 "prototype header would be a stand-in for a file the image lacks -- DESIGN.md section 5)",
                    "march": "x86-64-v3", "flags": "-O3 -ffast-math -fopenmp -DSINGLE_PREC",
                    "sample": "compiled reference (oracle/_ref, -O3 -ffast-math, OpenMP) on a 1M-transition synthetic replay of "
                              "the same shape and distributions: 400-step probes at threads=%s on %d host CPUs, then %s at the best count"
                              % ([t for t, _ in tried], ncpu, "6000 gradient steps after 100 warm-up" if long_run else "(long run failed) the probe"),
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
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the us/step of the other BASELINE configurations")
    ap.add_argument("--no-roofline", action="store_true", help="skip the per-kernel HIP-event passes (counter-collection runs, which stall under their graph replays)")
    ap.add_argument("--no-diagnostics", action="store_true", help="skip the repeated call and the 2000-step sustained rate after the timed region (counter-collection runs)")
    args = ap.parse_args()
    # the ONE JSON line goes to the real stdout; everything else this process or its libraries print there (RCCL's version banner when
    # a communicator comes up, gloo's connection notes) is sent to stderr
    real_out_fd = os.dup(1)
    sys.stdout.flush(); os.dup2(2, 1)
    real_out = os.fdopen(real_out_fd, "w")

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
    # SMARTIES_BENCH_EXCHANGE=host: replicas exchange through the host (gloo + the split-step entry points) instead of RCCL
    # inside the library -- the slow but dependency-free protocol, also the automatic fall-back if the communicator cannot be set up
    host_exchange = os.environ.get("SMARTIES_BENCH_EXCHANGE", "") == "host"
    ndev = torch.cuda.device_count()
    torch.cuda.set_device(local_rank % ndev)
    host_group = None
    dog = None
    if n_ranks > 1 and not host_exchange:
        # The RCCL exchange (communicator set-up, the collective inside the replayed graph) is the one path this
        # repository could never run on more than one device before the scaling bench itself.  Set-up and a two-step
        # probe run under a watchdog: if anything wedges, every rank starts over in the host-exchange protocol
        # (exercised by the gloo tests) rather than losing the whole measurement.
        import threading

        def start_over():
            print("rank %d: RCCL exchange did not come up within %ds: restarting with the host exchange" % (
                rank, PROBE_SECONDS), file=sys.stderr, flush=True)
            env = dict(os.environ, SMARTIES_BENCH_EXCHANGE="host", SMARTIES_BENCH_REEXEC="1")
            env.pop("TORCHELASTIC_USE_AGENT_STORE", None)     # (else every rank of the second life is a store client)
            os.dup2(real_out_fd, 1)      # (the second life prints its line where this one would have)
            os.execve(sys.executable, [sys.executable] + sys.argv, env)

        dog = threading.Timer(PROBE_SECONDS, start_over)
        dog.daemon = True
        dog.start()
        if os.environ.get("SMARTIES_BENCH_TEST_HANG"):     # (proves that the second life works, see tools/README.md)
            time.sleep(10 * PROBE_SECONDS)
    if n_ranks > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("SMARTIES_BENCH_REEXEC"):
            # second life after the watchdog below fired: the launcher's store still holds the first life's keys,
            # so rank 0 opens a store of its own next to it
            dist.init_process_group(backend="gloo", rank=rank, world_size=n_ranks, init_method="tcp://127.0.0.1:%d" % (
                int(os.environ.get("MASTER_PORT", "29500")) + 17))
        else:
            # (SMARTIES_BENCH_PG=gloo: handles and flags travel over gloo -- two ranks on ONE device, as on the 1-GPU test box)
            dist.init_process_group(backend="gloo" if (host_exchange or os.environ.get("SMARTIES_BENCH_PG") == "gloo") else "nccl",
                                    rank=rank, world_size=n_ranks)

    from smarties_amd import capi, load_hip

    # replicas reach their first collectives seconds apart (set-up, graph capture): a generous bound for the exchange kernel's wait
    os.environ.setdefault("SMARTIES_HIP_XCHG_TIMEOUT_MS", "60000")
    api = load_hip()
    state = {"host_exchange": host_exchange, "host_group": None}

    def build(cfg_kw, n_episodes):
        """learner of this rank for the configuration cfg_kw (batch and replay budget are GLOBAL figures, split over the replicas
        as Settings/HyperParameters.cpp:186-197 does), filled with its share of n_episodes synthetic episodes, replicas connected"""
        cfg = capi.make_config(n_ranks=n_ranks, rank=rank, device_id=local_rank % ndev, **cfg_kw)
        per = n_episodes // n_ranks

        def make_learner():
            L_ = capi.Learner(api, cfg)
            L_.init_weights()
            t_ = time.time()
            for e in range(rank * per, (rank + 1) * per):
                L_.append_episode(**synthetic_episode(np, e))
            return L_, time.time() - t_

        L, t_fill = make_learner()
        transport = "single replica"
        if n_ranks > 1 and not state["host_exchange"]:
            # 1st choice: the library's own one-kernel exchange through peer-mapped windows (xchg.hip; handles travel through the
            # process group); 2nd: its RCCL communicator; 3rd: sums on the host (gloo).  Every decision is taken by ALL ranks.
            pg_dev = "cpu" if dist.get_backend() == "gloo" else "cuda"

            def all_ok(ok):
                flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=pg_dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                return int(flag.item()) == 1

            want = os.environ.get("SMARTIES_BENCH_EXCHANGE", "")
            done = False
            if want in ("", "xchg"):
                hd = None
                try:
                    hd = L.xchg_export()
                except Exception as e:  # noqa: BLE001
                    print("rank %d: hl_xchg_export failed (%s)" % (rank, e), file=sys.stderr)
                hs = [None] * n_ranks
                dist.all_gather_object(hs, hd)
                ok = all(h is not None for h in hs)
                if ok:
                    try:
                        L.xchg_connect(hs)
                    except Exception as e:  # noqa: BLE001
                        ok = False
                        print("rank %d: hl_xchg_connect failed (%s)" % (rank, e), file=sys.stderr)
                done = all_ok(ok)
                if done:
                    transport = "one-kernel exchange through peer-mapped windows over xGMI (xchg.hip), a node of the replayed graphs"
                elif ok:                       # (connected here, not everywhere: start over with a clean learner)
                    L.close()
                    L, t_fill = make_learner()
            if not done:
                ok = True
                try:
                    idbuf = torch.zeros(128, dtype=torch.uint8, device=pg_dev)
                    if rank == 0:
                        import ctypes as C
                        raw = (C.c_uint8 * 128)()
                        assert api.fn("comm_unique_id")(raw) == 0
                        idbuf = torch.tensor(list(raw), dtype=torch.uint8, device=pg_dev)
                    dist.broadcast(idbuf, 0)
                    L.comm_init(bytes(idbuf.cpu().tolist()))
                except Exception as e:  # noqa: BLE001
                    ok = False
                    print("rank %d: RCCL communicator of the library failed (%s): host exchange instead" % (rank, e), file=sys.stderr)
                if all_ok(ok):
                    transport = "RCCL inside the library (captured in the replayed graphs)"
                else:
                    state["host_exchange"] = True
                    state["host_group"] = dist.new_group(backend="gloo")
                    L.close()
                    L, t_fill = make_learner()
        if n_ranks > 1 and state["host_exchange"]:
            from smarties_amd import dist_host
            dist_host.init_replica_weights(L, dist, group=state["host_group"])
            # (the start-up counters and reward / state moments summed over ALL shards, as the reference's accurate start-up
            #  reductions do -- Learner.cpp:58-59; a plain hl_initialize would scale every replica by its own shard: ADVICE r05)
            dist_host.initialize_host_exchange(L, dist, group=state["host_group"])
        else:
            L.initialize()
        return L, t_fill, transport

    # N > 1: the single-replica value of THIS box first (rank 0 alone, the N = 1 workload, the same K steps after W warm-up steps), so
    # that the weak-scaling row can state its efficiency against N x that value in the same run; the other ranks wait at the barrier
    single_ref = None
    if n_ranks > 1 and os.environ.get("SMARTIES_BENCH_SINGLE_REF", "1") != "0":
        if rank == 0:
            try:
                L1 = capi.Learner(api, capi.make_config(device_id=local_rank % ndev, **CFG))
                L1.init_weights()
                for e in range(N_EPISODES):
                    L1.append_episode(**synthetic_episode(np, e))
                L1.initialize()
                if args.warmup > 0:
                    L1.prepare_steps(args.warmup)
                L1.prepare_steps(args.steps)
                L1.step(args.warmup); torch.cuda.synchronize(); L1.sync()
                t1 = time.perf_counter(); L1.step(args.steps); torch.cuda.synchronize(); L1.sync()
                single_ref = CFG["batchSize"] * args.steps / (time.perf_counter() - t1)
                L1.close()
            except Exception as e:  # noqa: BLE001
                print("rank 0: single-replica reference failed (%s)" % e, file=sys.stderr, flush=True)
        dist.barrier()

    L, t_fill, transport = build(CFG, N_EPISODES)
    host_exchange, host_group = state["host_exchange"], state["host_group"]
    if n_ranks > 1 and host_exchange:
        from smarties_amd import dist_host

    def run(n, Lx=None):
        Lx = Lx or L
        if n_ranks > 1 and host_exchange:
            dist_host.step_host_exchange(Lx, dist, n, group=host_group)
        else:
            Lx.step(n)

    def barrier(Lx=None):
        if n_ranks > 1:
            dist.barrier()
        # the whole device first: torch.cuda.synchronize() queues the completion marker of the library's stream WHILE the call's
        # kernels still run; after hl_sync's polled completion word it would start that round trip only then (tools/sync_cost.py:
        # 375.6 against 382.0 us per 20-step call)
        if os.environ.get("SMARTIES_BENCH_POLL_FIRST"):
            (Lx or L).sync(); torch.cuda.synchronize()
            return
        torch.cuda.synchronize()     # the whole device ...
        (Lx or L).sync()             # ... and the library's own stream (a completion stamp polled in pinned memory when the call was one graph)

    # ---- roofline of the dominant kernel ---------------------------------------------------------------
    def roofline(L):
        # hl_kernel_profile: `reps` launches of one kernel of the step (issued exactly as inside the
        # replayed step, rider workgroup included) captured into a graph and replayed between two HIP
        # events on the library's stream.  The empty-kernel profile shows how much of that is dispatch.
        gap = L.kernel_profile(12, 400)
        table = {}
        try:
            L.kernel_profile(26, 8)          # the fused forward/head/dX kernel serves this network?
            fused = True
        except capi.HlError:
            fused = False
        for pid, name, bound, work in step_kernels(L.B, 17, 6, CFG["hidden"], L.nParams, fused):
            us = L.kernel_profile(pid, 200)
            if bound == "mfma":
                ach, peak, unit = work / (us * 1e-6) / 1e12, PEAK_FP32_MFMA_TFLOPS, "TFLOP/s"
            else:
                ach, peak, unit = work / (us * 1e-6) / 1e9, PEAK_HBM_GBS, "GB/s"
            table[name] = {"bound": bound, "launch_us": round(us, 3), "work_per_launch": work, "achieved": ach,
                           "peak": peak, "unit": unit, "frac": ach / peak}
        dom = max(table, key=lambda n: table[n]["launch_us"])
        d = table[dom]
        # HBM traffic per launch of that kernel: PMC counters cannot be read from inside this process;
        # the committed rocprofv3 --pmc passes (profiles/r04_pmc.json: commands, corrections) are quoted
        traffic, traffic_src = pmc_traffic(dom)
        roof = {"kernel": dom, "bound": d["bound"], "achieved": d["achieved"], "peak": d["peak"], "unit": d["unit"],
                "frac": d["frac"], "traffic": traffic, "traffic_unit": "bytes/launch = 2 x FETCH_SIZE + WRITE_SIZE (rocprofv3 --pmc passes of eager launches, profiles/%s)" % traffic_src,
                "launch_us": d["launch_us"], "empty_launch_us": round(gap, 3),
                "step_kernels": table,
                "note": "latency-bound step: dependent launches of a few hundred workgroups; launch_us = HIP-event time "
                        "per launch of graph-replayed back-to-back launches (dispatch included, as a kernel trace "
                        "counts it); empty_launch_us = the same for an empty kernel"}
        return roof

    # calls of W and K steps will follow: their graphs are captured now (set-up, like hl_initialize's stock sizes), and the
    # address translations of the replay are made resident (hl_prepare_steps) -- before the roofline passes, so that nothing
    # but the W warm-up steps lies between those and the timed region (10 ms of idling cost the next call 20-30 us:
    # tools/first_call3.py)
    if not (n_ranks > 1 and host_exchange):
        if args.warmup > 0:
            L.prepare_steps(args.warmup)
        L.prepare_steps(args.steps)

    # (Round 2 ran the roofline passes of a second learner in front of the timed region "to bring the device to working clocks";
    # measured this round -- tools/first_call5.py -- that costs the timed call 15-20 us: they now follow the timed region.)
    roof, P = None, None

    if dog is not None:
        try:
            run(2)
            barrier()
        except Exception as e:  # noqa: BLE001  (e.g. a peer's message never arrived: the exchange kernel's bounded wait)
            print("rank %d: the two-step probe of the exchange failed (%s)" % (rank, e), file=sys.stderr, flush=True)
            dog.cancel()
            start_over()
        dog.cancel()

    run(args.warmup)
    barrier()
    t0 = time.perf_counter()
    run(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if n_ranks > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # replicas must hold the SAME parameters after the timed steps (every replica sums in rank order: bit for bit), or the line below
    # would time something that is not data-parallel training: checked before anything is printed
    if n_ranks > 1:
        import hashlib
        w_all = L.get_params()
        digest = hashlib.sha256(b"".join(np.ascontiguousarray(a).tobytes() for a in w_all)).hexdigest()
        digests = [None] * n_ranks
        dist.all_gather_object(digests, digest)
        if len(set(digests)) != 1:
            raise SystemExit("bench.py: the replicas' weights / Adam moments differ after %d steps (sha256 per rank: %s): no line printed" % (
                args.warmup + args.steps, ", ".join(d[:12] for d in digests)))

    B_global = CFG["batchSize"]
    value = B_global * args.steps / dt
    # which transport each rank ended up with (they agree by construction -- all_ok() -- but the line says so rank by rank)
    transports = ["host (gloo, split-step entry points)" if (n_ranks > 1 and host_exchange) else transport]
    if n_ranks > 1:
        try:
            gathered = [None] * n_ranks
            dist.all_gather_object(gathered, transports[0])
            transports = gathered
        except Exception as e:  # noqa: BLE001
            transports = transports + ["(gather failed: %s)" % e]

    # diagnostics, after the timed region and outside `value`: the same call once more (what a first call pays on top: page walks,
    # first launch of the call's graph, clocks) and the sustained rate over 2000 steps
    diag = {}
    if n_ranks == 1 and not args.no_diagnostics:
        t1 = time.perf_counter(); run(args.steps); barrier(); diag["same_call_again_ms_per_step"] = (time.perf_counter() - t1) / args.steps * 1e3
        run(400); barrier()
        t1 = time.perf_counter(); run(2000); barrier(); diag["sustained_ms_per_step_2000_steps"] = (time.perf_counter() - t1) / 2000 * 1e3

    if rank == 0 and roof is None and not args.no_roofline:
        roof = roofline(L)

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
                       "global_batch": B_global, "replay_transitions": 1000000, "parallelism": "dp%d" % n_ranks,
                       "exchange": "host (gloo, split-step entry points)" if (n_ranks > 1 and host_exchange) else transport,
                       "exchange_per_rank": transports,
                       "replicas_identical_after_timed_steps": (True if n_ranks > 1 else None)},
            "roofline": roof,
            "fill_seconds": t_fill,
            "diagnostics": diag,
        }
        if not args.no_other_configs and n_ranks == 1:
            try:
                out["other_configs"] = other_configs(api, steps=1000)
            except Exception as e:  # noqa: BLE001
                out["other_configs"] = {"error": str(e)}
        if not args.no_cpu_baseline and n_ranks == 1:
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as e:  # noqa: BLE001
                out["cpu_baseline"] = {"value": None, "unit": "transitions/s", "cores": 0, "kind": "port",
                                       "sample": "failed: %s" % e}
    # N > 1: the same steps with batch 256 and a 1M-transition replay PER REPLICA (weak scaling: what a replica of BASELINE config 3
    # sees -- the global batch grows with the replicas, Settings/HyperParameters.cpp:186-197 splits it back to 256 each), next to
    # the strong-scaling value.  Outside `value`; measured last, under a watchdog that prints the line without it.
    if n_ranks > 1 and os.environ.get("SMARTIES_BENCH_WEAK", "1") != "0":
        import threading

        def bail():
            if rank == 0:
                out["weak_scaling_row"] = {"scaling": "weak", "error": "did not finish within %d s" % PROBE_SECONDS}
                print(json.dumps(out), file=real_out, flush=True)
            os._exit(3)      # (non-zero: the peers may sit in collectives; the launcher tears the group down)
        wd = threading.Timer(PROBE_SECONDS, bail); wd.daemon = True; wd.start()
        try:
            L.close()
            wk = dict(CFG, batchSize=CFG["batchSize"] * n_ranks, maxTotObsNum=CFG["maxTotObsNum"] * n_ranks)
            L, _, tw = build(wk, N_EPISODES * n_ranks)
            host_exchange, host_group = state["host_exchange"], state["host_group"]
            if not host_exchange:
                if args.warmup > 0:
                    L.prepare_steps(args.warmup)
                L.prepare_steps(args.steps)
            run(args.warmup); barrier()
            tw0 = time.perf_counter(); run(args.steps); barrier(); dtw = time.perf_counter() - tw0
            t = torch.tensor([dtw], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtw = float(t.item())
            weak = {"scaling": "weak", "value": wk["batchSize"] * args.steps / dtw, "unit": "transitions/s", "ms_per_step": dtw / args.steps * 1e3,
                    "global_batch": wk["batchSize"], "replay_transitions": wk["maxTotObsNum"], "per_replica_batch": CFG["batchSize"], "exchange": tw,
                    # against N x the single-replica value measured by rank 0 at the start of this run (same K / W)
                    "single_replica_value": single_ref,
                    "efficiency_vs_n_single": (wk["batchSize"] * args.steps / dtw) / (n_ranks * single_ref) if single_ref else None}
        except Exception as e:  # noqa: BLE001
            weak = {"scaling": "weak", "error": str(e)}
        wd.cancel()
        if rank == 0:
            out["weak_scaling_row"] = weak
    if rank == 0:
        print(json.dumps(out), file=real_out, flush=True)
    try:
        L.close()
    except Exception:  # noqa: BLE001
        pass
    if n_ranks > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
