"""One replica per PROCESS (2 or 8 of them), all on cuda:0, exchanging through hipIpc-mapped windows (hl_xchg_export / hl_xchg_connect):
launched twice by test_hip_parity.py::test_one_kernel_exchange_between_two_processes through torch.distributed.run (gloo carries
the handles and, at the end, the weights).  Rank 0 then repeats the run with two host-summed replicas of its own and compares
bit for bit."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE]
import numpy as np
import torch
import torch.distributed as dist
from smarties_amd import capi, load_hip
from oracle_api import synth_cfg, synth_episode

CFG = dict(dimS=5, dimA=2, bounded=[1, 0], hidden=(32, 32), batchSize=16, maxTotObsNum=4096, randSeed=11)
SC = synth_cfg(seed=3, dimS=5, dimA=2, lenMin=8, lenMax=30, pTerm=0.5)
if os.environ.get("XCHG_IPC_SHAPE") == "north-star":      # BASELINE.json's metric configuration: a 292 KB message, 354 pushing dW tiles (round 6)
    CFG = dict(dimS=17, dimA=6, hidden=(256, 256), nnFunc="SoftSign", batchSize=256, maxTotObsNum=65536, randSeed=42)
    SC = synth_cfg(seed=7, dimS=17, dimA=6, lenMin=40, lenMax=200, pTerm=0.3)
CALLS = [1, 1, 3, 20, 70, 900, 10]          # 1005 steps: eager calls, replayed graphs, the 1000th-step sweep (moments exchange)


NR = int(os.environ.get("WORLD_SIZE", "2"))


def replica(api, r, w0=None):
    L = capi.Learner(api, capi.make_config(n_ranks=NR, rank=r, **CFG))
    L.init_weights()
    for e in range(r, 40 * max(1, NR // 2), NR):
        L.append_episode(**synth_episode(SC, e))
    return L


def main():
    dist.init_process_group(backend="gloo")
    rank = dist.get_rank()
    api = load_hip()
    L = replica(api, rank)
    # same start as the host-summed run below: common weights, statistics of the local shard (host-exchange mode), THEN connected
    w0 = [replica(api, 0).get_params()[0]] if rank != 0 else [L.get_params()[0]]
    wv, m1, m2 = L.get_params(); L.set_params(w0[0], m1, m2)
    L.initialize()
    handles = [None] * NR
    dist.all_gather_object(handles, L.xchg_export())
    L.xchg_connect(handles)
    for n in CALLS:
        L.step(n)
    L.sync()
    w = torch.from_numpy(L.get_params()[0].copy())
    beta = torch.tensor([L.scalars().beta], dtype=torch.float64)
    ws = [torch.zeros_like(w) for _ in range(NR)]; bs = [torch.zeros_like(beta) for _ in range(NR)]
    dist.all_gather(ws, w); dist.all_gather(bs, beta)
    ok = True
    if rank == 0:
        ok = all(bool(torch.equal(ws[0], ws[r])) and float(bs[0]) == float(bs[r]) for r in range(1, NR))
        # the same run with the sums formed on the host (hl_step_begin / hl_grad_exchange ... hl_step_end)
        H = [replica(api, r) for r in range(NR)]
        w0 = H[0].get_params()[0]
        for Lh in H:
            wv, m1, m2 = Lh.get_params(); Lh.set_params(w0, m1, m2); Lh.initialize()
        for _ in range(sum(CALLS)):
            for Lh in H:
                Lh.step_begin()
            gs = [Lh.grad_fetch() for Lh in H]
            g = gs[0].copy()
            for q in gs[1:]:
                g = (g + q).astype(np.float32)                 # rank order, fp32: what the exchange kernel does
            ms = [Lh.moments_fetch() for Lh in H]
            c = np.sum([Lh.counters_fetch() for Lh in H], axis=0)
            for Lh, m in zip(H, ms):
                Lh.grad_store(g)
                if m is not None:
                    mm = ms[0].copy()
                    for q in ms[1:]:
                        mm = mm + q
                    Lh.moments_store(mm)
                Lh.counters_store(c)
                Lh.step_end()
        wh = H[0].get_params()[0]
        same = np.array_equal(wh, ws[0].numpy())
        print("replicas identical: %s; equal to the host-summed run: %s (max diff %.3e); beta %.9f vs %.9f" % (
            ok, same, float(np.abs(wh - ws[0].numpy()).max()), float(bs[0]), H[0].scalars().beta), flush=True)
        ok = ok and same and float(bs[0]) == H[0].scalars().beta
        print("XCHG_IPC_OK" if ok else "XCHG_IPC_MISMATCH", flush=True)
    dist.barrier()
    L.close()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
