import os, sys, time, ctypes as C
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import numpy as np, torch
from smarties_amd import capi, load_hip
from oracle_api import synth_cfg, fill_synth
api = load_hip()
sc = synth_cfg(seed=7, dimS=17, dimA=6, lenMin=200, lenMax=200, pTerm=0.0)
for B in (4096, 16384):
    L = capi.Learner(api, capi.make_config(dimS=17, dimA=6, hidden=(256, 256), batchSize=B, maxTotObsNum=1048576))
    L.init_weights(); fill_synth(L, sc, 2500); L.initialize(); L.step(20); L.sync()
    api.fn("timing_enable")(L.h, 1); L.step(30); L.sync()
    tot = 0
    for nm in ("big_sample_ahead", "step_tail_kernel", "stack_gather", "gemm16_fwd0", "gemm16_fwd1", "gemm16_fwd2", "panel_head", "head_kernel", "gemm16_dx1", "gemm16_dx0", "big_dw", "gemm16_dw", "dw_direct", "splitk_reduce", "post_agg_chunks", "post_kernel"):
        ms, n = C.c_double(), C.c_int64(); api.fn("timing_get")(L.h, nm.encode(), C.byref(ms), C.byref(n))
        if n.value: print("  B %d %-18s %8.2f us x%d" % (B, nm, ms.value * 1e3, n.value)); tot += ms.value * 1e3 if nm != "big_sample_ahead" else 0
    print("  B %d sum without the sampler %.1f us" % (B, tot))
    L.close()
