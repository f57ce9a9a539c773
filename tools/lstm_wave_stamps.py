"""Device time stamps of the first sample's layer-0 wavefront (library built with -DREC_STAMPS) in the one-launch LSTM step
(lstm32_step_wave_kernel: entry, prologue, every iteration of the forward loop, head, backward, end) or, with
SMARTIES_HIP_GENERIC=4, in lstm32_forward_wave_kernel (entry, prologue, every iteration of the window loop, end)."""
import sys, ctypes as C
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from smarties_amd import capi, load_hip
from oracle_api import fill_synth, synth_cfg
api = load_hip()
cfg = dict(dimS=4, dimA=1, bounded=[1], hidden=(32, 32), nnFunc="Tanh", batchSize=128, maxTotObsNum=262144, randSeed=1, gamma=0.99,
           adv_kind=capi.ADV_GAUSSIAN, nn_type=capi.NN_LSTM, nnLambda=1e-6, explNoise=0.1, nnBPTTseq=16)
L = capi.Learner(api, capi.make_config(**cfg)); L.init_weights()
fill_synth(L, synth_cfg(seed=3, dimS=4, dimA=1, lenMin=100, lenMax=300, pTerm=0.7), 100)
L.initialize(); L.step(50); L.sync()
buf = (C.c_ulonglong * 256)()
for rep in range(4):
    L.step(8); L.sync()
    assert api.lib.hl_debug_rec_stamps(buf) == 0
    st = np.array(list(buf), dtype=np.int64)
    rel = (st - st[0]) * 10
    its = [int(rel[4 + i]) for i in range(19) if st[4 + i] >= st[0]]
    print("loads issued %d ns, drained %d, loop from %d to %d (end %d): %d iterations, %s ns each" % (
        rel[1], rel[2], its[0], its[-1], rel[250], len(its), np.diff(its).tolist()))
    if st[30] >= st[0]:
        print("   forward done %d ns, backward weights requested %d, head done (wavefront 1) %d, backward from %d to %d" % (rel[30], rel[31], rel[32], rel[33], rel[250]))
