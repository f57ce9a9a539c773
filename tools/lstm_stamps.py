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
lib = api.lib if hasattr(api, 'lib') else api._lib
buf = (C.c_ulonglong * 256)()
for rep in range(3):
    L.step(1); L.sync()
    assert lib.hl_debug_rec_stamps(buf) == 0
    st = np.array(list(buf), dtype=np.int64)
    t0 = st[0]
    rel = (st - t0) * 10  # ns (100 MHz)
    print("prologue ns:", rel[1], rel[2], rel[3], " end:", rel[250])
    ks = []
    for k in range(18):
        row = rel[4 + 5 * k: 9 + 5 * k]
        if st[4 + 5 * k] < t0: break
        ks.append(row.tolist())
    for k, row in enumerate(ks[:4] + ks[-2:]): print(k, [row[0], row[2], row[4]], [row[2] - row[0], row[4] - row[2]])
    print("steps stamped", len(ks))
