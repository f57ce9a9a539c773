import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from smarties_amd import capi, load_hip
from oracle_api import fill_synth, synth_cfg
api = load_hip()
for nn, nm in ((capi.NN_LSTM, "LSTM"), (capi.NN_MGU, "MGU")):
    res = []
    for run in range(2):
        cfg = dict(dimS=4, dimA=1, bounded=[1], hidden=(32, 32), nnFunc="Tanh", batchSize=128, maxTotObsNum=60000, randSeed=1, gamma=0.99,
                   adv_kind=capi.ADV_GAUSSIAN, nn_type=nn, nnLambda=1e-6, explNoise=0.1)
        L = capi.Learner(api, capi.make_config(**cfg)); L.init_weights()
        fill_synth(L, synth_cfg(seed=3, dimS=4, dimA=1, lenMin=100, lenMax=300, pTerm=0.7), 250)
        L.initialize()
        t0 = time.perf_counter(); L.step(6000); L.sync(); dt = time.perf_counter() - t0
        w = L.get_params()[0]; sc = L.scalars()
        res.append((w.copy(), sc.beta, sc.nFarPolicySteps))
        print(nm, 'run %d: 6000 steps %.1f us/step beta %.9g nFar %d |w| %.5f finite %s' % (run, dt / 6000 * 1e6, sc.beta, sc.nFarPolicySteps, float(np.linalg.norm(w)), bool(np.isfinite(w).all())))
        L.close()
    print(nm, 'two runs bit-identical:', np.array_equal(res[0][0], res[1][0]) and res[0][1] == res[1][1] and res[0][2] == res[1][2])
