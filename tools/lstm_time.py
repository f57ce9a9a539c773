import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from smarties_amd import capi, load_hip
from oracle_api import fill_synth, synth_cfg
api = load_hip()
KIND = sys.argv[1] if len(sys.argv) > 1 else "lstm"      # lstm | mgu
cfg = dict(dimS=4, dimA=1, bounded=[1], hidden=(32, 32), nnFunc="Tanh", batchSize=128, maxTotObsNum=262144, randSeed=1, gamma=0.99,
           adv_kind=capi.ADV_GAUSSIAN, nn_type=capi.NN_LSTM if KIND == 'lstm' else capi.NN_MGU, nnLambda=1e-6, explNoise=0.1)   # settings/RACER_RNN.json on cart-pole (4 observed states)
L = capi.Learner(api, capi.make_config(**cfg)); L.init_weights()
fill_synth(L, synth_cfg(seed=3, dimS=4, dimA=1, lenMin=100, lenMax=300, pTerm=0.7), 400)
L.initialize(); L.step(200); L.sync()
t0 = time.perf_counter(); L.step(3000); L.sync(); dt = time.perf_counter() - t0
print(KIND.upper() + ' RACER_RNN config: %.1f us per step, %.0f transitions/s (batch 128, BPTT 16)' % (dt / 3000 * 1e6, 128 * 3000 / dt))
L.timing_enable(True) if hasattr(L, 'timing_enable') else None
L.timing_enable(True); L.step(200); L.sync()
for k in ("step_tail_kernel", "rec_step_fused", "rec_forward", "head_kernel", "panel_head", "rec_backward", "gemm16_dw", "dw_wide", "splitk_reduce", "post_kernel", "adam_kernel"):
    try:
        print(k, L.timing_get(k))
    except Exception as e:
        print(k, "n/a", e)
