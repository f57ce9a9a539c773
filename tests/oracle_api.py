"""Test-only access to the CPU oracle (oracle/liboracle_port.so, prefix ol_) and to the
synthetic replay generator of oracle/synth.h.  Never imported by the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

from smarties_amd.capi import CApi, Learner

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_LIB = os.path.join(ROOT, "oracle", "liboracle_port.so")


class SynthCfg(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("dimS", C.c_int), ("dimA", C.c_int), ("lenMin", C.c_int),
                ("lenMax", C.c_int), ("pTerminated", C.c_double), ("muSpread", C.c_double),
                ("actNoise", C.c_double)]


_api = None


def oracle_api():
    global _api
    if _api is None:
        if not os.path.exists(ORACLE_LIB):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "port"])
        _api = CApi(ORACLE_LIB, "ol_")
        lib = _api.lib
        lib.ol_synth_episode_len.restype = C.c_int
        lib.ol_synth_episode_len.argtypes = [C.POINTER(SynthCfg), C.c_uint64, C.POINTER(C.c_int)]
        lib.ol_synth_episode_discrete.restype = None
        lib.ol_synth_episode_discrete.argtypes = [C.POINTER(SynthCfg), C.c_int, C.c_uint64, C.c_void_p, C.c_void_p,
                                                  C.c_void_p, C.c_void_p, C.c_void_p]
        lib.ol_synth_episode.restype = None
        lib.ol_synth_episode.argtypes = [C.POINTER(SynthCfg), C.c_uint64, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p]
        lib.ol_head_vracer.restype = None
        lib.ol_head_vracer.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return _api


def oracle_learner(cfg):
    return Learner(oracle_api(), cfg)


def synth_cfg(seed=7, dimS=17, dimA=6, lenMin=201, lenMax=201, pTerm=0.0, muSpread=0.5, actNoise=1.0):
    return SynthCfg(seed, dimS, dimA, lenMin, lenMax, pTerm, muSpread, actNoise)


def synth_episode(sc, e, n_options=0):
    """Episode `e` of the deterministic synthetic replay (oracle/synth.h); n_options > 0: the discrete-action variant
    (actions = label + 0.1, behaviour policies = n_options probabilities)."""
    lib = oracle_api().lib
    term = C.c_int()
    n = lib.ol_synth_episode_len(C.byref(sc), e, C.byref(term))
    S = np.zeros((n, sc.dimS), np.float32)
    A = np.zeros((n, sc.dimA), np.float64)
    MU = np.zeros((n, n_options if n_options else 2 * sc.dimA), np.float64)
    R = np.zeros(n, np.float64)
    V = np.zeros(n, np.float32)
    if n_options:
        lib.ol_synth_episode_discrete(C.byref(sc), n_options, e, S.ctypes.data, A.ctypes.data, MU.ctypes.data,
                                      R.ctypes.data, V.ctypes.data)
    else:
        lib.ol_synth_episode(C.byref(sc), e, S.ctypes.data, A.ctypes.data, MU.ctypes.data, R.ctypes.data,
                             V.ctypes.data)
    return dict(states=S, actions=A, mu=MU, rewards=R, values=V, terminated=term.value, tag=e)


def fill_synth(learner, sc, n_eps, first=0):
    for e in range(first, first + n_eps):
        ep = synth_episode(sc, e, getattr(learner, "nOptions", 0))
        learner.append_episode(**ep)


def head_vracer(bounded, O, act, mu, Qret, beta, Cmax, Cinv):
    lib = oracle_api().lib
    dA = len(act)
    bounded = np.ascontiguousarray(bounded, np.uint8)
    O, act, mu = [np.ascontiguousarray(a, np.float64) for a in (O, act, mu)]
    grad = np.zeros(1 + 2 * dA)
    rho, dkl, dq, V = C.c_double(), C.c_double(), C.c_double(), C.c_double()
    far = C.c_int()
    lib.ol_head_vracer(dA, bounded.ctypes.data, O.ctypes.data, act.ctypes.data, mu.ctypes.data, Qret,
                       beta, Cmax, Cinv, grad.ctypes.data, C.addressof(rho), C.addressof(dkl),
                       C.addressof(dq), C.addressof(far), C.addressof(V))
    return grad, rho.value, dkl.value, dq.value, far.value, V.value
