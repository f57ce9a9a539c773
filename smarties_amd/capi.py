"""ctypes binding of the hl_* C-ABI declared in include/smarties_hip.h.

The product library is ``smarties_amd/libsmarties_hip.so`` (hand-written HIP for
gfx950 behind a C-ABI).  There is NO CPU fallback: if the shared library is
missing or no HIP device is usable, loading / ``hl_create`` fail loudly.

``CApi`` is parametrised by (library path, symbol prefix) only so that the test
suite can drive the CPU oracle (``oracle/liboracle_port.so``, prefix ``ol_``)
through the very same call sequence; nothing in this package loads the oracle.
"""
import atexit
import ctypes as C
import os
import weakref

import numpy as np

HL_MAX_DIMA = 64
HL_MAX_HIDDEN = 8

FUNC = {"Linear": 0, "Tanh": 1, "SoftSign": 2, "Relu": 3, "LRelu": 4, "Sigm": 5, "HardSign": 6,
        "SoftPlus": 7, "ExpPlus": 8, "Exp": 9}
ADV_ZERO, ADV_GAUSSIAN, ADV_DISCRETE = 0, 1, 2
NN_FFNN, NN_LSTM, NN_MGU, NN_RNN = 0, 1, 2, 3
RET = {"retrace": 0, "default": 0, "retraceExplore": 1, "GAE": 2, "none": 3}
ORDER_STABLE, ORDER_REFERENCE = 0, 1

(TAP_FLAT, TAP_EPISODE, TAP_TSTEP, TAP_TAG, TAP_STATE, TAP_OUTPUT, TAP_OUTGRAD, TAP_RHO, TAP_DKL,
 TAP_DELTAQ, TAP_FAR, TAP_GRADSUM) = range(12)
XCHG_HANDLE_BYTES = 96
EP_RETURN, EP_VALUE, EP_ADVANTAGE, EP_IMPW, EP_DKL, EP_DELTAQ = range(6)

STATUS = {0: "HL_OK", 1: "HL_ERR_BAD_ARG", 2: "HL_ERR_NO_DEVICE", 3: "HL_ERR_HIP", 4: "HL_ERR_STATE",
          5: "HL_ERR_TOO_FEW_DATA", 6: "HL_ERR_COMM", 7: "HL_ERR_IO", 8: "HL_ERR_UNSUPPORTED"}


class HlConv2d(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("inpFeatures", "inpY", "inpX", "outFeatures", "outY", "outX", "filterx", "filtery",
                                         "stridex", "stridey", "paddinx", "paddiny")]


HL_MAX_CONV = 8


class HlConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("dimS", C.c_int32), ("dimA", C.c_int32),
        ("bounded", C.c_uint8 * HL_MAX_DIMA), ("n_hidden", C.c_int32),
        ("hidden", C.c_int32 * HL_MAX_HIDDEN), ("nnFunc", C.c_int32), ("adv_kind", C.c_int32),
        ("batchSize", C.c_int32), ("maxTotObsNum", C.c_int64), ("minTotObsNum", C.c_int64),
        ("gamma", C.c_double), ("lambda_", C.c_double), ("clipImpWeight", C.c_double),
        ("penalTol", C.c_double), ("epsAnneal", C.c_double), ("learnrate", C.c_double),
        ("nnLambda", C.c_double), ("explNoise", C.c_double), ("outWeightsPrefac", C.c_double),
        ("randSeed", C.c_uint64), ("n_ranks", C.c_int32), ("rank", C.c_int32),
        ("device_id", C.c_int32), ("episode_order", C.c_int32), ("ref_threads", C.c_int32),
        ("n_options", C.c_int32), ("nn_type", C.c_int32), ("nnBPTTseq", C.c_int32),
        ("nAppendedObs", C.c_int32), ("n_conv", C.c_int32), ("conv", HlConv2d * HL_MAX_CONV), ("ERoldSeqFilter", C.c_int32), ("dataSamplingAlgo", C.c_int32),
        ("returnsEstimator", C.c_int32), ("nnOutputFunc", C.c_int32), ("n_encoder", C.c_int32), ("encoder", C.c_int32 * HL_MAX_HIDDEN), ("encoder_rnn", C.c_int32),
    ]


class HlScalars(C.Structure):
    _fields_ = [("beta", C.c_double), ("alpha", C.c_double), ("CmaxRet", C.c_double),
                ("CinvRet", C.c_double), ("nGradSteps", C.c_int64), ("nStoredSteps", C.c_int64),
                ("nStoredEps", C.c_int64), ("nFarPolicySteps", C.c_int64), ("nSeenSteps", C.c_int64),
                ("nSeenEps", C.c_int64), ("adam_beta_t_1", C.c_double), ("adam_beta_t_2", C.c_double),
                ("adam_nStep", C.c_int64)]


class HlStats(C.Structure):
    _fields_ = [("avgKLdivergence", C.c_double), ("avgSquaredErr", C.c_double),
                ("maxAbsError", C.c_double), ("avgReturn", C.c_double), ("avgQ", C.c_double),
                ("stdevQ", C.c_double), ("minQ", C.c_double), ("maxQ", C.c_double),
                ("nFarPolicySteps", C.c_int64), ("countReturnsEstimateUpdates", C.c_int64),
                ("sumReturnsEstimateErrors", C.c_double)]


def make_config(dimS=17, dimA=6, bounded=None, hidden=(256, 256), nnFunc="SoftSign", batchSize=256,
                maxTotObsNum=1000000, minTotObsNum=0, gamma=0.995, lambda_=1.0, clipImpWeight=4.0,
                penalTol=0.1, epsAnneal=0.0, learnrate=1e-4, nnLambda=0.0, explNoise=0.4472135955,
                outWeightsPrefac=0.1, randSeed=42, n_ranks=1, rank=0, device_id=-1,
                episode_order=ORDER_STABLE, ref_threads=1, adv_kind=ADV_ZERO, n_options=0, nn_type=0, nnBPTTseq=0,
                nAppendedObs=0, conv=(), ERoldSeqFilter="oldest", dataSamplingAlgo="uniform",
                returnsEstimator="retrace", nnOutputFunc="Linear", encoder=(), encoder_rnn=0):
    """Defaults = the north-star synthetic of BASELINE.md (cfg-NS)."""
    c = HlConfig()
    c.struct_size = C.sizeof(HlConfig)
    c.dimS, c.dimA = dimS, dimA
    bounded = [1] * dimA if bounded is None else list(bounded)
    for i in range(dimA):
        c.bounded[i] = int(bounded[i])
    c.n_hidden = len(hidden)
    for i, hsz in enumerate(hidden):
        c.hidden[i] = int(hsz)
    c.nnFunc = FUNC[nnFunc] if isinstance(nnFunc, str) else int(nnFunc)
    c.adv_kind = adv_kind
    c.n_options = n_options if adv_kind == ADV_DISCRETE else 0
    c.nn_type, c.nnBPTTseq = nn_type, nnBPTTseq
    # conv: (input_width, input_height, input_features, kernels_num, filters_size, stride) as passed to
    # Communicator::setPreprocessingConv2d (Communicator.cpp:136-162)
    c.ERoldSeqFilter = {"oldest": 0, "default": 0, "farpolfrac": 1, "maxkldiv": 2, "minerror": 3}[ERoldSeqFilter] if isinstance(ERoldSeqFilter, str) else int(ERoldSeqFilter)
    c.dataSamplingAlgo = {"uniform": 0, "PERrank": 1, "PERerr": 2, "PERseq": 3}[dataSamplingAlgo] if isinstance(dataSamplingAlgo, str) else int(dataSamplingAlgo)
    c.returnsEstimator = RET[returnsEstimator] if isinstance(returnsEstimator, str) else int(returnsEstimator)
    c.nnOutputFunc = FUNC[nnOutputFunc] if isinstance(nnOutputFunc, str) else int(nnOutputFunc)
    c.n_encoder = len(encoder)
    c.encoder_rnn = int(encoder_rnn)
    for i, hsz in enumerate(encoder):
        c.encoder[i] = int(hsz)
    c.nAppendedObs, c.n_conv = nAppendedObs, len(conv)
    for i, (iw, ih, ic, kn, fs, st) in enumerate(conv):
        d = c.conv[i]
        d.inpFeatures, d.inpY, d.inpX, d.outFeatures = ic, ih, iw, kn
        d.filterx = d.filtery = fs
        d.stridex = d.stridey = st
        d.paddinx = d.paddiny = 0
        d.outY = (d.inpY - d.filterx + 2 * d.paddinx) // d.stridex + 1
        d.outX = (d.inpX - d.filtery + 2 * d.paddiny) // d.stridey + 1
    c.batchSize = batchSize
    c.maxTotObsNum, c.minTotObsNum = int(maxTotObsNum), int(minTotObsNum)
    c.gamma, c.lambda_, c.clipImpWeight, c.penalTol = gamma, lambda_, clipImpWeight, penalTol
    c.epsAnneal, c.learnrate, c.nnLambda = epsAnneal, learnrate, nnLambda
    c.explNoise, c.outWeightsPrefac = explNoise, outWeightsPrefac
    c.randSeed, c.n_ranks, c.rank, c.device_id = randSeed, n_ranks, rank, device_id
    c.episode_order, c.ref_threads = episode_order, ref_threads
    return c


class HlError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("%s (%d): %s" % (STATUS.get(status, "?"), status, msg))
        self.status = status


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class CApi:
    """Loads a shared library exporting <prefix>create, <prefix>step, ... (include/smarties_hip.h)."""

    def __init__(self, path, prefix="hl_"):
        if not os.path.exists(path):
            raise FileNotFoundError(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`; "
                "there is no CPU fallback for the HIP path" % path)
        self.path, self.prefix = path, prefix
        self.lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
        self._sig()

    def fn(self, name):
        return getattr(self.lib, self.prefix + name)

    def has(self, name):
        return hasattr(self.lib, self.prefix + name)

    def _sig(self):
        P, I32, I64 = C.c_void_p, C.c_int32, C.c_int64
        pf, pd = C.POINTER(C.c_float), C.POINTER(C.c_double)
        pi64, pu32, pi32 = C.POINTER(C.c_int64), C.POINTER(C.c_uint32), C.POINTER(C.c_int32)
        sig = {
            "create": (C.c_int, [C.POINTER(HlConfig), C.POINTER(P)]),
            "destroy": (C.c_int, [P]),
            "last_error": (C.c_char_p, [P]),
            "num_params": (I64, [P]), "num_outputs": (I32, [P]), "num_layers": (I32, [P]),
            "param_layout": (C.c_int, [P, pi64, pi64, pi64, pi64]),
            "init_weights": (C.c_int, [P]),
            "set_params": (C.c_int, [P, pf, pf, pf]), "get_params": (C.c_int, [P, pf, pf, pf]),
            "set_rng_state": (C.c_int, [P, pu32]), "get_rng_state": (C.c_int, [P, pu32]),
            "append_episode": (C.c_int, [P, I32, pf, pd, pd, pd, pf, pf, I32, I64]),
            "get_scaling": (C.c_int, [P, pf, pf, pf]), "set_scaling": (C.c_int, [P, pf, pf, pf]),
            "get_episode_field": (C.c_int, [P, I64, I32, pf, I32]),
            "get_episode_info": (C.c_int, [P, I64, pi64, pi32, pi32]),
            "get_episode_stats": (C.c_int, [P, I64, pf]),
            "initialize": (C.c_int, [P]),
            "initialize_begin": (C.c_int, [P]),
            "initialize_end": (C.c_int, [P]),
            "step": (C.c_int, [P, I32, pi64]),
            "step_begin": (C.c_int, [P, pi64]),
            "grad_exchange": (C.c_int, [P, pf, I32]),
            "counters_exchange": (C.c_int, [P, pi64, I32]),
            "moments_exchange": (C.c_int, [P, pd, I32]),
            "step_end": (C.c_int, [P]),
            "sync": (C.c_int, [P]),
            "prepare_steps": (C.c_int, [P, I32]),
            "set_tap": (C.c_int, [P, I32]),
            "readback": (C.c_int, [P, I32, C.c_void_p, I64]),
            "get_scalars": (C.c_int, [P, C.POINTER(HlScalars)]),
            "get_stats": (C.c_int, [P, C.POINTER(HlStats)]),
        }
        for name, (res, args) in sig.items():
            f = self.fn(name)
            f.restype, f.argtypes = res, args
        for name, (res, args) in {
            "comm_unique_id": (C.c_int, [C.POINTER(C.c_uint8)]),
            "comm_init": (C.c_int, [P, C.POINTER(C.c_uint8)]),
            "xchg_export": (C.c_int, [P, C.POINTER(C.c_uint8)]),
            "xchg_connect": (C.c_int, [P, C.POINTER(C.c_uint8)]),
            "get_counts": (C.c_int, [P, pi64, pi64, pi64, pi64, pi64]),
            "impweight_histogram": (C.c_int, [P, C.c_char_p, I32, pi64]),
            "timing_enable": (C.c_int, [P, I32]),
            "timing_get": (C.c_int, [P, C.c_char_p, pd, pi64]),
            "kernel_profile": (C.c_int, [P, I32, I32, pd]),
            "forward": (C.c_int, [P, I32, pf, pd]),
            "forward_sequence": (C.c_int, [P, I32, pf, pd]),
            "save": (C.c_int, [P, C.c_char_p]),
            "metrics": (C.c_int, [P, C.c_char_p, I32, C.c_char_p, I32]),
            "grad_stats": (C.c_int, [P, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
            "set_log_base": (C.c_int, [P, C.c_char_p]),
            "set_episode_log": (C.c_int, [P, C.c_char_p]),
            "save_memory": (C.c_int, [P, C.c_char_p, I32]),
            "restart_memory": (C.c_int, [P, C.c_char_p, I32]),
            "packed_episode_size": (C.c_int64, [P, I32]),
            "append_packed_episode": (C.c_int, [P, pf, I64]),
            "pack_episode": (C.c_int, [P, I64, pf, I64]),
            "restart": (C.c_int, [P, C.c_char_p]),
            "status_string": (C.c_char_p, [C.c_int]),
            "version": (C.c_int, []),
        }.items():
            if self.has(name):
                f = self.fn(name)
                f.restype, f.argtypes = res, args


def _ptr(a, ct):
    return None if a is None else a.ctypes.data_as(C.POINTER(ct))


_LIVE = weakref.WeakSet()


@atexit.register
def _close_all():
    # destroy device state while the HIP runtime is still alive (python atexit runs before the
    # runtime's own static destructors)
    for L in list(_LIVE):
        L.close()


class Learner:
    """Thin object wrapper over one hl_learner handle (names follow include/smarties_hip.h)."""

    def __init__(self, api, cfg):
        self.api, self.cfg = api, cfg
        self.h = C.c_void_p()
        rc = api.fn("create")(C.byref(cfg), C.byref(self.h))
        if rc:
            msg = (api.fn("last_error")(self.h) or b"").decode() if self.h else ""
            if self.h:
                api.fn("destroy")(self.h)
                self.h = C.c_void_p()
            raise HlError(rc, "hl_create failed: " + msg)
        _LIVE.add(self)
        self.nParams = api.fn("num_params")(self.h)
        self.nOut = api.fn("num_outputs")(self.h)
        self.dS, self.dA = cfg.dimS, cfg.dimA
        self.dIn = cfg.dimS * (1 + cfg.nAppendedObs)      # network input: the observed state and the appended past ones
        self.nOptions = cfg.n_options                      # discrete head: options of the one action variable
        self.polDim = cfg.n_options if cfg.n_options else 2 * cfg.dimA
        self.B = max(1, cfg.batchSize // cfg.n_ranks) if cfg.batchSize > 1 else cfg.batchSize

    def close(self):
        if self.h:
            self.api.fn("destroy")(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc:
            raise HlError(rc, (self.api.fn("last_error")(self.h) or b"").decode())

    # -- parameters -----------------------------------------------------------
    def layout(self):
        n = self.api.fn("num_layers")(self.h)
        arrs = [np.zeros(n, np.int64) for _ in range(4)]
        self._ck(self.api.fn("param_layout")(self.h, *[_ptr(a, C.c_int64) for a in arrs]))
        return dict(indW=arrs[0], nW=arrs[1], indB=arrs[2], nB=arrs[3])

    def init_weights(self):
        self._ck(self.api.fn("init_weights")(self.h))

    def set_params(self, w=None, m1=None, m2=None):
        w, m1, m2 = [None if a is None else _f32(a) for a in (w, m1, m2)]
        for a in (w, m1, m2):
            assert a is None or a.size == self.nParams
        self._ck(self.api.fn("set_params")(self.h, _ptr(w, C.c_float), _ptr(m1, C.c_float), _ptr(m2, C.c_float)))

    def get_params(self):
        w, m1, m2 = [np.zeros(self.nParams, np.float32) for _ in range(3)]
        self._ck(self.api.fn("get_params")(self.h, _ptr(w, C.c_float), _ptr(m1, C.c_float), _ptr(m2, C.c_float)))
        return w, m1, m2

    def set_rng_state(self, st):
        st = np.ascontiguousarray(st, dtype=np.uint32)
        assert st.size == 625
        self._ck(self.api.fn("set_rng_state")(self.h, _ptr(st, C.c_uint32)))

    def get_rng_state(self):
        st = np.zeros(625, np.uint32)
        self._ck(self.api.fn("get_rng_state")(self.h, _ptr(st, C.c_uint32)))
        return st

    # -- replay -----------------------------------------------------------------
    def append_episode(self, states, actions, mu, rewards, values, terminated, tag, advantages=None):
        states, values = _f32(states), _f32(values)
        actions, mu, rewards = _f64(actions), _f64(mu), _f64(rewards)
        n = rewards.size
        assert states.size == n * self.dS and actions.size == n * self.dA and mu.size == n * self.polDim
        adv = None if advantages is None else _f32(advantages)
        self._ck(self.api.fn("append_episode")(
            self.h, n, _ptr(states, C.c_float), _ptr(actions, C.c_double), _ptr(mu, C.c_double),
            _ptr(rewards, C.c_double), _ptr(values, C.c_float), _ptr(adv, C.c_float), int(terminated), int(tag)))

    def get_scaling(self):
        m, s, r = np.zeros(self.dS, np.float32), np.zeros(self.dS, np.float32), np.zeros(3, np.float32)
        self._ck(self.api.fn("get_scaling")(self.h, _ptr(m, C.c_float), _ptr(s, C.c_float), _ptr(r, C.c_float)))
        return m, s, r

    def set_scaling(self, m, s, r):
        m, s, r = _f32(m), _f32(s), _f32(r)
        self._ck(self.api.fn("set_scaling")(self.h, _ptr(m, C.c_float), _ptr(s, C.c_float), _ptr(r, C.c_float)))

    def episode_info(self, pos):
        tag, n, term = C.c_int64(), C.c_int32(), C.c_int32()
        self._ck(self.api.fn("get_episode_info")(self.h, pos, C.byref(tag), C.byref(n), C.byref(term)))
        return tag.value, n.value, term.value

    def episode_stats(self, pos):
        """totR, avgKLDivergence, fracFarPolSteps, avgSquaredErr, maxAbsError, sumSquaredQ, sumQ, maxQ, minQ (Episode.h:82-85)"""
        out = np.zeros(9, np.float32)
        self._ck(self.api.fn("get_episode_stats")(self.h, pos, _ptr(out, C.c_float)))
        return out

    def episode_field(self, pos, field):
        _, n, _ = self.episode_info(pos)
        out = np.zeros(n, np.float32)
        self._ck(self.api.fn("get_episode_field")(self.h, pos, field, _ptr(out, C.c_float), n))
        return out

    # -- training -----------------------------------------------------------------
    def initialize(self):
        self._ck(self.api.fn("initialize")(self.h))

    def initialize_begin(self):
        """First half of `initialize` (host-exchange mode: sum `counters_fetch` / `moments_fetch` over the replicas, store, then `initialize_end`)."""
        self._ck(self.api.fn("initialize_begin")(self.h))

    def initialize_end(self):
        self._ck(self.api.fn("initialize_end")(self.h))

    def step(self, n=1, flat=None):
        if flat is not None:
            flat = np.ascontiguousarray(flat, dtype=np.int64)
            assert flat.size == n * self.B
        self._ck(self.api.fn("step")(self.h, n, _ptr(flat, C.c_int64)))

    def step_begin(self, flat=None):
        if flat is not None:
            flat = np.ascontiguousarray(flat, dtype=np.int64)
        self._ck(self.api.fn("step_begin")(self.h, _ptr(flat, C.c_int64)))

    def step_end(self):
        self._ck(self.api.fn("step_end")(self.h))

    def grad_fetch(self):
        g = np.zeros(self.nParams, np.float32)
        self._ck(self.api.fn("grad_exchange")(self.h, _ptr(g, C.c_float), 0))
        return g

    def grad_store(self, g):
        g = _f32(g)
        self._ck(self.api.fn("grad_exchange")(self.h, _ptr(g, C.c_float), 1))

    def counters_fetch(self):
        c = np.zeros(4, np.int64)
        self._ck(self.api.fn("counters_exchange")(self.h, _ptr(c, C.c_int64), 0))
        return c

    def counters_store(self, c):
        c = np.ascontiguousarray(c, dtype=np.int64)
        self._ck(self.api.fn("counters_exchange")(self.h, _ptr(c, C.c_int64), 1))

    def moments_fetch(self):
        m = np.zeros(2 * self.dS + 3, np.float64)
        rc = self.api.fn("moments_exchange")(self.h, _ptr(m, C.c_double), 0)
        return None if rc else m

    def moments_store(self, m):
        m = _f64(m)
        self._ck(self.api.fn("moments_exchange")(self.h, _ptr(m, C.c_double), 1))

    def append_packed_episode(self, data):
        """Episode in the reference's wire format (Episode::packEpisode)."""
        d = np.ascontiguousarray(data, dtype=np.float32)
        self._ck(self.api.fn("append_packed_episode")(self.h, _ptr(d, C.c_float), d.size))

    def pack_episode(self, pos):
        _, n, _ = self.episode_info(pos)
        out = np.zeros(int(self.api.fn("packed_episode_size")(self.h, n)), np.float32)
        self._ck(self.api.fn("pack_episode")(self.h, pos, _ptr(out, C.c_float), out.size))
        return out

    def forward_sequence(self, states):
        """network outputs for the last of the given consecutive raw states (recurrent nets: forwarded from a zero state)"""
        states = _f32(states).reshape(-1, self.dS)
        out = np.zeros(self.nOut, np.float64)
        self._ck(self.api.fn("forward_sequence")(self.h, states.shape[0], _ptr(states, C.c_float), _ptr(out, C.c_double)))
        return out

    def metrics(self):
        """(header, line) of <learner>_stats.txt as Learner::logStats formats them."""
        hd, ln = C.create_string_buffer(1024), C.create_string_buffer(1024)
        self._ck(self.api.fn("metrics")(self.h, hd, 1024, ln, 1024))
        return hd.value.decode(), ln.value.decode()

    def grad_stats(self):
        """mean, RMS of every network output's gradient over the last minibatch (StatsTracker)."""
        m, r = np.zeros(self.nOut, np.float64), np.zeros(self.nOut, np.float64)
        self._ck(self.api.fn("grad_stats")(self.h, _ptr(m, C.c_double), _ptr(r, C.c_double)))
        return m, r

    def set_log_base(self, base):
        self._ck(self.api.fn("set_log_base")(self.h, base.encode() if base else None))

    def save_memory(self, base, rank=0):
        """Replay memory + ReF-ER state in the reference's files (MemoryBuffer::save)."""
        self._ck(self.api.fn("save_memory")(self.h, str(base).encode(), rank))

    def restart_memory(self, base, rank=0):
        self._ck(self.api.fn("restart_memory")(self.h, str(base).encode(), rank))

    def save(self, base):
        """Checkpoint in the reference's format: <base>_weights.raw, _1stMom.raw, _2ndMom.raw."""
        self._ck(self.api.fn("save")(self.h, str(base).encode()))

    def restart(self, base):
        self._ck(self.api.fn("restart")(self.h, str(base).encode()))

    def forward(self, states):
        """Network outputs [n][nOut] (float64) for raw states [n][dimS] with the current weights."""
        st = np.ascontiguousarray(states, dtype=np.float32).reshape(-1, self.dIn)
        out = np.zeros((st.shape[0], self.nOut), np.float64)
        self._ck(self.api.fn("forward")(self.h, st.shape[0], _ptr(st, C.c_float), _ptr(out, C.c_double)))
        return out

    def set_episode_log(self, path):
        self._ck(self.api.fn("set_episode_log")(self.h, str(path).encode() if path else None))

    def impweight_histogram(self):
        """(text block as Learner::logStats prints it, 81 bin counts) of the stored importance weights"""
        buf = C.create_string_buffer(8192)
        cnt = np.zeros(81, np.int64)
        self._ck(self.api.fn("impweight_histogram")(self.h, buf, 8192, _ptr(cnt, C.c_int64)))
        return buf.value.decode(), cnt

    def counts(self):
        """(nStoredSteps, nStoredEps, nGradSteps, nSeenSteps, nSeenEps): host-side counters, no device wait"""
        v = [C.c_int64() for _ in range(5)]
        self._ck(self.api.fn("get_counts")(self.h, *[C.byref(x) for x in v]))
        return tuple(int(x.value) for x in v)

    def sync(self):
        self._ck(self.api.fn("sync")(self.h))

    def prepare_steps(self, n):
        """announce calls of n steps (hl_prepare_steps): the graph of exactly n steps is captured now"""
        self._ck(self.api.fn("prepare_steps")(self.h, int(n)))

    # -- inspection -----------------------------------------------------------------
    def set_tap(self, on=True):
        self._ck(self.api.fn("set_tap")(self.h, int(on)))

    def readback(self, what):
        B, nO = self.B, self.nOut
        shape, dt = {
            TAP_FLAT: ((B,), np.int64), TAP_EPISODE: ((B,), np.int64), TAP_TSTEP: ((B,), np.int64),
            TAP_TAG: ((B,), np.int64), TAP_STATE: ((B, self.dIn), np.float32),
            TAP_OUTPUT: ((B, nO), np.float64), TAP_OUTGRAD: ((B, nO), np.float64),
            TAP_RHO: ((B,), np.float64), TAP_DKL: ((B,), np.float64), TAP_DELTAQ: ((B,), np.float64),
            TAP_FAR: ((B,), np.uint8), TAP_GRADSUM: ((self.nParams,), np.float32)}[what]
        out = np.zeros(shape, dt)
        self._ck(self.api.fn("readback")(self.h, what, out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    def scalars(self):
        s = HlScalars()
        self._ck(self.api.fn("get_scalars")(self.h, C.byref(s)))
        return s

    def stats(self):
        s = HlStats()
        self._ck(self.api.fn("get_stats")(self.h, C.byref(s)))
        return s

    # -- RCCL / timing (HIP library only) --------------------------------------------
    def comm_init(self, unique_id):
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        self._ck(self.api.fn("comm_init")(self.h, buf))

    def xchg_export(self):
        """Handle (XCHG_HANDLE_BYTES bytes) of this replica's exchange window: gather all replicas' in rank order."""
        buf = (C.c_uint8 * XCHG_HANDLE_BYTES)()
        self._ck(self.api.fn("xchg_export")(self.h, buf))
        return bytes(buf)

    def xchg_connect(self, handles):
        raw = b"".join(bytes(hd) for hd in handles)
        buf = (C.c_uint8 * len(raw)).from_buffer_copy(raw)
        self._ck(self.api.fn("xchg_connect")(self.h, buf))

    def timing_enable(self, on=True):
        self._ck(self.api.fn("timing_enable")(self.h, int(on)))

    def timing_get(self, kernel):
        ms, n = C.c_double(), C.c_int64()
        self._ck(self.api.fn("timing_get")(self.h, kernel.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def kernel_profile(self, which, reps=200):
        """Average microseconds per launch of one kernel of the step (see hl_kernel_profile)."""
        us = C.c_double()
        self._ck(self.api.fn("kernel_profile")(self.h, int(which), int(reps), C.byref(us)))
        return us.value


_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB = os.path.join(_HERE, "libsmarties_hip.so")


_hip_api = None


def load_hip():
    """The product library.  Raises if it has not been built (no CPU fallback).

    PyTorch-ROCm bundles its own copy of the HIP runtime / RCCL with the same SONAMEs as the system
    ROCm.  Whichever copy is mapped first serves the whole process, and torch breaks when it finds
    the system copy already resident, so torch (when installed) is imported BEFORE the library is
    dlopen-ed; libsmarties_hip.so then binds to the runtime torch brought in.
    """
    global _hip_api
    if _hip_api is None:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _hip_api = CApi(HIP_LIB, "hl_")
    return _hip_api
