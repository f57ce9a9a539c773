"""Host-side replica protocol for N > 1 learners (one process per GPU).

Reference behaviour: with `nLearners` > 1 the global batch and the replay budget are
split over the replicas (Settings/HyperParameters.cpp:186-197), every gradient step all-reduces the
fp32 gradient sum (Network/Optimizer.cpp:110-132), the four replay counters
(Utils/DelayedReductor.cpp:53-83, read by MemoryProcessing.cpp:46-92) and -- every 1000th step -- the
2*dS+3 reward/state moments (MemoryProcessing.cpp:139-150).

The product does those exchanges on the device with RCCL inside `hl_step` (learner.cpp,
`hl_comm_init`).  This module is the SAME protocol driven from the host through the split-step
entry points (`hl_step_begin` / `hl_*_exchange` / `hl_step_end`) and any `torch.distributed`
backend; it exists so that an embedding which already owns an MPI / gloo communicator (the
reference does) can keep it, and so that the N > 1 order of operations is covered by
world_size-2 `gloo` tests on machines without a GPU.
"""
import numpy as np


def init_replica_weights(L, dist, src=0, group=None):
    """Replica 0's initial weights everywhere (reference: Network broadcast in
    Learner_approximator::initializeApproximators -> Optimizer MPI_Bcast, Core/Optimizer.cpp:60-70)."""
    import torch
    w, m1, m2 = L.get_params()
    t = torch.from_numpy(w)
    dist.broadcast(t, src, group=group)
    L.set_params(w, m1, m2)


def initialize_host_exchange(L, dist, group=None):
    """Learner::initializeLearner of one replica (Learners/Learner.cpp:47-72): the start-up counters and reward / state moments are
    accurate reductions over the learners (DelayedReductor::get(true)), every replica starts from the global statistics.
    Returns the summed (counters, moments): what a run in the reference's stale timing starts from (step_host_exchange)."""
    import torch
    L.initialize_begin()
    c = np.asarray(L.counters_fetch(), dtype=np.int64)
    dist.all_reduce(torch.from_numpy(c), op=dist.ReduceOp.SUM, group=group)
    L.counters_store(c)
    m = L.moments_fetch()
    dist.all_reduce(torch.from_numpy(m), op=dist.ReduceOp.SUM, group=group)
    L.moments_store(m)
    L.initialize_end()
    return c.copy(), m.copy()


def step_host_exchange(L, dist, n_steps=1, flat=None, group=None, stale=None):
    """`n_steps` gradient steps of one replica; collectives through `dist` (torch.distributed; `group`: a process group
    that takes host tensors, e.g. a gloo group next to an nccl default group).

    `stale`: None = every step uses ITS OWN global counter / moment sums (what the library's device exchange does; the reference with
    every MPI_Test finding its reduction complete).  A `[counters, moments]` list (start it with what initialize_host_exchange
    returned) = the reference's other timing, one reduction behind (Utils/DelayedReductor.cpp:34-60 when no poll completes: step k
    updates beta from the sums of step k - 1, a 1000th step's statistics from the last completed moments); the list is updated in
    place and carries over to the next call.  Both timings are pinned against the compiled reference (tests/test_two_rank_reference.py)."""
    import torch
    for s in range(n_steps):
        L.step_begin(None if flat is None else flat[s])
        g = L.grad_fetch()
        dist.all_reduce(torch.from_numpy(g), op=dist.ReduceOp.SUM, group=group)
        L.grad_store(g)
        m = L.moments_fetch()          # None unless this is a 1000th step (same on every replica)
        if m is not None:
            dist.all_reduce(torch.from_numpy(m), op=dist.ReduceOp.SUM, group=group)
            if stale is not None:
                L.moments_store(stale[1]); stale[1] = m.copy()
            else:
                L.moments_store(m)
        c = np.asarray(L.counters_fetch(), dtype=np.int64)
        dist.all_reduce(torch.from_numpy(c), op=dist.ReduceOp.SUM, group=group)
        if stale is not None:
            L.counters_store(stale[0]); stale[0] = c.copy()
        else:
            L.counters_store(c)
        L.step_end()
