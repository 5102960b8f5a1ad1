"""Multi-GPU sharding of a swarm of independent agents (one process per GPU, torch.distributed).

Agents never read each other's state (reference: src/extended_kalman_filter.cpp:64-78,
src/lqr_speed_steer_control.cpp:85-151, src/model_predictive_control.cpp:255-346), so the data
path needs no collective: rank r owns the contiguous agent range shard_range(n, r, world).  The
only exchange the system adds is the concatenation of per-rank results (estimated trajectories /
final states) — one all-gather (RCCL over xGMI on GPUs; gloo in the CPU tests).
"""
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous balanced partition: the first n % world ranks get one extra agent."""
    base, rem = divmod(int(n), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sizes(n, world):
    return [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]


class _Done:
    """A finished collective (what the host-staged path hands back in place of an async Work)."""
    def wait(self):
        return True


def _all_gather_into(out, local, group=None, async_op=False):
    """dist.all_gather_into_tensor, plus one portability case: device tensors on a gloo group (several ranks sharing one GPU in
    bench.py's --oversubscribe dry run — RCCL refuses that, gloo has no device all-gather) are staged through the host,
    synchronously.  RCCL groups and CPU tensors take the collective as it is."""
    if local.is_cuda and dist.get_backend(group) == "gloo":
        oc = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(oc, local.contiguous().cpu(), group=group)
        out.copy_(oc)
        return _Done() if async_op else None
    return dist.all_gather_into_tensor(out, local, group=group, async_op=async_op)


def gather_agents(local, n_total, group=None):
    """All-gather per-agent rows.  `local` is [n_local, ...] on this rank (agent-major); returns
    [n_total, ...] in global agent order on every rank.  Equal shards use one
    all_gather_into_tensor (a single RCCL all-gather); ragged shards pad to the largest shard."""
    world = dist.get_world_size(group)
    sizes = shard_sizes(n_total, world)
    assert local.shape[0] == sizes[dist.get_rank(group)]
    tail = tuple(local.shape[1:])
    if len(set(sizes)) == 1:
        out = torch.empty((n_total,) + tail, dtype=local.dtype, device=local.device)
        _all_gather_into(out, local.contiguous(), group)
        return out
    m = max(sizes)
    padded = torch.zeros((m,) + tail, dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    buf = torch.empty((world * m,) + tail, dtype=local.dtype, device=local.device)
    _all_gather_into(buf, padded, group)
    return torch.cat([buf[r * m: r * m + sizes[r]] for r in range(world)], dim=0)


def gather_time_major(local, n_total, group=None):
    """All-gather a time-major history [T, n_local, C] into [T, n_total, C] (global agent order).
    The wire format is the rank-major block layout [world][T][n_local][C] that a single
    all-gather produces; the permutation back to time-major is a local copy."""
    world = dist.get_world_size(group)
    sizes = shard_sizes(n_total, world)
    T, nl, Cc = local.shape
    if len(set(sizes)) != 1:
        return gather_agents(local.transpose(0, 1).contiguous(), n_total, group).transpose(0, 1).contiguous()
    buf = torch.empty((world * T, nl, Cc), dtype=local.dtype, device=local.device)   # rank-major blocks
    _all_gather_into(buf, local.contiguous(), group)
    return buf.view(world, T, nl, Cc).permute(1, 0, 2, 3).reshape(T, n_total, Cc).contiguous()


class RingGather:
    """Per-step all-gather of the agents' final estimates, overlapped with the following launches: step k's collective is issued
    asynchronously (its own stream on GPUs) into buffer k % ring, and the caller's stream waits for the collectives only once per
    lap of the ring — `begin_step(k)` before anything of step k that touches state buffer k % ring is enqueued.  The caller keeps
    `ring` state buffers too, so that a collective never reads a buffer a later launch is resetting.  (A cross-stream wait in
    front of every launch costs ~30 us of queue bubbles per step on MI355X; see bench.py.)  Equal shards only."""

    def __init__(self, n_total, tail, ring, device, dtype=torch.float32, group=None):
        self.ring, self.group = int(ring), group
        self.out = [torch.empty((n_total,) + tuple(tail), dtype=dtype, device=device) for _ in range(self.ring)]
        # Every collective still in flight.  RCCL runs a group's collectives in issue order on one stream, so waiting for the
        # last one would do there; gloo hands them to a pool of worker threads that may finish out of order (the world-2 CPU
        # test caught a lap whose first gather was still running when its buffer was reused) — so all of them are waited for.
        self.pending = []

    def slot(self, step):
        return step % self.ring

    def begin_step(self, step):
        if step % self.ring == 0:
            self.wait()             # stream-level on GPUs: every gather of the previous lap is done

    def gather(self, step, local):
        out = self.out[step % self.ring]
        self.pending.append(_all_gather_into(out, local, self.group, async_op=True))
        return out

    def wait(self):
        for w in self.pending:
            w.wait()
        self.pending = []


class ChunkedTrajectoryGather:
    """The trajectory concat of north_star, chunked and overlapped: the fused T-step launch is cut into `chunks` launches of
    T/chunks steps (the filter state carries over in place, results bit-identical to one launch) and the all-gather of chunk
    k's history runs on the collective's own stream while chunk k+1 computes.

    Wire/buffer layout (no transpose kernel; a consumer iterates shard by shard):
        gathered[c][r][t][a][:] = estimate of agent (shard r's first agent + a) after step c*Tc + t
    i.e. shape [chunks, world, Tc, n_local, C] — global step = c*Tc + t, global agent = r*n_local + a.  `time_major()` gives the
    [T, n_total, C] view as a copy for consumers that want it.  Equal shards only (the bench's weak-scaling layout)."""

    def __init__(self, T, n_local, C, chunks, device, dtype=torch.float32, group=None):
        assert T % chunks == 0, "the chunk count must divide the number of steps"
        self.T, self.nl, self.C, self.chunks, self.Tc, self.group = T, n_local, C, chunks, T // chunks, group
        self.world = dist.get_world_size(group)
        self.local = torch.empty((chunks, self.Tc, n_local, C), dtype=dtype, device=device)
        self.gathered = torch.empty((chunks, self.world, self.Tc, n_local, C), dtype=dtype, device=device)
        self.pending = []

    def run(self, launch):
        """launch(c, t0, t1, hist): enqueue steps [t0, t1) writing their history into hist ([Tc, n_local, C])."""
        self.wait()                                   # the previous pass's gathers still read self.local
        for c in range(self.chunks):
            launch(c, c * self.Tc, (c + 1) * self.Tc, self.local[c])
            self.pending.append(_all_gather_into(self.gathered[c].view(self.world * self.Tc, self.nl, self.C), self.local[c],
                                                 self.group, async_op=True))
        return self

    def wait(self):
        for w in self.pending:
            w.wait()
        self.pending = []

    def bytes_received_per_rank(self):
        return self.gathered.numel() * self.gathered.element_size()

    def time_major(self):
        self.wait()
        return self.gathered.permute(0, 2, 1, 3, 4).reshape(self.T, self.world * self.nl, self.C).contiguous()


def _env_int(name):
    """The integer value of an environment variable, or None (unset, empty, or not a number — said once, not raised)."""
    import os
    cur = os.environ.get(name)
    if cur is None or not cur.strip():
        return None
    try:
        return int(cur)
    except ValueError:
        import warnings
        warnings.warn(f"{name}={cur!r} is not a number: ignored")
        return None


def want_hw_queues(streams):
    """The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); two streams that share a queue
    run their kernels one after the other.  A pipelined round — one launch stream + `depth` planner streams (+ RCCL's) — needs a queue per
    stream: measured on one MI355X at depth 6, 0.74 ms per round with 4 queues, 0.49 with 8 (profiles/r05/swarm_hw_queues.txt).  The
    variable is read when the runtime initialises the device, so this only has an effect before the process first touches the GPU: a
    process-start call (bench.py, scripts/swarm_bench.py and tests/swarm_checks.py export the variable themselves; SwarmShard calls this
    as its first statement and otherwise only CHECKS).  Returns the queue count in effect for a device initialised from now on — None
    if the device was already initialised without the variable (the caller then runs on the runtime's default).  A value the user set
    is never lowered, and it is raised only while that still has an effect — with a warning, since it overrides a deliberate setting."""
    import os
    want = max(8, int(streams))
    cur = _env_int("GPU_MAX_HW_QUEUES")
    if cur is not None and cur >= want:
        return cur
    if torch.cuda.is_available() and torch.cuda.is_initialized():
        return cur
    if cur is not None:
        import warnings
        warnings.warn(f"GPU_MAX_HW_QUEUES={cur} raised to {want}: {int(streams)} streams need a hardware queue each")
    os.environ["GPU_MAX_HW_QUEUES"] = str(want)
    return want


_planner_stream_cache = {}


def planner_streams(device, depth):
    """The first `depth` planner streams of a device, created once per process and shared by every MixedSwarmRound on it.  Each hardware
    queue a solver launch has ever run on keeps a scratch (private-memory) reservation sized for that kernel — the MPC solve needs
    ~0.4 MB per wave — and the reservations of all queues come out of one pool: a process that built one round object after another,
    each with fresh streams (torch hands out pool streams round-robin, so they land on different hardware queues), died on the fifth
    with HSA_STATUS_ERROR_OUT_OF_RESOURCES once all 16 queues held one (round 5, gpurun_out/r05m).  With shared streams a process touches
    depth + 1 queues with the solver however many rounds it builds."""
    key = str(torch.device(device))
    have = _planner_stream_cache.setdefault(key, [])
    while len(have) < depth:
        have.append(torch.cuda.Stream(device=device))
    return have[:depth]


def _with_slot(plan_launch, self_depth=1):
    """plan_launch(est, slot) is the signature since round 5; a callable written against the earlier plan_launch(est) is accepted while
    a single slot exists (depth 1: nothing else is in flight, so it cannot write into another round's buffers) and refused with a clear
    message otherwise — instead of a TypeError from inside run()."""
    import inspect
    try:
        params = [p for p in inspect.signature(plan_launch).parameters.values()
                  if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD, p.VAR_POSITIONAL)]
    except (TypeError, ValueError):
        return plan_launch
    if any(p.kind == p.VAR_POSITIONAL for p in params) or len(params) >= 2:
        return plan_launch
    if max(1, int(self_depth)) > 1:
        raise TypeError("MixedSwarmRound: plan_launch must take (est, slot) when depth > 1 — the planners of the other slots are still "
                        "running, every buffer a launch writes has to belong to its slot")
    return lambda est, slot: plan_launch(est)


class MixedSwarmRound:
    """One round of BASELINE.json configs[4] on this rank's shard of a mixed EKF + MPC swarm (bench.py and scripts/swarm_bench.py run it
    on the GPUs, tests/test_swarm_gpu.py checks it there against the oracle, tests/test_dist_cpu.py with gloo and the CPU oracle
    standing in for the launches):

      1. the shard's vehicles run T fused EKF steps — when the trajectories are gathered over more than one rank, cut into `chunks`
         launches whose histories are all-gathered one by one while the next chunk computes (ChunkedTrajectoryGather); ONE launch
         otherwise (nothing to overlap: every extra launch pays the kernel's start-up and drain again — round 4 ran the 1-GPU shard at
         0.285 of 8 TB/s in four launches of 25 steps).  `gather="final"` gathers the final estimates only, "none" nothing;
      2. every `plan_every`-th vehicle plans from its final estimate: plan_launch(est, slot) — on the slot's own stream on GPUs.  The
         round keeps `depth` planner slots (estimate buffer, stream, events) and uses them round-robin, so the planners of rounds
         r - depth + 1 .. r are in flight together and overlap the EKF launches of the following rounds: one planner launch is a
         latency chain (its slowest agent's sweeps) on a quarter of the SIMDs, `depth` of them fill the chip.  A slot's
         estimate buffer is refilled only after the planners that read it have finished (an event, not a stream join).

    ekf_launch(c, t0, t1, hist): enqueue EKF steps [t0, t1) of this shard, history into hist [t1 - t0, n_local, C]; it owns the
    filter state and must reset it when c == 0.  final_state(): the shard's [n_local, C] estimate after the last chunk.
    plan_launch(est, slot): enqueue the planners on est [n_plan, C] (a buffer owned by this object); every buffer it writes must
    belong to `slot` (0 .. depth-1) — the planners of the other slots are running.  Its return value is plans_of(round).  (The one-argument
    form of rounds 1-4 is still accepted at depth 1.)

    Things a caller should know: `chunks` is a request — `self.chunks` is the count in effect (1 unless trajectories are gathered over
    more than one rank; `requested_chunks` keeps the argument); `plans` / `final` are valid after wait(); the planner streams are
    per-device and shared by every round object of the process (planner_streams), so two live rounds on one device queue their
    planners on the same streams, in issue order."""

    def __init__(self, n_local, T, C, chunks, plan_every, device, ekf_launch, final_state, plan_launch, gather="traj", n_total=None, group=None,
                 depth=1):
        self.nl, self.T, self.C, self.every, self.gather_kind, self.group = n_local, T, C, int(plan_every), gather, group
        self.ekf_launch, self.final_state, self.plan_launch = ekf_launch, final_state, _with_slot(plan_launch, self_depth=depth)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.n_total = n_total if n_total is not None else n_local * self.world
        self.cuda = torch.device(device).type == "cuda"
        self.depth = max(1, int(depth))
        assert T % chunks == 0, "the chunk count must divide the number of steps"
        gathered = gather == "traj" and dist.is_initialized()
        # chunked only where a gather overlaps the chunks (more than one rank): see the docstring
        self.requested_chunks = int(chunks)
        self.chunks = chunks if (gathered and self.world > 1) else 1      # the EFFECTIVE count ekf_launch sees
        self.cg = ChunkedTrajectoryGather(T, n_local, C, self.chunks, device, group=group) if gathered else None
        self.local_hist = None if self.cg is not None else torch.empty((self.chunks, T // self.chunks, n_local, C), dtype=torch.float32, device=device)
        n_plan = (n_local + self.every - 1) // self.every
        self.est = [torch.empty((n_plan, C), dtype=torch.float32, device=device) for _ in range(self.depth)]
        self.plan_streams = planner_streams(device, self.depth) if self.cuda else None
        self.ekf_done = [torch.cuda.Event() for _ in range(self.depth)] if self.cuda else None
        self.plan_done = [torch.cuda.Event() for _ in range(self.depth)] if self.cuda else None
        self._plans = [None] * self.depth
        self.round = 0
        # gather="final": the all-gather of round r's final estimates runs asynchronously (RCCL's stream on GPUs) out of a snapshot of the
        # state — the next round resets it — into the slot's own buffer, and is waited for only when the slot comes round again (or in
        # wait()): a synchronous gather in front of every round would stall the launch stream once per round (one rank, RCCL: 0.79 ms per
        # round against 0.47 without any gather).  Equal shards; ragged shards fall back to the synchronous gather_agents.
        self._final_async = gather == "final" and dist.is_initialized() and len(set(shard_sizes(self.n_total, self.world))) == 1
        if self._final_async:
            self._final_src = [torch.empty((n_local, C), dtype=torch.float32, device=device) for _ in range(self.depth)]
            self._final_out = [torch.empty((self.n_total, C), dtype=torch.float32, device=device) for _ in range(self.depth)]
            self._final_work = [None] * self.depth
        self._final_sync = None

    @property
    def final(self):
        """[n_total, C]: the gathered final estimates of the most recent round (gather="final"; valid after wait())."""
        if not self._final_async:
            return self._final_sync
        return self._final_out[(self.round - 1) % self.depth] if self.round else None

    @property
    def plans(self):
        """The plans of the most recent round (valid after wait())."""
        return self._plans[(self.round - 1) % self.depth] if self.round else None

    def plans_of(self, rnd):
        """The plans of round `rnd` (0-based; one of the last `depth` rounds; valid after wait())."""
        assert max(0, self.round - self.depth) <= rnd < self.round, "that round's slot has been reused"
        return self._plans[rnd % self.depth]

    def run(self):
        main = torch.cuda.current_stream() if self.cuda else None
        slot = self.round % self.depth
        if self.cg is not None:
            self.cg.run(self.ekf_launch)
        else:
            Tc = self.T // self.chunks
            for c in range(self.chunks):
                self.ekf_launch(c, c * Tc, (c + 1) * Tc, self.local_hist[c])
        if self.cuda and self.round >= self.depth:
            main.wait_event(self.plan_done[slot])         # the planners of round r - depth still read self.est[slot]
        x = self.final_state()
        self.est[slot].copy_(x[:: self.every])
        if self.cuda:
            self.ekf_done[slot].record(main)
            ps = self.plan_streams[slot]
            with torch.cuda.stream(ps):
                ps.wait_event(self.ekf_done[slot])
                self._plans[slot] = self.plan_launch(self.est[slot], slot)
                self.plan_done[slot].record(ps)
        else:
            self._plans[slot] = self.plan_launch(self.est[slot], slot)
        if self._final_async:
            if self._final_work[slot] is not None:
                self._final_work[slot].wait()               # round r - depth's gather has left this slot's buffers
            self._final_src[slot].copy_(x)
            self._final_work[slot] = _all_gather_into(self._final_out[slot], self._final_src[slot], self.group, async_op=True)
        elif self.gather_kind == "final" and dist.is_initialized():
            self._final_sync = gather_agents(x, self.n_total, self.group)
        self.round += 1
        return self

    def wait(self):
        if self.cg is not None:
            self.cg.wait()
        if self._final_async:
            for k, w in enumerate(self._final_work):
                if w is not None:
                    w.wait()
                    self._final_work[k] = None
        if self.cuda:
            for ps in self.plan_streams:
                torch.cuda.current_stream().wait_stream(ps)

    def gathered_bytes_per_rank(self):
        if not dist.is_initialized() or self.world == 1:
            return 0
        return {"traj": 4 * self.C * self.T * self.n_total, "final": 4 * self.C * self.n_total, "none": 0}[self.gather_kind]

    def trajectory_time_major(self):
        """[T, n_total, C] in global agent order (a copy), from the chunked gather; without one (a single process, or a round
        that gathers final estimates / nothing) the shard's own history [T, n_local, C]."""
        if self.cg is None:
            return self.local_hist.reshape(self.T, self.nl, self.C).clone()
        return self.cg.time_major()


class SwarmShard:
    """This rank's shard of BASELINE.json configs[4] on the engine's kernels — what bench.py (`extra.swarm_configs4` /
    `multi_gpu.swarm_configs4`), scripts/swarm_bench.py and tests/test_swarm_gpu.py run.  Per round (MixedSwarmRound):

      1. every vehicle of the shard runs T fused EKF steps from its start state (crx_ekf_run_batch_dev, xEst history out) — the body of
         the reference's main loop, /root/reference/src/extended_kalman_filter.cpp:171-188;
      2. every `plan_every`-th vehicle plans from its final estimate at the commanded speed: nearest course point, calc_ref_trajectory,
         mpc_solve over Tm knots — one pass of mpc_simulation's loop, /root/reference/src/model_predictive_control.cpp:371-385.  (The
         estimated pose, not the filter's 4th state: that one integrates the noisy velocity input every step — F(3,3) = 1 and B(3,0) = 1,
         extended_kalman_filter.cpp:27,34 — a random walk, not a speed estimate.)

    Every input is keyed by the GLOBAL agent id (start poses drawn for the whole swarm and sliced; Philox draws counted from
    rank * n), so a shard computes what the whole swarm would have computed for its agents.  `input_sets` independent sets of
    measurements (seed + set index) are used round-robin — the parity test gives consecutive rounds different inputs, so that a
    planner launch reading another round's estimates could not go unnoticed.

    course: the (cx, cy, cyaw, ck, sp) float32 arrays of the shared course; Q, R: the filter's noise matrices (column-major).
    mpc_fn(est, xref, Tm, out): the planner launch (default cpprobotics_amd.mpc_solve)."""

    def __init__(self, n, T, course, Q, R, device, rank=0, world=1, Tm=21, plan_every=8, depth=6, chunks=4, gather="traj", seed=99, v_cmd=2.5,
                 input_sets=1, mpc_fn=None, group=None, record_ekf_events=False):
        # before anything touches the device (after the first tensor it can no longer set the variable)
        self.hw_queues = want_hw_queues(depth + 2) if torch.device(device).type == "cuda" else None
        import numpy as np

        import cpprobotics_amd as crx
        from .ekf import _qr
        self.crx, self.n, self.T, self.Tm, self.dev, self.rank, self.world, self.v_cmd = crx, n, T, Tm, device, rank, world, float(v_cmd)
        self.n_total, self.n_plan, self.every = n * world, (n + plan_every - 1) // plan_every, plan_every
        self.q, self.r = _qr(Q, R)
        self.dc = crx.Course.from_numpy(course, device=device)
        # the planner launches of `depth` rounds and the EKF launches share the GPU: say so (crx_mpc_params.shared_gpu, a hint that
        # selects the solve's low-traffic form; the answer is the same bits)
        from .mpc import default_params
        self.mpc_params = default_params()
        self.mpc_params.shared_gpu = 1 if depth > 1 else 0
        self.mpc_fn = mpc_fn if mpc_fn is not None else (lambda est, xref, Tm_, out: crx.mpc_solve(est, xref, Tm_, params=self.mpc_params, out=out))
        ci = np.random.default_rng(seed).integers(0, len(course[0]) - 30, self.n_total)[rank * n:(rank + 1) * n]
        self.start_index = ci
        cit = torch.from_numpy(ci).to(device)
        cx, cy, cyaw = (torch.from_numpy(np.ascontiguousarray(a)).to(device) for a in course[:3])
        self.x0 = torch.stack([cx[cit], cy[cit], cyaw[cit], torch.full((n,), self.v_cmd, device=device)], dim=1).contiguous()
        self.P0 = torch.eye(4, device=device).reshape(1, 16).repeat(n, 1).contiguous()
        u_true = torch.zeros((n, 2), dtype=torch.float32, device=device)               # (accel, yaw rate): the vehicles cruise
        self.z, self.ud = [], []
        for s in range(input_sets):
            w = crx.normal_draws(n, T, agent0=rank * n, seed=seed + s, device=device)
            z, ud = crx.ekf_simulate_inputs(u_true, self.x0.clone(), self.x0.clone(), w)
            self.z.append(z); self.ud.append(ud)
            del w
        self.x, self.P = self.x0.clone(), self.P0.clone()
        nv = crx.mpc_n_vars(Tm)
        self.slots = [dict(tind=torch.zeros(self.n_plan, dtype=torch.int32, device=device),
                           e=torch.empty(self.n_plan, dtype=torch.float32, device=device),
                           xref=torch.empty((self.n_plan, 4 * Tm), dtype=torch.float32, device=device),
                           sol=torch.empty((self.n_plan, nv), dtype=torch.float32, device=device),
                           status=torch.empty(self.n_plan, dtype=torch.int32, device=device),
                           cost=torch.empty(self.n_plan, dtype=torch.float64, device=device)) for _ in range(max(1, depth))]
        self.ekf_events = [] if record_ekf_events else None
        if torch.device(device).type == "cuda" and depth + 1 > (self.hw_queues or 4):
            import warnings
            warnings.warn(f"SwarmShard: {depth} planner streams + the launch stream on {self.hw_queues or 4} hardware queues — streams that share a "
                          f"queue run one after the other; set GPU_MAX_HW_QUEUES >= {depth + 2} before the process first touches the GPU "
                          "(cpprobotics_amd.swarm.want_hw_queues)")
        self.rnd = MixedSwarmRound(n, T, 4, chunks, plan_every, device, self._ekf_launch, lambda: self.x, self._plan_launch, gather=gather,
                                   n_total=self.n_total, group=group, depth=depth)

    def _ekf_launch(self, c, t0, t1, hist):
        s = self.rnd.round % len(self.z)
        if c == 0:
            self.x.copy_(self.x0); self.P.copy_(self.P0)
        ev = self.ekf_events
        if ev is not None:
            e0 = torch.cuda.Event(enable_timing=True); e0.record()
        self.crx.ekf_run(self.x, self.P, self.z[s][t0:t1], self.ud[s][t0:t1], self.q, self.r, x_hist=hist)
        if ev is not None:
            e1 = torch.cuda.Event(enable_timing=True); e1.record(); ev.append((e0, e1))

    def _plan_launch(self, est, slot):
        b = self.slots[slot]
        est[:, 3] = self.v_cmd
        self.crx.calc_nearest_index(est, self.dc, b["tind"], e=b["e"])
        self.crx.calc_ref_trajectory(est, self.dc, b["tind"], self.Tm, out=b["xref"])
        self.mpc_fn(est, b["xref"], self.Tm, (b["sol"], b["status"], b["cost"]))
        return b

    def run(self):
        self.rnd.run()
        return self

    def wait(self):
        self.rnd.wait()

    def algorithmic_bytes_ekf(self):
        """HBM bytes the round's EKF launches must move (32 B per update + 160 B per vehicle per launch)."""
        return 32.0 * self.n * self.T + 160.0 * self.n * self.rnd.chunks
