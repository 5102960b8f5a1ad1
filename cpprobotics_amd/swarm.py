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


class MixedSwarmRound:
    """One round of BASELINE.json configs[4] on this rank's shard of a mixed EKF + MPC swarm (scripts/swarm_bench.py runs it on the
    GPUs, tests/test_dist_cpu.py with gloo and the CPU oracle standing in for the launches):

      1. the shard's vehicles run T fused EKF steps — cut into `chunks` launches whose histories are all-gathered one by one while
         the next chunk computes (ChunkedTrajectoryGather; `gather="final"` gathers the final estimates only, "none" nothing);
      2. every `plan_every`-th vehicle plans from its final estimate: plan_launch(est) — on a second stream on GPUs, so that the
         planners of round r overlap the EKF launches of round r + 1 (they only read `est`, which the next round refills after
         waiting for them).

    ekf_launch(c, t0, t1, hist): enqueue EKF steps [t0, t1) of this shard, history into hist [t1 - t0, n_local, C]; it owns the
    filter state and must reset it when c == 0.  final_state(): the shard's [n_local, C] estimate after the last chunk.
    plan_launch(est): enqueue the planners on est [n_plan, C] (a buffer owned by this object); its return value is `plans`."""

    def __init__(self, n_local, T, C, chunks, plan_every, device, ekf_launch, final_state, plan_launch, gather="traj", n_total=None, group=None):
        self.nl, self.T, self.C, self.every, self.gather_kind, self.group = n_local, T, C, int(plan_every), gather, group
        self.ekf_launch, self.final_state, self.plan_launch = ekf_launch, final_state, plan_launch
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.n_total = n_total if n_total is not None else n_local * self.world
        self.cuda = torch.device(device).type == "cuda"
        self.chunks = chunks
        assert T % chunks == 0, "the chunk count must divide the number of steps"
        self.cg = ChunkedTrajectoryGather(T, n_local, C, chunks, device, group=group) if (gather == "traj" and dist.is_initialized()) else None
        self.local_hist = None if self.cg is not None else torch.empty((chunks, T // chunks, n_local, C), dtype=torch.float32, device=device)
        self.est = torch.empty(((n_local + self.every - 1) // self.every, C), dtype=torch.float32, device=device)
        self.plan_stream = torch.cuda.Stream(device=device) if self.cuda else None
        self.ekf_done = torch.cuda.Event() if self.cuda else None
        self.plans = None
        self.final = None

    def run(self):
        main = torch.cuda.current_stream() if self.cuda else None
        if self.cg is not None:
            self.cg.run(self.ekf_launch)
        else:
            Tc = self.T // self.chunks
            for c in range(self.chunks):
                self.ekf_launch(c, c * Tc, (c + 1) * Tc, self.local_hist[c])
        if self.cuda:
            main.wait_stream(self.plan_stream)            # the planners of the previous round still read self.est
        x = self.final_state()
        self.est.copy_(x[:: self.every])
        if self.cuda:
            self.ekf_done.record(main)
            with torch.cuda.stream(self.plan_stream):
                self.plan_stream.wait_event(self.ekf_done)
                self.plans = self.plan_launch(self.est)
        else:
            self.plans = self.plan_launch(self.est)
        if self.gather_kind == "final" and dist.is_initialized():
            self.final = gather_agents(x, self.n_total, self.group)
        return self

    def wait(self):
        if self.cg is not None:
            self.cg.wait()
        if self.cuda:
            torch.cuda.current_stream().wait_stream(self.plan_stream)

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
