"""Agent-sharded plan step for ONE large mission on several GPUs (BASELINE.json config C4).

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

What shards, and what does not:
  * Corridor::update shards by agent.  SFC boxes are independent per agent (rbp_corridor.hpp:154), RSFC rows are
    independent per pair and only read the two agents' waypoints (:342-392).  Rank r computes the agents of its
    contiguous slice and the pair rows (qi, qj) whose qi lies in the slice; ONE fused all-gather of a packed byte buffer then gives every
    rank the complete corridor, bit-identical to the unsharded call (the exchange moves bytes, never adds floats).
  * RBPPlanner::update in the reference schedule (sequential=true) is a Gauss-Seidel sweep: batch l is solved against
    the answers of batches < l (rbp_planner.hpp:140-203), so inside one mission it is serial by construction.  Every rank
    therefore runs the same QP sweep on the gathered corridor (replicas: same code, same inputs, same bits) and no
    further exchange is needed; scale-out of the QP comes from sharding MISSIONS (bench.py).  A Jacobi sweep over
    ranks would parallelise it but is a different algorithm with different answers (both agents of a frozen pair move
    at once, so the half-space rows no longer guarantee separation) and is deliberately not offered.
  * RBPPlanner::update with plan/sequential = false (the reference's code default, param.hpp:67) is ONE joint QP over all agents whose
    pair rows couple every agent with every other (rbp_planner.hpp:638-684): it does not split by agent either, but its Newton system
    -- block tridiagonal over the knots, blocks of order 9 N -- does: the grid-wide solver eliminates it from both ends towards the
    middle knot, and PAIRS of ranks share that factorisation (rank 2k the lower chain, rank 2k+1 the upper one; include/rbp.h
    rbp_session_shard_joint, planner.Session.shard_joint).  Per interior-point iteration a pair trades the explicit inverse of each
    chain's last knot (42 MB at 256 agents) and two vectors per Newton solve; every rank ends with the bits of the unsharded solve.
    Two chains = two ranks per mission: with more ranks the pairs are replicas (or take different missions).
"""
import numpy as np

from . import planner
from .types import Mission, Param, PlanResult, World


def agent_slices(n_agents: int, world_size: int):
    """contiguous agent ranges [begin, end) per rank, as even as possible (the SFC growth, ~0.5 M distance samples per
    agent, is the expensive part; the RSFC rows an agent owns are a few flops each); every agent belongs to one rank."""
    base, extra = divmod(n_agents, world_size)
    cuts = [0]
    for r in range(world_size):
        cuts.append(cuts[-1] + base + (1 if r < extra else 0))
    return [(cuts[r], cuts[r + 1]) for r in range(world_size)]


def pair_offset(n_agents: int, qi: int) -> int:
    """index of pair (qi, qi+1) in the qi-major upper-triangular order of RSFC (rbp_corridor.hpp:342-344)."""
    return qi * n_agents - qi * (qi + 1) // 2


def _shard_layout(n_agents: int, M: int, MB: int, slices):
    """byte length of every rank's packed shard: [sfc_count | sfc_box | sfc_time | rsfc_normal rows of the agents it owns]"""
    per_agent = 4 + MB * 6 * 8 + MB * 8
    row = M * 3 * 4
    offs = [(pair_offset(n_agents, sb), pair_offset(n_agents, se)) for sb, se in slices]
    lens = [(se - sb) * per_agent + (o1 - o0) * row for (sb, se), (o0, o1) in zip(slices, offs)]
    return per_agent, row, offs, lens


def pack_shard(plan: PlanResult, n_agents: int, sl, off):
    (b, e), (o0, o1) = sl, off
    parts = [plan.sfc_count[b:e], plan.sfc_box[b:e], plan.sfc_time[b:e], plan.rsfc_normal[o0:o1]]
    return np.concatenate([np.ascontiguousarray(a).view(np.uint8).reshape(-1) for a in parts])


def unpack_shard(plan: PlanResult, buf: np.ndarray, sl, off):
    (b, e), (o0, o1) = sl, off
    pos = 0
    for arr, lo, hi in ((plan.sfc_count, b, e), (plan.sfc_box, b, e), (plan.sfc_time, b, e), (plan.rsfc_normal, o0, o1)):
        dst = arr[lo:hi]
        n = dst.nbytes
        dst.view(np.uint8).reshape(-1)[:] = buf[pos:pos + n]
        pos += n
    assert pos == buf.size


def gather_corridor(dist, plan: PlanResult, n_agents: int, slices, dev="cpu"):
    """exchange the shards written by rbp_corridor_update_range so that `plan` holds the whole corridor on every rank: ONE fused
    all-gather of a packed byte buffer (SFC counts, boxes, end times and the RSFC rows of the owned agents; shards padded to the
    longest one).  Bytes only -- nothing is added, so -0.0 normals and every bit of the boxes survive."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    M, MB = plan.M, plan.sfc_box.shape[1]
    _, _, offs, lens = _shard_layout(n_agents, M, MB, slices)
    maxb = max(max(lens), 1)
    buf = np.zeros(maxb, dtype=np.uint8)
    mine = pack_shard(plan, n_agents, slices[rank], offs[rank])
    assert mine.size == lens[rank]
    buf[:mine.size] = mine
    t = torch.from_numpy(buf).to(dev)
    out = torch.empty(world * maxb, dtype=torch.uint8, device=dev)
    # (the path is chosen from the backend, not by catching what the collective throws: a genuine RCCL failure must surface, and ranks
    # that disagree about a fallback would issue mismatched collectives and hang)
    if dist.get_backend() == "gloo":   # gloo: the list form (tests)
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        out = torch.cat(outs)
    else:
        dist.all_gather_into_tensor(out, t)          # one collective: RCCL over xGMI on the GPU box
    host = out.cpu().numpy()
    for r in range(world):
        if r != rank:
            unpack_shard(plan, host[r * maxb:r * maxb + lens[r]], slices[r], offs[r])


# ---- the same exchange on device buffers (no host round trip between Corridor::update and RBPPlanner::update) -------------------
def pack_shard_device(arrs, sl, off):
    """rank-local shard of the session's corridor arrays (torch tensors over HBM, planner.Session.device_arrays) as one byte tensor"""
    import torch
    (b, e), (o0, o1) = sl, off
    parts = [arrs["sfc_count"][b:e], arrs["sfc_box"][b:e], arrs["sfc_time"][b:e], arrs["rsfc_normal"][o0:o1]]
    return torch.cat([t.contiguous().view(torch.uint8).reshape(-1) for t in parts])


def unpack_shard_device(arrs, buf, sl, off):
    import torch
    (b, e), (o0, o1) = sl, off
    pos = 0
    for name, lo, hi in (("sfc_count", b, e), ("sfc_box", b, e), ("sfc_time", b, e), ("rsfc_normal", o0, o1)):
        dst = arrs[name][lo:hi]
        n = dst.numel() * dst.element_size()
        dst.view(torch.uint8).reshape(-1).copy_(buf[pos:pos + n])
        pos += n
    assert pos == buf.numel()


def gather_corridor_device(dist, arrs, n_agents, slices):
    """gather_corridor on the session's own device arrays: pack (device), ONE all_gather_into_tensor (RCCL over xGMI), unpack (device)"""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    M, MB = arrs["rsfc_normal"].shape[1], arrs["sfc_box"].shape[1]
    _, _, offs, lens = _shard_layout(n_agents, M, MB, slices)
    maxb = max(max(lens), 1)
    mine = pack_shard_device(arrs, slices[rank], offs[rank])
    assert mine.numel() == lens[rank]
    t = torch.zeros(maxb, dtype=torch.uint8, device=mine.device)
    t[:mine.numel()] = mine
    out = torch.empty(world * maxb, dtype=torch.uint8, device=mine.device)
    if dist.get_backend() == "gloo":   # tests: two ranks on one GPU (RCCL refuses that); gloo has neither the flat form nor device all_gather
        outs = [torch.empty(maxb, dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(outs, t.cpu())
        out = torch.cat(outs).to(mine.device)
    else:
        dist.all_gather_into_tensor(out, t)          # RCCL: one collective on the device buffers
    for r in range(world):
        if r != rank:
            unpack_shard_device(arrs, out[r * maxb:r * maxb + lens[r]], slices[r], offs[r])


_pair_groups = {}


def pair_group(dist):
    """the process group {2k, 2k+1} this rank shares a joint factorisation with (None: odd world size, or a single rank).  Created once
    per world size -- every rank has to take part in the creation of every pair's group."""
    ws, rank = dist.get_world_size(), dist.get_rank()
    if ws < 2 or ws % 2:
        return None
    if ws == 2:
        return dist.group.WORLD
    if ws not in _pair_groups:
        _pair_groups[ws] = [dist.new_group([2 * k, 2 * k + 1]) for k in range(ws // 2)]
    return _pair_groups[ws][rank // 2]


def plan_sharded_device(world: World, mission: Mission, param: Param, plan: PlanResult, dist=None, device=None, stats=None):
    """plan_sharded with the mission RESIDENT in one session per rank: the CORRIDOR stage runs on the rank's agent slice, the shards are
    exchanged between the sessions' HBM arrays (gather_corridor_device), the PLANNER stage runs on the completed corridor -- nothing but
    the 4-byte ok flag crosses to the host between the two stages.  Returns (ok, error text)."""
    import torch
    from . import _abi as A
    dev = torch.device(device if device is not None else "cuda")
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    n = mission.qn
    ws = 1 if dist is None else dist.get_world_size()
    rank = 0 if dist is None else dist.get_rank()
    slices = agent_slices(n, ws)
    stream = torch.cuda.current_stream(idx).cuda_stream
    sess = planner.Session([world], [mission], param, [plan], device=idx)
    try:
        arrs = sess.device_arrays(0)
        sess.set_agent_range(*slices[rank])
        sess.run(A.RBP_STAGE_CORRIDOR, stream)
        flag = (arrs["status"] != 0).to(torch.int32)
        if dist is not None:
            dist.all_reduce(flag)  # Corridor::update returns false if any agent / pair failed on any rank
        if int(flag.item()) != 0:
            st = sess.download(stream)
            return False, planner.ERROR_TEXT.get(st[0], "corridor failed on another rank")
        if dist is not None and ws > 1:
            gather_corridor_device(dist, arrs, n, slices)
        sess.set_agent_range(0, n)
        # a joint QP on the grid-wide solver: the ranks of a pair share its factorisation (module docstring)
        if dist is not None and not param.sequential and n >= planner.solver_opts().joint_wide_min_agents > 0:
            grp = pair_group(dist)
            if grp is not None:
                sess.shard_joint(dist, grp)
        sess.run(A.RBP_STAGE_PLANNER, stream)
        if stats is not None:
            stats["exchanges"], stats["exchange_bytes"] = getattr(sess, "exchanges", 0), getattr(sess, "exchange_bytes", 0)
        st = sess.download(stream)
        return st[0] == 0, ("" if st[0] == 0 else planner.ERROR_TEXT.get(st[0], str(st[0])))
    finally:
        sess.close()


class ShardedCorridor:
    """Corridor::update for one mission whose agents are spread over the ranks of `dist` (None = single process)."""

    def __init__(self, world: World, mission: Mission, param: Param, dist=None, device="cpu", compute=None):
        self.world, self.mission, self.param, self.dist, self.device = world, mission, param, dist, device
        self.last_error = ""
        # compute(plan, begin, end) -> bool: the shard kernel.  Default: the HIP path through the C ABI.
        self._cor = planner.Corridor(world, mission, param) if compute is None else None
        self._compute = compute or (lambda plan, b, e: self._cor.update_range(plan, b, e))

    def update(self, log: bool, plan: PlanResult) -> bool:
        import torch
        n = self.mission.qn
        ws = 1 if self.dist is None else self.dist.get_world_size()
        rank = 0 if self.dist is None else self.dist.get_rank()
        slices = agent_slices(n, ws)
        ok = bool(self._compute(plan, *slices[rank]))
        if self._cor is not None:
            self.last_error = self._cor.last_error
        if self.dist is not None:
            flag = torch.tensor([0 if ok else 1], dtype=torch.int32, device=self.device)
            self.dist.all_reduce(flag)  # Corridor::update returns false if any agent / pair failed
            ok = int(flag.item()) == 0
            if ok:
                gather_corridor(self.dist, plan, n, slices, self.device)
        return ok


def plan_sharded(world: World, mission: Mission, param: Param, plan: PlanResult, dist=None, device="cpu"):
    """Corridor (agent-sharded + all-gather) then RBPPlanner (replicated sweep) for one mission; every rank returns with
    the complete PlanResult.  Returns (ok, error text).  On a GPU the mission stays resident in a session and the exchange runs on
    device buffers (plan_sharded_device); the host-buffer path below serves the CPU (gloo) tests of the exchange logic."""
    import torch
    if torch.device(device).type == "cuda":
        return plan_sharded_device(world, mission, param, plan, dist, device)
    cor = ShardedCorridor(world, mission, param, dist, device)
    if not cor.update(False, plan):
        return False, cor.last_error or "corridor failed on another rank"
    pl = planner.RBPPlanner(mission, param)
    ok = pl.update(False, plan)
    return ok, pl.last_error
