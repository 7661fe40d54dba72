"""Agent-range sharding across the GPUs of one node (one process per GPU).

Every agent's QP / NLP depends only on its own state, goal, previous input
and obstacle rows (the reference's multi-robot example steps robots fully
independently, examples/test_multi_robot.py:77-80), so the batch shards by
contiguous agent ranges with NO collective on the solve path.  Collectives
(RCCL on GPUs, gloo in the CPU tests) appear only at the edges: scattering
inputs / gathering results when the host asks, the timing reduction of
bench.py, the broadcast of a shared obstacle table and -- an extension with
no reference counterpart -- the all-gather of agent states used when agents
are each other's obstacles.
"""
import torch
import torch.distributed as dist


def agent_range(n_agents, world_size, rank):
    """Contiguous range [lo, hi) owned by ``rank``; sizes differ by at most one."""
    base, rem = divmod(int(n_agents), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sizes(n_agents, world_size):
    return [agent_range(n_agents, world_size, r)[1] - agent_range(n_agents, world_size, r)[0]
            for r in range(world_size)]


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def scatter_agents(full, n_agents, src=0):
    """Rank ``src`` holds ``full`` [n_agents, ...]; every rank gets its contiguous shard."""
    ws, rank = world()
    if ws == 1:
        return full
    sizes = shard_sizes(n_agents, ws)
    meta = [None]
    if rank == src:
        meta = [(tuple(full.shape[1:]), full.dtype, full.device.type)]
    dist.broadcast_object_list(meta, src=src)
    tail, dtype, _ = meta[0]
    dev = full.device if rank == src else (torch.device("cuda", torch.cuda.current_device())
                                          if dist.get_backend() == "nccl" else torch.device("cpu"))
    out = torch.empty((sizes[rank],) + tail, dtype=dtype, device=dev)
    if rank == src:
        chunks = list(torch.split(full.contiguous(), sizes, dim=0))
        # dist.scatter needs equal sizes: send point-to-point instead
        for r, ch in enumerate(chunks):
            if r == src:
                out.copy_(ch)
            else:
                dist.send(ch.contiguous(), dst=r)
    else:
        dist.recv(out, src=src)
    return out


def gather_agents(local, n_agents, dst=0):
    """Inverse of scatter_agents: rank ``dst`` returns the [n_agents, ...] tensor, others None."""
    ws, rank = world()
    if ws == 1:
        return local
    sizes = shard_sizes(n_agents, ws)
    if rank == dst:
        parts = []
        for r in range(ws):
            if r == dst:
                parts.append(local)
            else:
                buf = torch.empty((sizes[r],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
                dist.recv(buf, src=r)
                parts.append(buf)
        return torch.cat(parts, dim=0)
    dist.send(local.contiguous(), dst=dst)
    return None


def broadcast_obstacle_table(table, src=0):
    """Shared [K,7] obstacle table: one broadcast per step (K*7 values, latency-bound)."""
    ws, _ = world()
    if ws > 1:
        dist.broadcast(table, src=src)
    return table


def all_gather_states(X_local, n_agents):
    """Neighbour-state exchange (extension): every rank gets all agents' states [n_agents, nx].

    One all-gather per control step; over xGMI this is latency-bound
    (65 536 agents x 16 B = 1 MiB in total).  Shards may differ by one agent,
    so the padded all_gather_into_tensor form is used and the padding dropped.
    """
    ws, rank = world()
    if ws == 1:
        return X_local
    sizes = shard_sizes(n_agents, ws)
    m = max(sizes)
    pad = torch.zeros((m,) + tuple(X_local.shape[1:]), dtype=X_local.dtype, device=X_local.device)
    pad[: X_local.shape[0]] = X_local
    out = torch.empty((ws * m,) + tuple(X_local.shape[1:]), dtype=X_local.dtype, device=X_local.device)
    dist.all_gather_into_tensor(out, pad)
    return torch.cat([out[r * m: r * m + sizes[r]] for r in range(ws)], dim=0)


def max_over_ranks(seconds, device=None):
    """bench.py timing contract: the job time is the MAX over ranks."""
    ws, _ = world()
    if ws == 1:
        return float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64,
                     device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def neighbor_obstacles(X_local, n_agents, K, neighbour_radius):
    """Agents-as-obstacles step (extension, BASELINE config 4): all-gather every agent's state
    (one RCCL all-gather per control step when sharded; a no-op on one rank), then let the HIP kernel
    pick each local agent's K nearest other agents as moving circular obstacles.

    Returns ``obs [B_local, K, 7]`` ready for ``BatchedCBFQP.solve`` (C3BF / DPCBF use the velocity columns).
    """
    import ctypes as C

    from . import _lib
    ws, rank = world()
    X_all = all_gather_states(X_local, n_agents).contiguous()
    lo, hi = agent_range(n_agents, ws, rank)
    assert hi - lo == X_local.shape[0], "X_local must be this rank's agent_range shard"
    obs = torch.empty((hi - lo, K, 7), dtype=X_local.dtype, device=X_local.device)
    io = _lib.DTYPE_F32 if X_local.dtype == torch.float32 else _lib.DTYPE_F64
    stream = torch.cuda.current_stream(X_local.device).cuda_stream
    lib = _lib.load()
    nbytes = int(lib.sc_neighbor_workspace_bytes(io, n_agents, hi - lo, K))
    ws_ = torch.empty((max(nbytes, 8),), dtype=torch.uint8, device=X_local.device)   # caching allocator: no hipMalloc per step
    rc = lib.sc_neighbor_obstacles_batch_ws(io, n_agents, lo, hi - lo, K, float(neighbour_radius), X_all.data_ptr(),
                                            obs.data_ptr(), ws_.data_ptr(), nbytes, stream)
    _lib.check(rc, "sc_neighbor_obstacles_batch_ws")
    return obs


def _all_gather_into(out, inp):
    """all_gather_into_tensor; device tensors under the gloo backend (the 1-GPU control-flow test only) go through the host."""
    if inp.is_cuda and dist.get_backend() == "gloo":
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(o, inp.cpu())
        out.copy_(o)
    else:
        dist.all_gather_into_tensor(out, inp)


class NeighborExchange:
    """Persistent buffers for the per-step neighbour-state exchange + selection (BASELINE config 4): the all-gather
    destination, the obstacle rows and the selection workspace are allocated once; with equal shards the all-gather writes
    straight into the final [n_agents, nx] layout (``all_gather_into_tensor``, no padding, no concatenation), with ragged
    shards into a padded buffer whose valid rows are packed by one index_select on a precomputed index."""

    def __init__(self, n_agents, K, neighbour_radius, nx=4, dtype=torch.float32, device="cuda"):
        from . import _lib
        self.ws, self.rank = world()
        self.n_agents, self.K, self.radius = int(n_agents), int(K), float(neighbour_radius)
        self.lo, self.hi = agent_range(n_agents, self.ws, self.rank)
        self.sizes = shard_sizes(n_agents, self.ws)
        self.equal = len(set(self.sizes)) == 1
        m = max(self.sizes)
        dev = torch.device(device)
        self.X_all = torch.empty((n_agents, nx), dtype=dtype, device=dev)
        if self.ws > 1 and not self.equal:
            self.pad_in = torch.zeros((m, nx), dtype=dtype, device=dev)
            self.pad_out = torch.empty((self.ws * m, nx), dtype=dtype, device=dev)
            self.pack = torch.cat([torch.arange(r * m, r * m + self.sizes[r]) for r in range(self.ws)]).to(dev)
        self.obs = torch.empty((self.hi - self.lo, K, 7), dtype=dtype, device=dev)
        self.io = _lib.DTYPE_F32 if dtype == torch.float32 else _lib.DTYPE_F64
        self._lib = _lib.load()
        self.nbytes = int(self._lib.sc_neighbor_workspace_bytes(self.io, n_agents, self.hi - self.lo, K))
        self.work = torch.empty((max(self.nbytes, 8),), dtype=torch.uint8, device=dev)

    def gather(self, X_local):
        """All agents' states on every rank (one RCCL all-gather; a copy on one rank)."""
        if self.ws == 1:
            self.X_all.copy_(X_local)
        elif self.equal:
            _all_gather_into(self.X_all, X_local.contiguous())
        else:
            self.pad_in[: X_local.shape[0]] = X_local
            _all_gather_into(self.pad_out, self.pad_in)
            torch.index_select(self.pad_out, 0, self.pack, out=self.X_all)
        return self.X_all

    def step(self, X_local):
        """gather + K nearest other agents as moving obstacles for the local shard -> obs [B_local, K, 7]."""
        from . import _lib
        X_all = self.gather(X_local)
        stream = torch.cuda.current_stream(X_local.device).cuda_stream
        rc = self._lib.sc_neighbor_obstacles_batch_ws(self.io, self.n_agents, self.lo, self.hi - self.lo, self.K, self.radius,
                                                      X_all.data_ptr(), self.obs.data_ptr(), self.work.data_ptr(), self.nbytes, stream)
        _lib.check(rc, "sc_neighbor_obstacles_batch_ws")
        return self.obs
