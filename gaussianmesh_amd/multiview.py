"""View-sharded multi-GPU rendering of one (deforming) Gaussian cloud -- one process per GPU.

The reference is single-GPU only; this is the MI355X-native scale-out defined by BASELINE.json:
cameras of a trajectory are independent given the deformed cloud, so views shard across ranks with no
collective inside a frame.  The only exchange steps are
  * broadcast_cloud():      once, the static cloud (positions, covariances, opacities, SH, mesh binding)
  * broadcast_mesh_state(): per deformation frame, the proxy-mesh state (V1, R, S) = 84 B per vertex
                            (0.63 MB for 7.5k vertices) -- every rank then runs the deform kernel locally,
                            instead of shipping the 36-48 MB deformed cloud.
  * MeshStatePipe:          the exchange of a pipelined frame loop: the per-frame state of several consecutive frames in one
                            broadcast, one batch ahead of their use, on a stream of its own.  bench.py ships only the deformed
                            VERTEX POSITIONS (12 B per vertex, 90 KB per frame) and lets every rank derive (R, S) itself
                            (gm_mesh_rs_packed on the frame's stream, as at N = 1), so all ranks run the same kernels.
Both are torch.distributed broadcasts: RCCL over xGMI with backend "nccl" on the GPUs, gloo in the CPU tests.
"""
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_views(n_views, rank=None, world_size=None):
    """Contiguous block partition of view indices: rank r renders [r*n/W, (r+1)*n/W) (remainder to low ranks)."""
    if rank is None or world_size is None:
        rank, world_size = world()
    base, rem = divmod(n_views, world_size)
    start = rank * base + min(rank, rem)
    return list(range(start, start + base + (1 if rank < rem else 0)))


def view_for_step(step, n_views, rank=None, world_size=None):
    """Camera index rendered by `rank` at loop step `step`: ranks walk their own block of the trajectory."""
    if rank is None or world_size is None:
        rank, world_size = world()
    mine = shard_views(n_views, rank, world_size)
    if not mine:
        return step % n_views
    return mine[step % len(mine)]


def _staged(t):
    """gloo moves host memory: device tensors are staged through the host (functional tests of the N>1 path on one GPU,
    CPU tests); with RCCL (backend "nccl") the collective works on the device buffer itself."""
    return t.is_cuda and dist.get_backend() == "gloo"


def _broadcast(t, src):
    if _staged(t):
        h = t.detach().cpu()
        dist.broadcast(h, src=src)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src)
    return t


def broadcast_cloud(tensors, src=0):
    """One-time broadcast of the shared cloud.  `tensors`: dict name -> tensor allocated with the right shape and
    dtype on every rank (contents only meaningful on `src`)."""
    rank, ws = world()
    if ws == 1:
        return tensors
    for name in sorted(tensors):
        _broadcast(tensors[name], src)
    return tensors


def broadcast_mesh_state(state, src=0):
    """Per-frame broadcast of the packed mesh state [Vm, 21] = (V1 | R row-major | S row-major)."""
    rank, ws = world()
    if ws > 1:
        _broadcast(state, src)
    return state


class MeshStatePipe:
    """Per-frame mesh states of a frame loop, exchanged in batches, one batch ahead of their use.

    Rank `src` owns the animation: produce(i, out) writes the state of loop step i into `out` (rank src only; on a GPU it
    enqueues on the current stream).  frame(i) returns the state of step i on every rank.  Steps must be requested in
    non-decreasing order, by all ranks alike.  The states of `batch` consecutive steps travel in ONE broadcast, issued (on the
    pipe's own stream) when the batch before it is first used, so the collective and the producer kernels of batch b+1 overlap
    the rendering of batch b and a frame costs 1/batch of a collective.  The only wait is on the host - for an event recorded
    a whole batch earlier - so the render streams carry no cross-stream dependency.  batch = 1 is a broadcast per frame (an
    interactive editor: no look-ahead).

    frames_in_flight bounds how many steps after frame(i) the device may still be reading the state of step i (the caller's
    pipelining depth); a slot is overwritten only after that many further steps have been requested."""

    def __init__(self, produce, frame_shape, batch, device, src=0, frames_in_flight=1, dtype=torch.float32, produce_batch=None):
        """produce_batch(b, buf) (optional, instead of produce): fills all `batch` states of batch b at once - buf[j] is the state
        of step b * batch + j - e.g. one gather launch instead of `batch` copies."""
        self.produce, self.produce_batch, self.batch, self.src = produce, produce_batch, max(1, int(batch)), src
        self.rank, self.ws = world()
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.slots = -(-int(frames_in_flight) // self.batch) + 2          # in use + being filled + still read by frames in flight
        self.bufs = [torch.empty((self.batch,) + tuple(frame_shape), dtype=dtype, device=self.device) for _ in range(self.slots)]
        self.stream = torch.cuda.Stream(device=self.device) if self.cuda else None
        self.ready = {}                                                   # batch index -> event (None on the CPU)
        self.last = None
        self.broadcasts = 0

    def _issue(self, b):
        buf = self.bufs[b % self.slots]

        def fill():
            if self.rank == self.src:
                if self.produce_batch is not None:
                    self.produce_batch(b, buf)
                else:
                    for j in range(self.batch):
                        self.produce(b * self.batch + j, buf[j])
            if self.ws > 1:
                _broadcast(buf, self.src)
                self.broadcasts += 1
        if self.cuda:
            with torch.cuda.stream(self.stream):
                fill()
                ev = torch.cuda.Event()
                ev.record(self.stream)
            self.ready[b] = ev
        else:
            fill()
            self.ready[b] = None

    def frame(self, i):
        b, j = divmod(i, self.batch)
        if self.last is not None and b < self.last:
            raise ValueError("MeshStatePipe: steps must be requested in non-decreasing order")
        if b not in self.ready:
            self._issue(b)
        if b + 1 not in self.ready:
            self._issue(b + 1)                                            # one batch ahead
        if b != self.last:                                                # once per batch
            if self.ready[b] is not None:
                self.ready[b].synchronize()                               # host-side; recorded a batch ago, normally long complete
            for k in [k for k in self.ready if k < b]:
                del self.ready[k]
        self.last = b
        return self.bufs[b % self.slots][j]


def unpack_mesh_state(state):
    return state[:, 0:3].contiguous(), state[:, 3:12].contiguous(), state[:, 12:21].contiguous()


def max_over_ranks(value, device):
    """MAX all-reduce of a python float (timing)."""
    rank, ws = world()
    if ws == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device):
    """SUM all-reduce of a python float (bench.py: an all-reduce of ones = the number of ranks the backend saw)."""
    rank, ws = world()
    if ws == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_over_ranks(value, device):
    """Every rank's python float, as a list indexed by rank (timing diagnostics: a slow rank shows in the bench line)."""
    rank, ws = world()
    if ws == 1:
        return [float(value)]
    dev = "cpu" if dist.get_backend() == "gloo" else device
    mine = torch.tensor([value], dtype=torch.float64, device=dev)
    got = [torch.empty_like(mine) for _ in range(ws)]
    dist.all_gather(got, mine)
    return [float(t.item()) for t in got]


def rank_core_set(local_rank, local_world, allowed):
    """The cores rank `local_rank` of `local_world` ranks on one node gets out of the sorted list `allowed`: a contiguous block of
    len(allowed) // local_world (at least one core; blocks wrap when there are more ranks than cores)."""
    allowed = sorted(allowed)
    n = len(allowed)
    if local_world <= 1 or n == 0:
        return set(allowed)
    per = max(1, n // local_world)
    start = (local_rank * per) % n
    return {allowed[(start + k) % n] for k in range(per)}


def pin_rank_to_cores(local_rank, local_world):
    """One process per GPU, eight on a node: each rank's frame loop needs ~0.08 ms of ONE host core per 0.2-ms frame, plus the
    runtime's helper threads.  Left to the scheduler, eight Python ranks migrate and now and then share a core, and the slowest
    rank sets a max-over-ranks timing.  Give every rank its own block of the cores this process may use (os.sched_setaffinity:
    inherited by the threads HIP and RCCL start afterwards, so call it before the runtime initialises).  Returns the sorted core
    list, or None where the platform has no affinity call or GM_RANK_AFFINITY=0 asks for none."""
    import os
    if os.environ.get("GM_RANK_AFFINITY", "1") == "0" or not hasattr(os, "sched_setaffinity") or local_world <= 1:
        return None
    try:
        cores = rank_core_set(local_rank, local_world, os.sched_getaffinity(0))
        os.sched_setaffinity(0, cores)
        return sorted(cores)
    except OSError:
        return None


def render_trajectory(n_views, n_frames, mesh_state_of, deform_and_render, state_buffer, src=0):
    """Reference driver of the sharded loop (used by the gloo test and mirrored by bench.py):
    for every deformation frame t, rank `src` produces the mesh state, all ranks receive it and render their views.
      mesh_state_of(t) -> [Vm,21] tensor (called on rank src only)
      deform_and_render(state, view_index) -> image
    Returns {(t, view_index): image} for the views owned by this rank."""
    rank, ws = world()
    out = {}
    for t in range(n_frames):
        if rank == src:
            state_buffer.copy_(mesh_state_of(t))
        broadcast_mesh_state(state_buffer, src)
        for v in shard_views(n_views, rank, ws):
            out[(t, v)] = deform_and_render(state_buffer, v)
    return out
