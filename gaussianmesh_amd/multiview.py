"""View-sharded multi-GPU rendering of one (deforming) Gaussian cloud -- one process per GPU.

The reference is single-GPU only; this is the MI355X-native scale-out defined by BASELINE.json:
cameras of a trajectory are independent given the deformed cloud, so views shard across ranks with no
collective inside a frame.  The only exchange steps are
  * broadcast_cloud():      once, the static cloud (positions, covariances, opacities, SH, mesh binding)
  * broadcast_mesh_state(): per deformation frame, the proxy-mesh state (V1, R, S) = 84 B per vertex
                            (0.63 MB for 7.5k vertices) -- every rank then runs the deform kernel locally,
                            instead of shipping the 36-48 MB deformed cloud.
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
