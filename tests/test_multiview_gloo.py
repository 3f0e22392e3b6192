"""N>1 path on CPU: two gloo ranks shard the views of a deforming cloud; the static cloud is broadcast once and the
mesh state once per frame (gaussianmesh_amd/multiview.py).  The renderer is pluggable; here it is the CPU oracle
(tests may use the oracle as a checker / stand-in renderer, the product path cannot)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def test_shard_views_partition():
    from gaussianmesh_amd.multiview import shard_views, view_for_step
    for n, w in [(64, 8), (10, 4), (3, 8), (7, 2)]:
        parts = [shard_views(n, r, w) for r in range(w)]
        assert sorted(sum(parts, [])) == list(range(n))
        assert max(map(len, parts)) - min(map(len, parts)) <= 1
    assert shard_views(64, 3, 8) == list(range(24, 32))         # C4: 8 views per GPU
    assert [view_for_step(s, 64, 1, 8) for s in range(10)] == [8, 9, 10, 11, 12, 13, 14, 15, 8, 9]


def test_rank_core_blocks_are_disjoint_and_cover():
    """multiview.rank_core_set: the block of host cores each of the node's ranks is pinned to (bench.py, before HIP starts)"""
    from gaussianmesh_amd import multiview
    allowed = list(range(4, 132))                                # e.g. a cgroup that starts at core 4
    blocks = [multiview.rank_core_set(r, 8, allowed) for r in range(8)]
    assert all(len(b) == 16 for b in blocks) and set().union(*blocks) == set(allowed)
    assert all(blocks[i].isdisjoint(blocks[j]) for i in range(8) for j in range(i))
    assert blocks[0] == set(range(4, 20)) and blocks[7] == set(range(116, 132))
    assert multiview.rank_core_set(0, 1, allowed) == set(allowed)
    three = [multiview.rank_core_set(r, 8, [0, 1, 2]) for r in range(8)]          # more ranks than cores: one core each, wrapping
    assert all(len(b) == 1 for b in three) and set().union(*three) == {0, 1, 2}
    assert multiview.pin_rank_to_cores(0, 1) is None             # single rank: nothing is pinned


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
    from gaussianmesh_amd import multiview, scenes
    from oracle import oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N, W, H, views, frames = 400, 48, 32, 4, 2
    verts, faces = scenes.torus_mesh(12, 8)
    Vm = verts.shape[0]
    shapes = dict(tri=((N, 3), torch.int32), weights=((N, 3), torch.float32), pos=((N, 3), torch.float32),
                  cov=((N, 3, 3), torch.float32), opac=((N, 1), torch.float32), shs=((N, 16, 3), torch.float32),
                  verts=((Vm, 3), torch.float32))
    cloud = {k: torch.zeros(s, dtype=d) for k, (s, d) in shapes.items()}
    if rank == 0:                                               # only rank 0 knows the scene
        cl = scenes.bind_cloud_to_mesh(N, verts, faces, seed=3)
        cl["scales"] *= 6
        cov = scenes.cov3d_from_scale_rot(cl["scales"], cl["rots"]).astype(np.float32)
        src = dict(tri=cl["tri"], weights=cl["weights"], pos=cl["means"], cov=cov, opac=cl["opac"], shs=cl["shs"], verts=verts)
        for k in cloud:
            cloud[k].copy_(torch.tensor(np.asarray(src[k]), dtype=shapes[k][1]).reshape(shapes[k][0]))
    multiview.broadcast_cloud(cloud, src=0)
    c = {k: v.numpy() for k, v in cloud.items()}

    def mesh_state_of(t):
        V1, Rv, Sv = scenes.twist_bend_frame(verts, t + 3, period=16)
        return torch.tensor(np.concatenate([V1, Rv.reshape(-1, 9), Sv.reshape(-1, 9)], 1), dtype=torch.float32)

    def deform_and_render(state, v):
        V1, Rv, Sv = (x.numpy() for x in multiview.unpack_mesh_state(state))
        p, cv, r = orc.deform(c["tri"], c["weights"], V1 - c["verts"], Rv.reshape(-1, 3, 3), Sv.reshape(-1, 3, 3), c["cov"], c["pos"])
        cam = scenes.orbit_camera(v, views, W, H, radius=6.0)
        rgb = orc.sh_colors_rotated(p, cam["campos"], r, c["shs"])
        sc = dict(means=p, opac=c["opac"], colors_precomp=rgb, cov3D_precomp=scenes.strip_symmetric(cv))
        return orc.forward_fast(sc, cam, np.ones(3, np.float32), use_precomp_cov=True, use_precomp_color=True)[0]

    state = torch.zeros((Vm, 21), dtype=torch.float32)
    out = multiview.render_trajectory(views, frames, mesh_state_of, deform_and_render, state, src=0)
    t = multiview.max_over_ranks(float(rank + 1), torch.device("cpu"))
    every = multiview.gather_over_ranks(float(rank + 1), torch.device("cpu"))
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), tmax=t, tall=np.array(every), **{"%d_%d" % k: v for k, v in out.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_matches_single_process(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = dict(np.load(tmp_path / "rank0.npz")); r1 = dict(np.load(tmp_path / "rank1.npz"))
    assert float(r0.pop("tmax")) == 2.0 and float(r1.pop("tmax")) == 2.0      # MAX all-reduce over ranks
    assert r0.pop("tall").tolist() == [1.0, 2.0] and r1.pop("tall").tolist() == [1.0, 2.0]      # every rank's own value, by rank
    assert sorted(r0) == ["0_0", "0_1", "1_0", "1_1"] and sorted(r1) == ["0_2", "0_3", "1_2", "1_3"]
    # single-process reference of the same trajectory
    port2 = _free_port()
    single = tmp_path / "single"; single.mkdir()
    mp.spawn(_worker, args=(1, port2, str(single)), nprocs=1, join=True)
    s = dict(np.load(single / "rank0.npz")); s.pop("tmax"); assert s.pop("tall").tolist() == [1.0]
    both = {**r0, **r1}
    assert sorted(both) == sorted(s)
    for k in s:
        assert np.array_equal(both[k], s[k]), k
        assert s[k].std() > 0                                    # something was actually rendered


def _pipe_worker(rank, world, port, out_dir, batch, in_flight):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from gaussianmesh_amd import multiview
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    produced = []

    def produce(i, out):                                        # rank 0 only: the state of loop step i
        assert rank == 0
        produced.append(i)
        out.copy_(torch.full((5, 3), float(i)) + torch.arange(15, dtype=torch.float32).view(5, 3) / 100)

    pipe = multiview.MeshStatePipe(produce, (5, 3), batch, "cpu", src=0, frames_in_flight=in_flight)
    held = []                                                   # states a pipelined caller would still be reading
    got = []
    for i in range(-7, 30):                                     # (bench.py's loop starts below zero)
        st = pipe.frame(i)
        held.append((i, st))
        held = held[-in_flight:]
        for k, h in held:                                       # nothing in flight has been overwritten by a later batch
            assert float(h[0, 0]) == float(k), (i, k, float(h[0, 0]))
        got.append(st.clone())
    with pytest.raises(ValueError):
        pipe.frame(-7 - 2 * batch)                              # going back is refused
    np.savez(os.path.join(out_dir, "pipe%d.npz" % rank), got=torch.stack(got).numpy(), broadcasts=pipe.broadcasts,
             produced=np.array(produced, dtype=np.int64), slots=pipe.slots)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("batch,in_flight", [(8, 8), (1, 3), (3, 8)])
def test_mesh_state_pipe_two_ranks(tmp_path, batch, in_flight):
    """MeshStatePipe: every rank sees the state rank 0 produced for each step, in order, one broadcast per `batch` steps,
    and a slot is not reused while a caller `in_flight` steps deep could still read it."""
    mp.spawn(_pipe_worker, args=(2, _free_port(), str(tmp_path), batch, in_flight), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "pipe0.npz"), np.load(tmp_path / "pipe1.npz")
    want = np.stack([np.full((5, 3), float(i), np.float32) + np.arange(15, dtype=np.float32).reshape(5, 3) / 100 for i in range(-7, 30)])
    assert np.array_equal(r0["got"], want) and np.array_equal(r1["got"], want)
    nb = int(r0["broadcasts"])
    assert nb == int(r1["broadcasts"]) and nb == len(set(i // batch for i in range(-7, 30))) + 1       # (+1: the batch issued ahead)
    assert len(r1["produced"]) == 0 and list(r0["produced"]) == sorted(r0["produced"]) and len(r0["produced"]) == nb * batch
    assert int(r0["slots"]) == -(-in_flight // batch) + 2
