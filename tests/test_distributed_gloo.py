"""world_size-2 gloo test of the photon-sharded SPPM iteration (CPU tensors only).  The
compute behind the sharding logic is the CPU oracle, which exposes the same methods as the
GPU context; the reduced image must equal the single-process pass."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytrace3_amd import distributed as D
from raytrace3_amd import scenes

SEED, ROUNDS, PER_ROUND = 5, 21, 100   # 2100 photons: an odd split point inside a round


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    from tests import oracle_lib as ol
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = scenes.scene_spheres_plane(True)
    cam = scenes.default_camera(20, 16)
    o = ol.OracleContext()
    o.set_scene(scene)
    pic = D.sharded_iteration(o, cam, SEED, 0, ROUNDS, PER_ROUND, 1, rank, world)
    t = torch.from_numpy(pic)            # CPU tensor: gloo never sees device memory
    D.all_reduce_image(t)
    if rank == 0:
        np.save(out_path, t.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_photon_shard_covers_the_pass():
    for total in (0, 1, 7, 2100, 10 ** 6 + 3):
        for world in (1, 2, 3, 8):
            spans = [D.photon_shard(total, r, world) for r in range(world)]
            assert spans[0][0] == 0
            assert sum(c for _, c in spans) == total
            for (b0, c0), (b1, _) in zip(spans, spans[1:]):
                assert b0 + c0 == b1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def test_two_rank_sharded_iteration_matches_single_process(tmp_path):
    from tests import oracle_lib as ol
    ol.load()
    out = str(tmp_path / "pic.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    reduced = np.load(out)
    o = ol.OracleContext()
    o.set_scene(scenes.scene_spheres_plane(True))
    o.eye_pass(scenes.default_camera(20, 16), SEED, 0)
    o.photon_pass(ROUNDS, PER_ROUND, SEED, 0)
    full = o.resolve()
    assert full.sum() > 0
    assert np.allclose(reduced, full, rtol=1e-12, atol=1e-15)


# ------------------------------------------------------------------ the host logic of bench.py's loop, on the CPU
class _HostContext:
    """The methods sharded_iteration uses, over the CPU oracle: resolve_device writes the float image where the given
    pointer says (a CPU tensor's storage here), synchronize is a no-op.  What is exercised is the host logic: shard
    bounds, the empty shard, the whole-pass normaliser, the reduction through host copies (via_host)."""

    def __init__(self, oracle):
        self.o = oracle
        self.calls = []

    def eye_pass(self, cam, seed, replica=0):
        self.o.eye_pass(cam, seed, replica)

    def photon_pass(self, rounds, per_round, seed, replica=0, photon_begin=0, photon_count=0):
        self.calls.append((photon_begin, photon_count))
        self.o.photon_pass(rounds, per_round, seed, replica, photon_begin=photon_begin, photon_count=photon_count)

    def resolve_device(self, ptr):
        import ctypes
        pic = self.o.resolve().astype(np.float32)
        ctypes.memmove(ptr, pic.ctypes.data, pic.nbytes)

    def synchronize(self):
        self.calls.append("sync")


def _host_logic_worker(rank, world, port, out_path, rounds, per_round):
    from tests import oracle_lib as ol
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene, cam = scenes.scene_spheres_plane(True), scenes.default_camera(20, 16)
    o = ol.OracleContext()
    o.set_scene(scene)
    ctx = _HostContext(o)
    pic = torch.zeros((16, 20, 3), dtype=torch.float32)
    D.sharded_iteration(ctx, cam, SEED, 0, rounds, per_round, 1, rank, world, pic_tensor=pic, via_host=True)
    begin, count = D.photon_shard(rounds * per_round, rank, world)
    # an empty shard still runs a (zero-length) pass, so that the resolve has the normaliser of the whole pass
    assert ctx.calls[0] == ((begin, count) if count > 0 else (rounds * per_round, 1)) and "sync" in ctx.calls
    if rank == 0:
        np.save(out_path, pic.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("rounds,per_round", [(21, 100), (1, 2)])
def test_three_rank_host_logic_with_an_uneven_and_an_empty_shard(tmp_path, rounds, per_round):
    """Three ranks: 2 100 photons split 700 / 700 / 700, and 2 photons split 1 / 1 / 0 (the third rank's shard is
    empty).  The partial float images, each resolved with the normaliser of the whole pass and reduced through host copies
    as ranks that share a GPU do, sum to the single-rank image."""
    from tests import oracle_lib as ol
    ol.load()
    out = str(tmp_path / "pic3.npy")
    mp.spawn(_host_logic_worker, args=(3, _free_port(), out, rounds, per_round), nprocs=3, join=True)
    reduced = np.load(out)
    o = ol.OracleContext()
    o.set_scene(scenes.scene_spheres_plane(True))
    o.eye_pass(scenes.default_camera(20, 16), SEED, 0)
    o.photon_pass(rounds, per_round, SEED, 0)
    full = o.resolve()
    assert np.allclose(reduced, full.astype(np.float32), rtol=1e-5, atol=1e-7 * max(full.mean(), 1e-30))
    if rounds > 1:
        assert full.sum() > 0


# ------------------------------------------------------------------ the same through the HIP path
def _gpu_worker(rank, world, port, out_path):
    """Two processes share the one GPU of the box; each traces its photon shard through libsppm_hip.so and
    hands gloo a HOST copy of its partial image (gloo gets CPU tensors only)."""
    from raytrace3_amd import SppmContext
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = scenes.scene_teapot(1)
    cam = scenes.default_camera(48, 36)
    g = SppmContext(0)
    g.set_scene(scene)
    pic = D.sharded_iteration(g, cam, SEED, 3, 60, PER_ROUND, 1, rank, world)   # host image of this rank's shard
    t = torch.from_numpy(np.ascontiguousarray(pic))
    D.all_reduce_image(t)
    if rank == 0:
        np.save(out_path, t.numpy())
    dist.barrier()
    g.close()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_sharded_iteration_on_the_gpu(tmp_path):
    from raytrace3_amd import SppmContext
    out = str(tmp_path / "pic_gpu.npy")
    mp.spawn(_gpu_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    reduced = np.load(out)
    g = SppmContext(0)
    g.set_scene(scenes.scene_teapot(1))
    g.eye_pass(scenes.default_camera(48, 36), SEED, 3)
    g.photon_pass(60, PER_ROUND, SEED, 3)
    full = g.resolve()
    g.close()
    assert full.sum() > 0
    # the same deposits, gathered in two shards and summed in fp32
    assert np.allclose(reduced, full, rtol=2e-4, atol=1e-6 * full.mean())


@pytest.mark.gpu
def test_all_reduce_ordered_by_the_contexts_stream():
    """What bench.py does on N GPUs, with the one rank this box has: the flux image is resolved on the context's
    own stream and all-reduced by RCCL (backend "nccl") under torch's handle of that stream, with no host-side
    wait between the two; two contexts alternate as in the benchmark.  The reduced images are the resolved ones."""
    from raytrace3_amd import SppmContext
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    ctxs = [SppmContext(0), SppmContext(0)]
    try:
        scene, cam = scenes.scene_teapot(1), scenes.default_camera(96, 72)
        streams = [torch.cuda.ExternalStream(c.stream_handle()) for c in ctxs]
        pics = [torch.zeros((72, 96, 3), dtype=torch.float32, device="cuda") for _ in ctxs]
        for c in ctxs:
            c.set_scene(scene)
        for i in range(6):
            k = i % 2
            # world = 2 would shard the photons; the collective path is taken by hand here with the whole pass
            ctxs[k].eye_pass(cam, SEED, i)
            ctxs[k].photon_pass(ROUNDS, PER_ROUND, SEED, i)
            ctxs[k].resolve_device(pics[k].data_ptr())
            with torch.cuda.stream(streams[k]):
                dist.all_reduce(pics[k], op=dist.ReduceOp.SUM)
        for c in ctxs:
            c.synchronize()
        torch.cuda.synchronize()
        for k, replica in ((0, 4), (1, 5)):
            ctxs[k].eye_pass(cam, SEED, replica)
            ctxs[k].photon_pass(ROUNDS, PER_ROUND, SEED, replica)
            want = ctxs[k].resolve()
            assert want.sum() > 0
            assert np.allclose(pics[k].cpu().numpy(), want, rtol=2e-4, atol=1e-6 * want.mean())
    finally:
        for c in ctxs:
            c.close()
        dist.destroy_process_group()
