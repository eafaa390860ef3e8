"""The N>1 path with the real engine: two processes share the one GPU of the test box, each renders its tiles through the
C ABI, a gloo gather (staged through the host, because two ranks cannot form an RCCL group on one device) brings them to
rank 0, which assembles the frame on the GPU.  The result must be the single-process frame, byte for byte."""
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, S, B, SEED = 333, 190, 2, 4, 11


def worker(rank, world, port, result):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, ROOT)
        import raytracer_amd as rt
        from raytracer_amd.distributed import TileRenderer
        scene = rt.default_scene(0x5EED0001)
        eng = rt.Engine(rt.default_config(width=W, height=H, samples_per_axis=S, bounces=B, device=0, tile_w=32, tile_h=32))
        eng.set_scene(scene)
        eng.set_camera(rt.camera_look((0.0, 3.0, -10.0), (0.0, 0.0, 1.0)))
        tr = TileRenderer(eng, "cuda:0", rank, world)
        for _ in range(2):                         # twice: buffers are reused between frames
            frame = tr.render_frame(SEED)
        if rank == 0:
            got = tr.frame_numpy()
            ref = eng.render_frame(SEED)           # the whole frame on one rank
            result["equal"] = bool(np.array_equal(got, ref))
            result["lit"] = int((got[..., 0] == 255).sum())
        else:
            assert frame is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_two_processes_one_frame(world):
    port = 29800 + world + (os.getpid() % 150)
    with mp.Manager() as m:
        result = m.dict()
        mp.spawn(worker, args=(world, port, result), nprocs=world, join=True)
        assert result["equal"], "gathered + assembled frame differs from the single-process frame"
        assert result["lit"] == W * H
