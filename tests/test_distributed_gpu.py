"""The N>1 path with the real engine: two processes share the one GPU of the test box, each renders its tiles through the
C ABI, a gloo gather (staged through the host, because two ranks cannot form an RCCL group on one device) brings them to
rank 0, which assembles the frame on the GPU.  The result must be the single-process frame, byte for byte.  A group of one
rank, the most one GPU allows, passes the packed tiles through the RCCL gather on device buffers."""
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, S, B, SEED = 333, 190, 2, 4, 11


def worker(rank, world, port, result, in_flight=1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, ROOT)
        import raytracer_amd as rt
        from raytracer_amd.distributed import TileRenderer
        scene = rt.default_scene(0x5EED0001)
        engs = []
        for _ in range(in_flight):
            eng = rt.Engine(rt.default_config(width=W, height=H, samples_per_axis=S, bounces=B, device=0, tile_w=32, tile_h=32))
            eng.set_scene(scene)
            eng.set_camera(rt.camera_look((0.0, 3.0, -10.0), (0.0, 0.0, 1.0)))
            engs.append(eng)
        tr = TileRenderer(engs, "cuda:0", rank, world)
        for _ in range(2 * in_flight):             # every slot twice: buffers are reused between frames
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


def test_several_frames_in_flight():
    """two ranks with three frames in flight each (three engines, streams and buffer sets per rank)"""
    port = 29900 + (os.getpid() % 80)
    with mp.Manager() as m:
        result = m.dict()
        mp.spawn(worker, args=(2, port, result, 3), nprocs=2, join=True)
        assert result["equal"], "frame of a pipelined sequence differs from the single-process frame"
        assert result["lit"] == W * H


def rccl_worker(rank, world, port, result):
    import torch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        sys.path.insert(0, ROOT)
        import raytracer_amd as rt
        eng = rt.Engine(rt.default_config(width=W, height=H, samples_per_axis=S, bounces=B, device=0, tile_w=32, tile_h=32))
        eng.set_scene(rt.default_scene(0x5EED0001))
        eng.set_camera(rt.camera_look((0.0, 3.0, -10.0), (0.0, 0.0, 1.0)))
        shares = 2
        nb = eng.tile_bytes(shares)
        st = torch.cuda.current_stream().cuda_stream
        frame = torch.zeros(H * W * 4, dtype=torch.uint8, device="cuda:0")
        for _ in range(2):
            parts = []
            for r in range(shares):                # the two shares of a two-rank frame, each through a gather of its own
                mine = torch.zeros(nb, dtype=torch.uint8, device="cuda:0")
                eng.render(SEED, r, shares, mine.data_ptr(), st)
                out = [torch.empty_like(mine)]
                dist.gather(mine, out, dst=0)
                parts.append(out[0])
            gathered = torch.cat(parts)
            eng.assemble(gathered.data_ptr(), shares, frame.data_ptr(), st)
        torch.cuda.synchronize()
        got = frame.cpu().numpy().reshape(H, W, 4)
        ref = eng.render_frame(SEED)
        result["equal"] = bool(np.array_equal(got, ref))
    finally:
        dist.destroy_process_group()


def test_tiles_through_an_rccl_gather():
    port = 29990 + (os.getpid() % 150)
    with mp.Manager() as m:
        result = m.dict()
        mp.spawn(rccl_worker, args=(1, port, result), nprocs=1, join=True)
        assert result["equal"], "frame gathered through RCCL differs from the directly rendered frame"
