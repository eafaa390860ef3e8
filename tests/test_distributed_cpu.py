"""The N>1 path on CPU: two gloo ranks run raytracer_amd.distributed.TileRenderer — tile ownership, packed per-rank
buffers, one gather to the display rank, scatter into the frame.  The GPU engine is replaced by a stand-in that fills the
rank's tiles from the CPU oracle (this is a test: the product never renders on the CPU); the tile layout it implements is
the one include/rt_engine.h documents for rt_engine_render / rt_engine_assemble."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

W, H, S, B, TW, TH, SEED = 150, 70, 2, 3, 32, 16, 5


class Cfg:
    width, height, samples_per_axis, bounces, tile_w, tile_h = W, H, S, B, TW, TH


def tile_pixels(rank, n_ranks):
    """(slot, y, x) of every frame pixel rank `rank` owns; slot = position in its packed buffer"""
    tx_n, ty_n = -(-W // TW), -(-H // TH)
    out = []
    for t, tile in enumerate(range(rank, tx_n * ty_n, n_ranks)):
        ty, tx = divmod(tile, tx_n)
        for iy in range(TH):
            for ix in range(TW):
                y, x = ty * TH + iy, tx * TW + ix
                if y < H and x < W:
                    out.append(((t * TH + iy) * TW + ix, y, x))
    return out


class OracleTileEngine:
    """stand-in with the engine's tile interface, rendering through the oracle"""

    def __init__(self, frame_rgba):
        self.cfg = Cfg
        self.full = frame_rgba            # the oracle's whole frame; a rank only copies the pixels it owns

    def tile_bytes(self, n_ranks):
        tiles = -(-W // TW) * -(-H // TH)
        return -(-tiles // n_ranks) * TW * TH * 4

    @staticmethod
    def _view(ptr, n):
        return np.ctypeslib.as_array((ctypes.c_uint8 * n).from_address(ptr))

    def render(self, seed, rank, n_ranks, d_out, stream):
        assert seed == SEED
        if n_ranks == 1:
            self._view(d_out, W * H * 4)[:] = self.full.reshape(-1)
            return
        buf = self._view(d_out, self.tile_bytes(n_ranks)).reshape(-1, 4)
        buf[:] = 0
        for slot, y, x in tile_pixels(rank, n_ranks):
            buf[slot] = self.full[y, x]

    def assemble(self, d_gathered, n_ranks, d_out, stream):
        nb = self.tile_bytes(n_ranks)
        src = self._view(d_gathered, nb * n_ranks).reshape(n_ranks, -1, 4)
        dst = self._view(d_out, W * H * 4).reshape(H, W, 4)
        for r in range(n_ranks):
            for slot, y, x in tile_pixels(r, n_ranks):
                dst[y, x] = src[r, slot]


def oracle_frame():
    sys.path.insert(0, ROOT)
    from oracle import orcbind as O
    import raytracer_amd as rt
    E = rt.engine
    a = os.path.join(E.ASSET_DIR, "src", "stb_image")
    s = O.Scene(O.MATH_PORTABLE)
    s.default_scene(E.load_obj(os.path.join(a, "tree.obj")), E.load_png(os.path.join(a, "tree_texture.png"), 4)[0],
                    E.load_png(os.path.join(a, "skybox12.png"), 3)[0])
    s.build(0x5EED0001)
    return s.render(O.camera((0, 3, -10), (0, 0, 1)), W, H, S, B, SEED, threads=2, want=("rgba",))["rgba"]


def worker(rank, world, port, result, in_flight=1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, ROOT)
        from raytracer_amd.distributed import TileRenderer
        full = oracle_frame()
        tr = TileRenderer([OracleTileEngine(full) for _ in range(in_flight)], "cpu", rank, world)
        for _ in range(2 * in_flight - 1):     # every slot takes a frame, the first ones twice
            frame = tr.render_frame(SEED)
        if rank == 0:
            got = tr.frame_numpy()
            result["equal"] = bool(np.array_equal(got, full))
            result["nonzero"] = int((got[..., 0] == 255).sum())
        else:
            assert frame is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_tile_partition_and_gather_gloo(world):
    port = 29600 + world + (os.getpid() % 200)
    with mp.Manager() as m:
        result = m.dict()
        mp.spawn(worker, args=(world, port, result), nprocs=world, join=True)
        assert result["equal"], "gathered + assembled frame differs from the single-process frame"
        assert result["nonzero"] == W * H


def test_several_frames_in_flight_gloo():
    """three slots (engine, buffers and stream each) take the frames of a sequence in turn: the gathers of all ranks
    must still pair up frame by frame"""
    port = 29700 + (os.getpid() % 200)
    with mp.Manager() as m:
        result = m.dict()
        mp.spawn(worker, args=(2, port, result, 3), nprocs=2, join=True)
        assert result["equal"] and result["nonzero"] == W * H


def test_single_rank_needs_no_process_group():
    sys.path.insert(0, ROOT)
    from raytracer_amd.distributed import TileRenderer
    full = oracle_frame()
    tr = TileRenderer(OracleTileEngine(full), "cpu")
    tr.render_frame(SEED)
    assert np.array_equal(tr.frame_numpy(), full)


def test_tiles_cover_the_frame_exactly_once():
    for n in (1, 2, 3, 8):
        seen = np.zeros((H, W), np.int32)
        for r in range(n):
            for _, y, x in tile_pixels(r, n):
                seen[y, x] += 1
        assert (seen == 1).all()
