"""One frame across the GPUs of a node: image-space tiles per rank, tiles gathered to the display rank.

Replaces the reference's 7-way split of the frame buffer across std::threads (FrameManager.cpp:17-21,80-82).  One process
per GPU.  The scene is replicated; the frame is cut into tile_w x tile_h tiles numbered row-major and rank r renders
tiles r, r+n, r+2n, ... (interleaved, so cheap sky tiles and expensive geometry tiles spread evenly).  Each rank's tiles
are packed into one buffer of equal size on every rank; a single gather (RCCL over xGMI with the "nccl" backend, gloo
for CPU rehearsals) brings them to the display rank, which scatters them into the [A,B,G,R] frame.  No other exchange
happens: pixels are independent because every (pixel, sample) owns its RNG stream.
"""
import torch
import torch.distributed as dist


class _Slot:
    """what one frame in flight needs of its own: an engine (queues, frame state), a stream, the rank's packed tiles and,
    on the display rank, the gathered tiles and the assembled frame"""

    def __init__(self, engine, device, rank, world_size, display_rank):
        self.e = engine
        W, H = engine.cfg.width, engine.cfg.height
        self.frame = torch.zeros(H * W * 4, dtype=torch.uint8, device=device) if rank == display_rank or world_size == 1 else None
        self.mine = self.gathered = None
        if world_size > 1:
            nb = engine.tile_bytes(world_size)
            self.mine = torch.zeros(nb, dtype=torch.uint8, device=device)
            self.gathered = torch.zeros(nb * world_size, dtype=torch.uint8, device=device) if rank == display_rank else None
        # Render, gather and assemble are ordered by ONE explicit stream: the engine gets its handle (a null handle
        # would mean "the engine's private stream", which torch knows nothing about), and the collective is issued with
        # that stream current, so RCCL's own stream waits for the render and the assemble waits for the gather.
        self.stream = torch.cuda.Stream(device) if device.type == "cuda" else None


class TileRenderer:
    """engine: raytracer_amd.Engine (or any object with tile_bytes / render / assemble and a cfg with width/height), or a
    list of them with the same scene, camera and configuration: then that many frames are in flight, the frames of a
    sequence taking the engines in turn, each on its own stream.  A rank's share of a frame runs passes that are too
    small to fill its GPU (the deep bounces of an eighth of a frame are a few thousand rays each); with two or three
    frames in flight those passes run beside the large passes of the next frame.  Measured on one MI355X with rank 0's
    share of the 1920 x 1080 x 64 spp frame: 8 ranks 4.80 ms per frame with one frame in flight, 3.86 with three; 4 ranks
    7.72 and 6.99; 2 ranks and the whole frame gain nothing (two full-size persistent grids only take turns).
    group: a torch.distributed process group or None for the default; world size 1 needs no process group at all."""

    def __init__(self, engine, device, rank=0, world_size=1, group=None, display_rank=0):
        engines = list(engine) if isinstance(engine, (list, tuple)) else [engine]
        self.e = engines[0]
        self.device = torch.device(device)
        self.rank, self.world, self.group, self.display = rank, world_size, group, display_rank
        self.slots = [_Slot(e, self.device, rank, world_size, display_rank) for e in engines]
        self.turn = 0
        self.frame = self.slots[0].frame        # the frame most recently queued (display rank)
        # a gloo group cannot gather device tensors: rehearsals of the multi-process path on GPUs stage through the host
        self.via_host = world_size > 1 and self.device.type == "cuda" and dist.get_backend(group) == "gloo"

    def render_frame(self, seed):
        """Enqueue one frame.  Returns the frame tensor (H*W*4 bytes) on the display rank, None elsewhere; with several
        frames in flight the tensor belongs to the frame's slot and is overwritten when its turn comes again.  The frame
        is rendered on a stream of the slot's own; the caller's current stream is made to wait for it, so work queued on
        that stream afterwards sees the finished frame."""
        slot = self.slots[self.turn]
        self.turn = (self.turn + 1) % len(self.slots)
        if slot.stream is None:
            out = self._render_frame(slot, seed)
        else:
            with torch.cuda.stream(slot.stream):
                out = self._render_frame(slot, seed)
            # the frame is written on the slot's stream: whatever the caller queues next on ITS current stream (a copy
            # of the returned tensor, say) waits for it on the GPU; the host does not wait, and the other slots' streams
            # are not held up
            torch.cuda.current_stream(self.device).wait_stream(slot.stream)
        if out is not None:
            self.frame = out
        return out

    def _render_frame(self, slot, seed):
        st = slot.stream.cuda_stream if slot.stream is not None else None
        if self.world == 1:
            slot.e.render(seed, 0, 1, slot.frame.data_ptr(), st)
            return slot.frame
        slot.e.render(seed, self.rank, self.world, slot.mine.data_ptr(), st)
        if self.via_host:
            slot.stream.synchronize()
            mine = slot.mine.cpu()
            if self.rank == self.display:
                parts = [torch.empty_like(mine) for _ in range(self.world)]
                dist.gather(mine, parts, dst=self.display, group=self.group)
                slot.gathered.copy_(torch.cat(parts))
                slot.e.assemble(slot.gathered.data_ptr(), self.world, slot.frame.data_ptr(), st)
                return slot.frame
            dist.gather(mine, None, dst=self.display, group=self.group)
            return None
        if self.rank == self.display:
            dist.gather(slot.mine, list(slot.gathered.chunk(self.world)), dst=self.display, group=self.group)
            slot.e.assemble(slot.gathered.data_ptr(), self.world, slot.frame.data_ptr(), st)
            return slot.frame
        dist.gather(slot.mine, None, dst=self.display, group=self.group)
        return None

    def frame_numpy(self):
        """the frame most recently queued, on the host (display rank)"""
        if self.frame is None:
            return None
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        return self.frame.cpu().numpy().reshape(self.e.cfg.height, self.e.cfg.width, 4)
