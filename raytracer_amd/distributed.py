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


class TileRenderer:
    """engine: raytracer_amd.Engine (or any object with tile_bytes / render / assemble and a cfg with width/height).
    group: a torch.distributed process group or None for the default; world size 1 needs no process group at all."""

    def __init__(self, engine, device, rank=0, world_size=1, group=None, display_rank=0):
        self.e = engine
        self.device = torch.device(device)
        self.rank, self.world, self.group, self.display = rank, world_size, group, display_rank
        W, H = engine.cfg.width, engine.cfg.height
        self.frame = torch.zeros(H * W * 4, dtype=torch.uint8, device=self.device) if rank == display_rank or world_size == 1 else None
        # a gloo group cannot gather device tensors: rehearsals of the multi-process path on GPUs stage through the host
        self.via_host = world_size > 1 and self.device.type == "cuda" and dist.get_backend(group) == "gloo"
        if world_size > 1:
            self.nb = engine.tile_bytes(world_size)
            self.mine = torch.zeros(self.nb, dtype=torch.uint8, device=self.device)
            self.gathered = torch.zeros(self.nb * world_size, dtype=torch.uint8, device=self.device) if rank == display_rank else None

        # Render, gather and assemble are ordered by ONE explicit stream: the engine gets its handle (a null handle
        # would mean "the engine's private stream", which torch knows nothing about), and the collective is issued with
        # that stream current, so RCCL's own stream waits for the render and the assemble waits for the gather.
        self.stream = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None

    def _stream(self):
        return self.stream.cuda_stream if self.stream is not None else None

    def render_frame(self, seed):
        """Enqueue one frame.  Returns the frame tensor (H*W*4 bytes) on the display rank, None elsewhere."""
        if self.stream is None:
            return self._render_frame(seed)
        with torch.cuda.stream(self.stream):
            return self._render_frame(seed)

    def _render_frame(self, seed):
        st = self._stream()
        if self.world == 1:
            self.e.render(seed, 0, 1, self.frame.data_ptr(), st)
            return self.frame
        self.e.render(seed, self.rank, self.world, self.mine.data_ptr(), st)
        if self.via_host:
            self.stream.synchronize()
            mine = self.mine.cpu()
            if self.rank == self.display:
                parts = [torch.empty_like(mine) for _ in range(self.world)]
                dist.gather(mine, parts, dst=self.display, group=self.group)
                self.gathered.copy_(torch.cat(parts))
                self.e.assemble(self.gathered.data_ptr(), self.world, self.frame.data_ptr(), st)
                return self.frame
            dist.gather(mine, None, dst=self.display, group=self.group)
            return None
        if self.rank == self.display:
            dist.gather(self.mine, list(self.gathered.chunk(self.world)), dst=self.display, group=self.group)
            self.e.assemble(self.gathered.data_ptr(), self.world, self.frame.data_ptr(), st)
            return self.frame
        dist.gather(self.mine, None, dst=self.display, group=self.group)
        return None

    def frame_numpy(self):
        if self.frame is None:
            return None
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        return self.frame.cpu().numpy().reshape(self.e.cfg.height, self.e.cfg.width, 4)
