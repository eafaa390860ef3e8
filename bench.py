#!/usr/bin/env python3
"""Headline benchmark: whole-job Mrays/s of the per-pixel path on N MI355X (BASELINE.json `metric`).

A step = one frame of the workload: generate -> (intersect -> shade -> compact) x (bounces+1) -> resolve, with the frame
cut into image tiles across the N ranks and the tiles gathered to rank 0 over RCCL.  A ray = one closest-hit query
(one World::GetRayColor invocation of the reference).  Scene, camera and queues are resident in HBM before the timed
region; nothing is read from the host inside it.

The headline line is the C2 workload (default scene, 1920x1080, 64 spp, 8 bounces).  The default invocation then runs,
in the same process, the traversal-stress scene at its full size (C3; with N > 1 ranks this is C4), the reference's own
settings (C1, with the reference's FrameManager frame on the host beside it) and the interactive camera path (C5, every
frame waited for and copied to page-locked host memory) and appends them as "also".

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c1|c3|c5] [--no-cpu-baseline] [--no-also]
    N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# The reference's tree depends on the random axis it draws per node (Object.cpp:191), so its cost per ray — and with it the
# roofline fraction, which prices a ray by the reference algorithm's work — depends on the build key.  Under 0x5EED0001
# the tree's mesh is the only object of its node and the reference walks it twice per ray (Object.cpp:198-200); the
# engine walks it once, which flatters that key.  The headline is therefore measured under the key whose figures are the
# median of the three, and the line reports all three (build_key_spread, roofline.frac_by_build_key / frac_min / frac_max).
SPREAD_KEYS = (0x5EED0001, 0x00C0FFEE, 0x12345678)
BUILD_KEY = 0x12345678
SEED = 7
CAM_DEFAULT = ((0.0, 3.0, -10.0), (0.0, 0.0, 1.0))          # World.cpp:48
CAM_SYNTHETIC = ((0.0, 20.0, -150.0), (0.0, 0.0, 1.0))
WORKLOADS = {
    # name: (description, W, H, S, B)
    "c1": ("default scene 1280x720, 9 spp, 3 bounces (the reference's Constants.h)", 1280, 720, 3, 3),
    "c2": ("default scene 1920x1080, 64 spp, 8 bounces", 1920, 1080, 8, 8),
    "c3": ("synthetic 100k sphere/box scene 3840x2160, 256 spp, 8 bounces", 3840, 2160, 16, 8),
    # interactive sequence: the camera moves every frame as the front end's input handler would move it (App.cpp:98,
    # 114-137: mouse (3, 1) per frame, key W for frames 0-119, D for 120-239); every frame is waited for and handed to the
    # host the way the front end takes it (FrameManager.cpp:60-71 + App.cpp:75), so a step's time is what a viewer sees
    "c5": ("default scene 1920x1080, 9 spp, 3 bounces, 240-frame camera path, one frame in flight, frame copied to the host", 1920, 1080, 3, 3),
}
# FLOPs the reference algorithm spends per counted event (sqrt and divide count 1, compares and libm calls excluded).
# The events of the closest-hit query (and, for primary rays, camera-ray generation) are the work of intersect_kernel,
# the dominant kernel; finishing the hit and the material switch belong to shade_kernel.
FLOP_INTERSECT = {"aabb": 16, "sph_test": 17, "cube_test": 12, "plane_test": 3, "tri_test": 52, "tri_uv": 64, "inst_enter": 21,
                  "samples": 45}
FLOP_SHADE = {"sph_hit": 25, "cube_hit": 21, "plane_hit": 6, "inst_hit": 21, "shade": 30}
# bytes of scene data the reference algorithm touches per counted event: 32 B per node visit, 16 B per sphere, 24 B per
# cube, 36 B per triangle (+40 B on the uv path).  The engine serves them from its LDS copy of the scene when it fits.
SCENE_BYTE_WEIGHTS = {"aabb": 32, "sph_test": 16, "cube_test": 24, "tri_test": 36, "tri_uv": 40}
# bytes intersect_kernel streams through HBM per ray: 32 B ray record read (bounce rays; primary rays are generated in
# registers) and 16 B hit record written
RAY_READ_BYTES, HIT_WRITE_BYTES = 32, 16
PEAK_FP32_TFLOPS = 157.3     # MI355X vector FP32 with packed FMA; unfused scalar-per-lane code can reach 1/4 of it
PEAK_HBM_GBS = 8000.0
# per-ray figures of the reference algorithm measured by the oracle's event counters over the CPU sample of each workload;
# used when the CPU leg is skipped (--no-cpu-baseline, N > 1)
FALLBACK_FLOP_PER_RAY = {"c1": 527.0, "c2": 527.0, "c3": 5960.0, "c5": 527.0}
FALLBACK_SCENE_BYTE_PER_RAY = {"c1": 800.0, "c2": 800.0, "c3": 11500.0, "c5": 800.0}
FALLBACK_BOX_PER_RAY = {"c1": 20.2, "c2": 20.2, "c3": 339.0, "c5": 20.2}
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r3_c2_pmc_summary.txt")


def build_scene(rt, workload, key=BUILD_KEY):
    """(scene, camera, seconds inside rt_scene_build): the tree construction of BVH_Node::BVH_Node (Object.cpp:181-225)
    plus flattening; asset decoding and the per-object calls from Python are not in the figure"""
    if workload in ("c1", "c2", "c5"):
        s = rt.Scene()
        s.default_objects()
        t0 = time.perf_counter()
        s.build(key)
        return s, rt.camera_look(*CAM_DEFAULT), time.perf_counter() - t0
    from raytracer_amd.synthetic import synthetic_tables

    class Timed:                                     # build_from_tables ends with S.build(key): time just that call
        def __init__(self, s):
            self.s, self.seconds = s, 0.0

        def __getattr__(self, name):
            return getattr(self.s, name)

        def build(self, k):
            t0 = time.perf_counter()
            self.s.build(k)
            self.seconds = time.perf_counter() - t0
    from raytracer_amd.synthetic import build_from_tables
    E = rt.engine
    sky, _ = E.load_png(os.path.join(E.ASSET_DIR, "src", "stb_image", "skybox12.png"), 3)
    obj, tex, mat = synthetic_tables(100000)
    t = Timed(rt.Scene())
    build_from_tables(t, tex, mat, obj, [[0.0, len(mat) - 1]], sky, key + 2)
    return t.s, rt.camera_look(*CAM_SYNTHETIC), t.seconds


def oracle_scene(workload, key=BUILD_KEY):
    from oracle import orcbind as O
    import raytracer_amd as rt
    E = rt.engine
    asset = os.path.join(E.ASSET_DIR, "src", "stb_image")
    sky, _ = E.load_png(os.path.join(asset, "skybox12.png"), 3)
    s = O.Scene(O.MATH_LIBM)
    if workload in ("c1", "c2", "c5"):
        verts = E.load_obj(os.path.join(asset, "tree.obj"))
        tex, _ = E.load_png(os.path.join(asset, "tree_texture.png"), 4)
        s.default_scene(verts, tex, sky)
        s.build(key)
        return s, O.camera(*CAM_DEFAULT)
    from raytracer_amd.synthetic import synthetic_tables, build_from_tables
    obj, tex_t, mat_t = synthetic_tables(100000)
    build_from_tables(s, tex_t, mat_t, obj, [[0.0, len(mat_t) - 1]], sky, key + 2)
    return s, O.camera(*CAM_SYNTHETIC)


def per_ray_figures(c, rep=None):
    """per-ray figures of the reference algorithm from the oracle's event counters `c`; `rep` = the part of them spent on
    second visits of leaves listed as both children of a node (Object.cpp:198-200), which the engine does not repeat"""
    flops = sum(FLOP_INTERSECT[k] * c[k] for k in FLOP_INTERSECT)
    flops_shade = sum(FLOP_SHADE[k] * c[k] for k in FLOP_SHADE)
    nbytes = sum(SCENE_BYTE_WEIGHTS[k] * c[k] for k in SCENE_BYTE_WEIGHTS)
    out = {"intersect_flop_per_ray": flops / c["rays"], "shade_flop_per_ray": flops_shade / c["rays"],
           "scene_byte_per_ray": nbytes / c["rays"], "rays_per_sample": c["rays"] / c["samples"],
           "box_tests_per_ray": c["aabb"] / c["rays"]}
    if rep is not None:
        # everything but the slab tests, counted once per leaf: what the engine's own slab-test count is added to for
        # roofline.frac_executed
        rest = sum(FLOP_INTERSECT[k] * (c[k] - rep[k]) for k in FLOP_INTERSECT if k != "aabb")
        out["intersect_flop_per_ray_without_box_and_repeats"] = rest / c["rays"]
        out["box_tests_per_ray_without_repeats"] = (c["aabb"] - rep["aabb"]) / c["rays"]
    return out


def cpu_baseline(workload, W, H, S, B, threads):
    """The reference's CPU path on the host cores, on a bounded sample of the workload.  Two legs over the same pixels and
    seeds: the CPU restatement (oracle/, libm math like the reference; it also counts the events the roofline is priced
    with) and, where it travelled with the tree, the compiled reference itself (oracle/_ref/libref.so: the reference's own
    sources behind a harness) — the stated baseline when it is there (kind "reference"), with the port beside it.  The two
    follow the same paths bit for bit (tests/test_oracle_vs_reference.py), so the port's ray count is the reference's."""
    s, cam = oracle_scene(workload)
    # sized for seconds on 16 host threads: the whole frame for c1/c2/c5; for c3 32 one-row segments of 240 pixels spread
    # over the frame's height and width, all 256 samples of each pixel
    n_strips, rows = {"c1": (1, H), "c2": (1, H), "c3": (32, 1), "c5": (1, H)}[workload]
    rects = []
    for k in range(n_strips):
        y0 = int((k + 0.5) * H / n_strips) if rows < H else 0
        x0, x1 = (0, W) if workload != "c3" else (k * W // n_strips, k * W // n_strips + W // 16)
        rects.append((x0, y0, x1, min(H, y0 + rows)))
    px = sum((r[2] - r[0]) * (r[3] - r[1]) for r in rects)
    s.counters(reset=True)
    t0 = time.perf_counter()
    for r in rects:
        s.render(cam, W, H, S, B, SEED, rect=r, threads=threads, want=("rgba",))
    dt = time.perf_counter() - t0
    c, rep = s.counters(), s.repeat_counters()
    # the reference's own thread count (THREAD_COUNT 7, Constants.h:6) on a smaller sample: 16 strips of 8 rows (c3:
    # every fourth segment of the sample above)
    before = c["rays"]
    t1 = time.perf_counter()
    for r in (rects[::4] if workload == "c3" else [(0, int((k + 0.5) * H / 16), W, min(H, int((k + 0.5) * H / 16) + 8)) for k in range(16)]):
        s.render(cam, W, H, S, B, SEED, rect=r, threads=7, want=("rgba",))
    v7 = (s.counters()["rays"] - before) / (time.perf_counter() - t1) / 1e6
    what = (f"{n_strips} strips x {rows} rows = {px} pixels of the {W}x{H} frame, all {S * S} spp and {B} bounces, {c['rays']} rays; "
            "contiguous pixel ranges per thread as FrameManager.cpp:80-82")
    port = {"value": c["rays"] / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port", "seconds": dt, "value_7_threads": v7}
    out = dict(port, sample=what + f"; {dt:.1f} s")
    if workload != "c3":     # the harness builds the reference's World, i.e. the default scene; the 100 000-object tree would take it minutes
        try:
            from oracle.refbind import Ref, available
            if available():
                R = Ref()
                w = R.world(BUILD_KEY)
                t0 = time.perf_counter()
                for r in rects:
                    R.render(w, W, H, S, B, SEED, rect=r, threads=threads, want=("rgba",))
                dtr = time.perf_counter() - t0
                R.free(w)
                out = {"value": c["rays"] / dtr / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "reference", "seconds": dtr,
                       "sample": what + f"; {dtr:.1f} s in the compiled reference (rays as counted by the port on the same pixels and seeds)",
                       "port": port}
        except Exception as ex:      # the compiled reference is optional test infrastructure
            out["reference_error"] = str(ex)[:200]
    out.update(per_ray_figures(c, rep))
    return out


def reference_c1_frame(gpu_rays):
    """BASELINE config 1: one headless frame through the reference's OWN scheduler (FrameManager.cpp: 7 std::threads,
    stride-37 pixel order) at its Constants.h settings, timed as its LOG_BENCHMARK print times it (FrameManager.cpp:27-29,
    50-57: from the threads' start to the last worker's exit)"""
    try:
        from oracle.refbind import Ref, available
        if not available():
            return {"error": "oracle/_ref/libref.so is not here"}
        R = Ref()
        i4, _ = R.constants()
        secs = [R.framemanager_frame(BUILD_KEY)[0] for _ in range(2)]
        best = min(secs)
        return {"frame_calculating_time_s": best, "runs_s": secs, "threads": 7, "settings": [int(v) for v in i4],
                "mrays_per_s": gpu_rays / best / 1e6,
                "note": "rays = the GPU frame's count for the same scene and settings (the reference has no ray counter; its "
                        "std::rand streams differ from the keyed ones, so its own count differs by a fraction of a per cent)"}
    except Exception as ex:
        return {"error": str(ex)[:200]}


def key_spread(rt, eng, tr, W, H, S, B, with_cpu):
    """the headline workload under three build keys: the reference's random-axis tree makes the cost per ray depend on the
    key, so one key's Mrays/s (and roofline fraction) is one sample of a spread"""
    import torch
    rows = []
    for key in SPREAD_KEYS:
        scene, cam, _ = build_scene(rt, "c2", key)
        eng.set_scene(scene)
        eng.set_camera(cam)
        tr.render_frame(SEED)
        torch.cuda.synchronize()
        eng.time_kernels(True)
        t0 = time.perf_counter()
        for _ in range(3):
            tr.render_frame(SEED)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        kern_ms, _ = eng.kernel_times()
        eng.time_kernels(False)
        st = eng.stats()
        row = {"build_key": f"0x{key:08X}", "ms_per_frame": ms, "mrays_per_s": st["rays"] / ms / 1e3,
               "intersect_kernel_ms": kern_ms, "executed_box_tests_per_ray": st["box_tests"] / st["rays"]}
        if with_cpu:
            s, ocam = oracle_scene("c2", key)
            s.counters(reset=True)
            for k in range(8):
                y0 = int((k + 0.5) * H / 8)
                s.render(ocam, W, H, S, B, SEED, rect=(0, y0, W, y0 + 2), threads=with_cpu, want=("rgba",))
            f = per_ray_figures(s.counters(), s.repeat_counters())
            tf = f["intersect_flop_per_ray"] * st["rays"] / (kern_ms / 1e3) / 1e12
            # under 0x5EED0001 the tree (a mesh instance) is the only object of its node, so the reference walks the
            # whole mesh twice per ray (Object.cpp:198-200: left == right) and the algorithmic figure counts both walks;
            # the engine walks it once, which is why its executed count is far below the algorithmic one for that key.
            # frac_executed prices a ray by the slab tests the engine executed plus the other tests counted once per leaf
            fx = 16.0 * row["executed_box_tests_per_ray"] + f["intersect_flop_per_ray_without_box_and_repeats"]
            row.update({"flop_per_ray": f["intersect_flop_per_ray"], "box_tests_per_ray": f["box_tests_per_ray"],
                        "roofline_achieved_tflops": tf, "roofline_frac": tf / PEAK_FP32_TFLOPS,
                        "flop_per_ray_executed": fx, "roofline_frac_executed": fx * st["rays"] / (kern_ms / 1e3) / 1e12 / PEAK_FP32_TFLOPS})
        rows.append(row)
    return rows


def pmc_summary(path, kernel):
    """(source hash, HBM bytes per frame, active-lane fractions) of one kernel from a summary written by
    tools/pmc_summary.py (separate rocprofv3 --pmc runs of this same command): FETCH_SIZE and WRITE_SIZE are in units of
    1 KiB, and on gfx950 FETCH_SIZE counts half of the bytes read, so  bytes = 1024 * (2 * FETCH_SIZE + WRITE_SIZE),
    summed over the kernel's template variants."""
    try:
        lines = open(path).read().splitlines()
    except OSError:
        return None, None, None
    src, total, inside, seen, lanes, name = None, 0.0, False, False, {}, None
    for ln in lines:
        if ln.startswith("source_hash:"):
            src = ln.split(":", 1)[1].strip()
        elif not ln.startswith(" "):
            inside = ln.startswith(kernel)
            name = ln.split("  dispatches")[0] if inside else None
        elif inside:
            f = ln.split()
            if f[0] == "FETCH_SIZE":
                total += 2048.0 * float(f[1]); seen = True
            elif f[0] == "WRITE_SIZE":
                total += 1024.0 * float(f[1]); seen = True
            elif f[:2] == ["active-lane", "fraction"]:
                lanes[name] = float(f[2])
    return src, (total if seen else None), (lanes or None)


class Job:
    """process-group plumbing shared by the workloads of one invocation"""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}: launch N>1 with torch.distributed.run --nproc-per-node N")
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the HIP engine is the only compute path")
        self.rehearsal = args.rehearse_on_one_gpu
        if self.rehearsal:
            self.local_rank = 0
        torch.cuda.set_device(self.local_rank)
        self.device = torch.device("cuda", self.local_rank)
        self.red_device = torch.device("cpu") if self.rehearsal else self.device     # where the timing reductions live
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.rehearsal:
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            else:
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.device)

    def barrier(self):
        self.torch.cuda.synchronize(self.device)
        if self.world > 1:
            self.dist.barrier()
            self.torch.cuda.synchronize(self.device)

    def reduce(self, values, op):
        t = self.torch.tensor(values, dtype=self.torch.float64, device=self.red_device)
        if self.world > 1:
            self.dist.all_reduce(t, op=getattr(self.dist.ReduceOp, op))
        return [float(v) for v in t]

    def gather_all(self, values):
        """every rank's values, as a list per value (rank order)"""
        t = self.torch.tensor(values, dtype=self.torch.float64, device=self.red_device)
        if self.world == 1:
            return [[float(v)] for v in t]
        parts = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(parts, t)
        return [[float(p[i]) for p in parts] for i in range(len(values))]

    def close(self):
        if self.world > 1:
            self.dist.barrier()
            self.dist.destroy_process_group()


def run_workload(job, rt, workload, steps, warmup, batch_paths=0, in_flight=1):
    """warm up, then time `steps` frames between barriers; returns this rank's measurements plus the reductions.
    in_flight > 1: that many engines take the frames in turn (raytracer_amd.distributed.TileRenderer); the figures of
    one frame (rays, kernel times) are the first engine's"""
    from raytracer_amd.distributed import TileRenderer
    desc, W, H, S, B = WORKLOADS[workload]
    scene, cam, build_s = build_scene(rt, workload)
    if job.world > 1 and not batch_paths:
        # the queues of an engine are sized for one batch: a rank only ever renders its share of the tiles, so a batch of
        # that size (not of the whole frame) keeps its passes in one piece and the allocation at 1/world
        c0 = rt.default_config(width=W, height=H, samples_per_axis=S, bounces=B)
        tiles = -(-W // c0.tile_w) * -(-H // c0.tile_h)
        batch_paths = min(-(-tiles // job.world) * c0.tile_w * c0.tile_h * S * S, 1 << 27)
    engs = []
    for _ in range(max(1, in_flight)):
        eng = rt.Engine(rt.default_config(width=W, height=H, samples_per_axis=S, bounces=B, device=job.local_rank,
                                          max_batch_paths=batch_paths))
        eng.set_scene(scene)
        eng.set_camera(cam)
        engs.append(eng)
    eng = engs[0]
    tr = TileRenderer(engs, job.device, job.rank, job.world)
    if len(engs) > 1:
        # set-up, not a step: every engine renders its share once (no gather), so that none of them meets its first
        # launch inside the timed region when there are fewer warm-up steps than engines
        import torch
        scratch = torch.zeros(max(eng.tile_bytes(job.world), W * H * 4), dtype=torch.uint8, device=job.device)
        for e_ in engs:
            e_.render(SEED, job.rank, job.world, scratch.data_ptr())
            e_.sync()
        del scratch
    frame_ms = []
    for _ in range(warmup):
        tr.render_frame(SEED)
    job.barrier()
    eng.time_kernels(True)
    t0 = time.perf_counter()
    if workload == "c5":
        host_frame = None
        for f in range(steps):
            cam = rt.camera_input(cam, 3.0, 1.0, w=int(f % 240 < 120), d=int(f % 240 >= 120))
            eng.set_camera(cam)
            t1 = time.perf_counter()
            if job.world == 1:
                # one GPU: the engine's own frame buffer, handed over through its page-locked host copy
                # (rt_engine_map_frame: the copy runs on the render's stream right behind it)
                eng.render(SEED + f)
                eng.map_frame(wait=True)
            else:
                frame = tr.render_frame(SEED + f)
                job.barrier()
                if frame is not None:
                    if host_frame is None:       # page-locked, as the one-GPU path's rt_engine_map_frame buffer is
                        host_frame = job.torch.empty(frame.numel(), dtype=frame.dtype, pin_memory=True)
                    host_frame.copy_(frame, non_blocking=True)
                    job.torch.cuda.synchronize(job.device)
            frame_ms.append((time.perf_counter() - t1) * 1e3)
    else:
        for _ in range(steps):
            tr.render_frame(SEED)
    job.barrier()
    dt = time.perf_counter() - t0
    kern_ms, kern_launches = eng.kernel_times()
    eng.time_kernels(False)
    st = eng.stats()                       # this rank's share of one frame (every step renders the same frame)
    one_ms = None
    if len(engs) > 1:
        # beside the throughput with several frames in flight: the time of a frame that has the GPUs to itself
        tr1 = TileRenderer(eng, job.device, job.rank, job.world)
        tr1.render_frame(SEED)
        job.barrier()
        t1 = time.perf_counter()
        for _ in range(5):
            tr1.render_frame(SEED)
            job.barrier()
        one_ms = job.reduce([(time.perf_counter() - t1) / 5 * 1e3], "MAX")[0]
    dt = job.reduce([dt], "MAX")[0]
    rays_frame, paths_frame, box_frame = job.reduce([st["rays"], st["paths"], st["box_tests"]], "SUM")
    per_rank = job.gather_all([st["rays"], kern_ms, kern_launches])
    return {"desc": desc, "W": W, "H": H, "S": S, "B": B, "eng": eng, "engs": engs, "tr": tr, "dt": dt, "steps": steps, "frame_ms": frame_ms,
            "kern_ms": max(per_rank[1]), "kern_launches": kern_launches, "stats": st, "rays_frame": rays_frame,
            "paths_frame": paths_frame, "box_frame": box_frame, "per_rank": per_rank, "scene_build_s": build_s,
            "one_in_flight_ms": one_ms}


def c5_on_one_gpu(job, rt, steps):
    """C5 on one GPU without the tile renderer in between: render into the engine's own frame buffer and take the frame
    from its page-locked host copy, as the front-end adapter does"""
    desc, W, H, S, B = WORKLOADS["c5"]
    scene, cam, _ = build_scene(rt, "c5")
    eng = rt.Engine(rt.default_config(width=W, height=H, samples_per_axis=S, bounces=B, device=job.local_rank))
    eng.set_scene(scene)
    eng.set_camera(cam)
    for _ in range(3):
        eng.render(SEED)
        eng.map_frame(wait=True)
    render_ms, total_ms, rays, kern_ms = [], [], [], []
    cam0 = cam
    for f in range(steps):
        cam = rt.camera_input(cam, 3.0, 1.0, w=int(f % 240 < 120), d=int(f % 240 >= 120))
        eng.set_camera(cam)
        t1 = time.perf_counter()
        eng.render(SEED + f)
        eng.sync()
        t2 = time.perf_counter()
        eng.map_frame(wait=True)
        t3 = time.perf_counter()
        render_ms.append((t2 - t1) * 1e3)
        total_ms.append((t3 - t1) * 1e3)
    # the same frames once more, untimed, for what each of them was: rays, and the intersect kernels' own time on the GPU
    eng.time_kernels(True)
    cam = cam0
    for f in range(steps):
        cam = rt.camera_input(cam, 3.0, 1.0, w=int(f % 240 < 120), d=int(f % 240 >= 120))
        eng.set_camera(cam)
        eng.render(SEED + f)
        rays.append(eng.stats()["rays"])
        kern_ms.append(eng.kernel_times()[0])
    eng.time_kernels(False)
    eng.close()
    pct = lambda v, q: float(np.percentile(v, q))
    r_ms, r_rays, r_kern = np.array(render_ms), np.array(rays, np.float64), np.array(kern_ms)
    slow = [int(i) for i in np.argsort(-r_ms)[:5]]
    return {"workload": desc, "frames": steps, "includes": "render + device-to-host copy of the frame into page-locked memory (8.3 MB)",
            "frame_latency_ms": {"p50": pct(total_ms, 50), "p99": pct(total_ms, 99), "max": float(max(total_ms))},
            "render_only_ms": {"p50": pct(render_ms, 50), "p99": pct(render_ms, 99)},
            # does the tail follow the view (rays per frame, the intersect kernels' own time on the GPU) or the host?
            "rays_per_frame": {"min": float(r_rays.min()), "p50": pct(rays, 50), "max": float(r_rays.max())},
            "intersect_kernel_ms": {"p50": pct(kern_ms, 50), "p99": pct(kern_ms, 99), "max": float(r_kern.max())},
            "corr_render_ms_vs_rays": float(np.corrcoef(r_ms, r_rays)[0, 1]),
            "corr_render_ms_vs_intersect_kernel_ms": float(np.corrcoef(r_ms, r_kern)[0, 1]),
            "slowest_frames": [{"frame": i, "render_ms": float(r_ms[i]), "intersect_kernel_ms": float(r_kern[i]), "rays": float(r_rays[i])} for i in slow],
            "render_ms_by_frame_every_10th": [round(float(v), 3) for v in r_ms[::10]]}


def roofline_of(r, workload, cpu):
    flop_per_ray = cpu["intersect_flop_per_ray"] if cpu else FALLBACK_FLOP_PER_RAY[workload]
    scene_byte_per_ray = cpu["scene_byte_per_ray"] if cpu else FALLBACK_SCENE_BYTE_PER_RAY[workload]
    box_per_ray = cpu["box_tests_per_ray"] if cpu else FALLBACK_BOX_PER_RAY[workload]
    # dominant kernel = intersect_kernel (one launch per bounce level and batch); the launches of the last timed frame on
    # the slowest rank, measured with HIP events on the render stream; every step renders the same frame
    kern_s = r["kern_ms"] / 1e3
    st, n = r["stats"], r["kern_launches"]
    ach_tflops = flop_per_ray * st["rays"] / kern_s / 1e12 if kern_s > 0 else None
    stream_bytes = HIT_WRITE_BYTES * st["rays"] + RAY_READ_BYTES * (st["rays"] - st["paths"])
    ach_gbs = stream_bytes / kern_s / 1e9 if kern_s > 0 else None
    roof = {"bound": "valu", "achieved": ach_tflops, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
            "frac": (ach_tflops / PEAK_FP32_TFLOPS) if ach_tflops else None, "traffic": None,
            # the reference's unfused, unpacked arithmetic (kept for bit-exact ids) can issue 1/4 of the peak
            "frac_of_unfused_peak": (ach_tflops / (PEAK_FP32_TFLOPS / 4)) if ach_tflops else None,
            "kernel": "intersect_kernel", "launches": n, "avg_launch_ms": (r["kern_ms"] / n) if n else None,
            "flop_per_ray": flop_per_ray, "rays_per_launch": st["rays"] / n if n else None,
            # slab tests: what the reference algorithm performs for these rays (the figure `achieved` is computed from) and
            # what the engine executed (fewer where it skips boxes behind the closest hit)
            "box_tests_per_ray": {"algorithmic": box_per_ray, "executed": st["box_tests"] / st["rays"] if st["rays"] else None}}
    if cpu and kern_s > 0 and st["rays"] and "intersect_flop_per_ray_without_box_and_repeats" in cpu:
        # the same kernel priced by what it executed: its own slab-test count x 16 plus the other tests of the reference
        # algorithm counted once per leaf (no credit for boxes it culls or for the reference's second walk of a leaf that
        # is both children of its node)
        fx = 16.0 * st["box_tests"] / st["rays"] + cpu["intersect_flop_per_ray_without_box_and_repeats"]
        roof["flop_per_ray_executed"] = fx
        roof["frac_executed"] = fx * st["rays"] / kern_s / 1e12 / PEAK_FP32_TFLOPS
    hbm = {"bound": "hbm", "achieved": ach_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": (ach_gbs / PEAK_HBM_GBS) if ach_gbs else None,
           "traffic": None, "algorithmic_bytes_per_launch": stream_bytes / n if n else None,
           "stream_byte_per_ray": stream_bytes / st["rays"] if st["rays"] else None,
           "scene_byte_per_ray": scene_byte_per_ray}
    return roof, hbm


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="only the headline workload (profiling runs)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--batch-paths", type=int, default=0)
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="engines per rank taking the frames in turn; 0 = 3 from two GPUs on, else 1 (what a rank's share of "
                         "the frame measured best with on one GPU: tools/frames_in_flight.py)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 ranks all on GPU 0 with a gloo group (staged through the host): checks the multi-process "
                         "path on a one-GPU box; the number it prints is not a measurement")
    args = ap.parse_args()

    import raytracer_amd as rt
    job = Job(args)
    rank, world = job.rank, job.world
    threads = args.cpu_threads or min(16, len(os.sched_getaffinity(0)))
    with_cpu = (not args.no_cpu_baseline) and world == 1

    in_flight = args.frames_in_flight or (3 if world >= 2 and args.workload == "c2" else 1)
    r = run_workload(job, rt, args.workload, args.steps, args.warmup, args.batch_paths, in_flight)
    desc, W, H, S, B = r["desc"], r["W"], r["H"], r["S"], r["B"]
    out = None
    if rank == 0:
        cpu = cpu_baseline(args.workload, W, H, S, B, threads) if with_cpu else None
        roof, hbm = roofline_of(r, args.workload, cpu)
        # measured HBM traffic and lane use of the same kernel: PMC passes cannot run inside this process, so the figures
        # come from the committed summary of tools/profile.sh runs of this command (C2, one GPU) — and only if that
        # summary was taken from the very kernel sources this library was compiled from
        note = "no PMC summary for this workload"
        if args.workload == "c2" and world == 1 and r["kern_launches"]:
            src, per_frame, lanes = pmc_summary(PMC_SUMMARY, "intersect_kernel")
            have = rt.engine.source_hash()
            if src is None:
                note = f"{os.path.relpath(PMC_SUMMARY, ROOT)} not found"
            elif src != have:
                note = f"{os.path.relpath(PMC_SUMMARY, ROOT)} was taken from kernel sources {src}, this library is {have}: not reported"
            else:
                roof["traffic"] = hbm["traffic"] = per_frame / r["kern_launches"] if per_frame else None
                roof["active_lane_fraction"] = lanes
                note = f"bytes per launch, PMC, {os.path.relpath(PMC_SUMMARY, ROOT)} (kernel sources {src})"
        roof["traffic_note"] = hbm["traffic_note"] = note
        value = r["rays_frame"] * args.steps / r["dt"] / 1e6
        fm = r["frame_ms"]
        out = {
            "metric": "Mrays/s", "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": r["dt"] / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic" if not args.rehearse_on_one_gpu else "synthetic (rehearsal: all ranks on one GPU, not a measurement)",
            "config": {"workload": desc, "width": W, "height": H, "spp": S * S, "bounces": B, "rays_per_frame": r["rays_frame"],
                       "paths_per_frame": r["paths_frame"], "parallelism": f"tiles{world}", "frames_in_flight": in_flight, "batch_paths": r["eng"].cfg.max_batch_paths,
                       "build_key": f"0x{BUILD_KEY:08X}", "scene_build_s": r["scene_build_s"], "kernel_sources": rt.engine.source_hash()},
            "ms_per_frame": r["dt"] / args.steps * 1e3,
            # with several frames in flight ms_per_step is the time between finished frames; this is one frame alone
            "ms_per_frame_one_in_flight": r["one_in_flight_ms"],
            "frame_latency_ms": ({"p50": float(np.percentile(fm, 50)), "p99": float(np.percentile(fm, 99)), "max": float(max(fm))} if fm else None),
            "roofline": roof, "roofline_hbm": hbm, "cpu_baseline": cpu,
            # what each rank did in the last frame: with tiles interleaved the rays should be even; the slowest rank's
            # kernel time against the step time shows what the gather and the launch gaps cost
            "per_rank": {"rays": r["per_rank"][0], "intersect_kernel_ms": r["per_rank"][1],
                         "rays_min_max": [min(r["per_rank"][0]), max(r["per_rank"][0])],
                         "intersect_kernel_ms_min_max": [min(r["per_rank"][1]), max(r["per_rank"][1])]},
        }
    if args.workload == "c2" and not args.no_also:
        if world == 1:
            spread = key_spread(rt, r["eng"], r["tr"], W, H, S, B, threads if with_cpu else 0)
            if out is not None:
                out["build_key_spread"] = spread
                fr = {row["build_key"]: row.get("roofline_frac") for row in spread}
                if all(v is not None for v in fr.values()):
                    # the headline key was chosen as the median of these; say so with the numbers of this very run
                    roof = out["roofline"]
                    roof["frac_by_build_key"] = fr
                    roof["frac_min"], roof["frac_max"] = min(fr.values()), max(fr.values())
                    roof["frac_median_key"] = sorted(fr, key=fr.get)[len(fr) // 2]
                    roof["frac_executed_by_build_key"] = {row["build_key"]: row.get("roofline_frac_executed") for row in spread}
                out["value_by_build_key"] = {row["build_key"]: row["mrays_per_s"] for row in spread}
        for e_ in r["engs"]:
            e_.close()
        also = {}
        r3 = run_workload(job, rt, "c3", 2, 1)
        if rank == 0:
            cpu3 = cpu_baseline("c3", r3["W"], r3["H"], r3["S"], r3["B"], threads) if with_cpu else None
            roof3, _ = roofline_of(r3, "c3", cpu3)
            also["c3"] = {"workload": r3["desc"] + (f", tiles over {world} GPUs (C4)" if world > 1 else ""), "steps": 2, "warmup": 1,
                          "ms_per_frame": r3["dt"] / 2 * 1e3, "mrays_per_s": r3["rays_frame"] * 2 / r3["dt"] / 1e6,
                          "rays_per_frame": r3["rays_frame"], "scene_build_s": r3["scene_build_s"], "roofline": roof3, "cpu_baseline": cpu3,
                          "per_rank": {"rays_min_max": [min(r3["per_rank"][0]), max(r3["per_rank"][0])],
                                       "intersect_kernel_ms_min_max": [min(r3["per_rank"][1]), max(r3["per_rank"][1])]}}
        for e_ in r3["engs"]:
            e_.close()
        if world == 1:
            # BASELINE config 1 (the reference's Constants.h settings): the GPU's frame, and the reference's own
            # FrameManager frame on this box's host cores beside it
            r1 = run_workload(job, rt, "c1", 20, 3)
            also["c1"] = {"workload": r1["desc"], "steps": 20, "warmup": 3, "ms_per_frame": r1["dt"] / 20 * 1e3,
                          "mrays_per_s": r1["rays_frame"] * 20 / r1["dt"] / 1e6, "rays_per_frame": r1["rays_frame"],
                          "reference_framemanager": reference_c1_frame(r1["rays_frame"]) if with_cpu else None}
            for e_ in r1["engs"]:
                e_.close()
            also["c5"] = c5_on_one_gpu(job, rt, 240)
        else:
            r5 = run_workload(job, rt, "c5", 240, 3)
            if rank == 0:
                fm = r5["frame_ms"]
                also["c5"] = {"workload": r5["desc"], "frames": 240, "includes": "render + tile gather + device-to-host copy on the display rank",
                              "frame_latency_ms": {"p50": float(np.percentile(fm, 50)), "p99": float(np.percentile(fm, 99)), "max": float(max(fm))}}
            for e_ in r5["engs"]:
                e_.close()
        if out is not None:
            out["also"] = also
    if out is not None:
        print(json.dumps(out), flush=True)
    job.close()


if __name__ == "__main__":
    main()
