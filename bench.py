#!/usr/bin/env python3
"""Headline benchmark: whole-job Mrays/s of the per-pixel path on N MI355X (BASELINE.json `metric`).

A step = one frame of the workload: generate -> (intersect -> shade -> compact) x (bounces+1) -> resolve, with the frame
cut into image tiles across the N ranks and the tiles gathered to rank 0 over RCCL.  A ray = one closest-hit query
(one World::GetRayColor invocation of the reference).  Scene, camera and queues are resident in HBM before the timed
region; nothing is read from the host inside it.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c1|c3] [--no-cpu-baseline]
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

BUILD_KEY = 0x5EED0001
SEED = 7
CAM_DEFAULT = ((0.0, 3.0, -10.0), (0.0, 0.0, 1.0))          # World.cpp:48
WORKLOADS = {
    # name: (description, W, H, S, B)
    "c1": ("default scene 1280x720, 9 spp, 3 bounces (the reference's Constants.h)", 1280, 720, 3, 3),
    "c2": ("default scene 1920x1080, 64 spp, 8 bounces", 1920, 1080, 8, 8),
    "c3": ("synthetic 100k sphere/box scene 3840x2160, 256 spp, 8 bounces", 3840, 2160, 16, 8),
    # interactive sequence: the camera moves every frame as the front end's input handler would move it (App.cpp:98,
    # 114-137: mouse (3, 1) per frame, key W for frames 0-119, D for 120-239); every frame is waited for, so a step's
    # time is the latency a viewer sees, gather included
    "c5": ("default scene 1920x1080, 9 spp, 3 bounces, 240-frame camera path, one frame in flight", 1920, 1080, 3, 3),
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
FALLBACK_FLOP_PER_RAY = {"c1": 1002.0, "c2": 1002.0, "c3": 7000.0, "c5": 1002.0}
FALLBACK_SCENE_BYTE_PER_RAY = {"c1": 1480.0, "c2": 1480.0, "c3": 13400.0, "c5": 1480.0}


def build_scene(rt, workload):
    if workload in ("c1", "c2", "c5"):
        return rt.default_scene(BUILD_KEY), rt.camera_look(*CAM_DEFAULT)
    from raytracer_amd.synthetic import synthetic_scene
    return synthetic_scene(rt, 100000, BUILD_KEY + 2), rt.camera_look((0.0, 20.0, -150.0), (0.0, 0.0, 1.0))


def cpu_baseline(workload, W, H, S, B, threads):
    """the CPU restatement of the reference (oracle/, libm math like the reference) on a bounded sample of the workload:
    horizontal strips spread over the frame, every sample and bounce of their pixels"""
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
        s.build(BUILD_KEY)
        cam = O.camera(*CAM_DEFAULT)
    else:
        from raytracer_amd.synthetic import synthetic_tables, build_from_tables
        obj, tex_t, mat_t = synthetic_tables(100000)
        build_from_tables(s, tex_t, mat_t, obj, [[0.0, len(mat_t) - 1]], sky, BUILD_KEY + 2)
        cam = O.camera((0.0, 20.0, -150.0), (0.0, 0.0, 1.0))
    # sized for roughly 10-30 s on 16 host threads: the whole frame for c1/c2, a few row segments for c3
    n_strips, rows = {"c1": (1, H), "c2": (1, H), "c3": (4, 1), "c5": (1, H)}[workload]
    s.counters(reset=True)
    t0 = time.perf_counter()
    px = 0
    for k in range(n_strips):
        y0 = int((k + 0.5) * H / n_strips) if rows < H else 0
        x1 = W if workload != "c3" else W // 16
        s.render(cam, W, H, S, B, SEED, rect=(0, y0, x1, min(H, y0 + rows)), threads=threads, want=("rgba",))
        px += x1 * (min(H, y0 + rows) - y0)
    dt = time.perf_counter() - t0
    c = s.counters()
    # the reference's own thread count (THREAD_COUNT 7, Constants.h:6) on 16 strips of 8 rows (c3: one segment)
    t0 = time.perf_counter()
    for k in range(16 if workload != "c3" else 1):
        y0 = int((k + 0.5) * H / 16)
        s.render(cam, W, H, S, B, SEED, rect=(0, y0, W if workload != "c3" else W // 16, min(H, y0 + (8 if workload != "c3" else 1))),
                 threads=7, want=("rgba",))
    dt7 = time.perf_counter() - t0
    rays7 = s.counters()["rays"] - c["rays"]
    flops = sum(FLOP_INTERSECT[k] * c[k] for k in FLOP_INTERSECT)
    flops_shade = sum(FLOP_SHADE[k] * c[k] for k in FLOP_SHADE)
    nbytes = sum(SCENE_BYTE_WEIGHTS[k] * c[k] for k in SCENE_BYTE_WEIGHTS)
    return {"value": c["rays"] / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "value_7_threads": rays7 / dt7 / 1e6,
            "sample": f"{n_strips} strips x {rows} rows = {px} pixels of the {W}x{H} frame, all {S * S} spp and {B} bounces, "
                      f"{c['rays']} rays in {dt:.1f} s; contiguous pixel ranges per thread as FrameManager.cpp:80-82",
            "intersect_flop_per_ray": flops / c["rays"], "shade_flop_per_ray": flops_shade / c["rays"],
            "scene_byte_per_ray": nbytes / c["rays"], "rays_per_sample": c["rays"] / c["samples"]}


def pmc_traffic(path, kernel):
    """HBM bytes per frame of one kernel from a summary written by tools/pmc_summary.py (separate rocprofv3 --pmc runs of
    this same command): FETCH_SIZE and WRITE_SIZE are in units of 1 KiB, and on gfx950 FETCH_SIZE counts half
    of the bytes read, so  bytes = 1024 * (2 * FETCH_SIZE + WRITE_SIZE), summed over the kernel's template variants."""
    try:
        lines = open(path).read().splitlines()
    except OSError:
        return None
    total, inside, seen = 0.0, False, False
    for ln in lines:
        if not ln.startswith(" "):
            inside = ln.startswith(kernel)
            continue
        if inside:
            name, val = ln.split()[:2]
            if name == "FETCH_SIZE":
                total += 2048.0 * float(val); seen = True
            elif name == "WRITE_SIZE":
                total += 1024.0 * float(val); seen = True
    return total if seen else None


def pmc_active_lanes(path, kernel):
    """Active-lane fraction SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU) per template variant of a kernel, from the
    same summary."""
    try:
        lines = open(path).read().splitlines()
    except OSError:
        return None
    out, name = {}, None
    for ln in lines:
        if not ln.startswith(" "):
            name = ln.split("  dispatches")[0] if ln.startswith(kernel) else None
        elif name and ln.split()[:2] == ["active-lane", "fraction"]:
            out[name] = float(ln.split()[2])
    return out or None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--batch-paths", type=int, default=0)
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 ranks all on GPU 0 with a gloo group (staged through the host): checks the multi-process "
                         "path on a one-GPU box; the number it prints is not a measurement")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import raytracer_amd as rt
    from raytracer_amd.distributed import TileRenderer

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 with torch.distributed.run --nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP engine is the only compute path")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    red_device = torch.device("cpu") if args.rehearse_on_one_gpu else device     # where the timing reductions live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    desc, W, H, S, B = WORKLOADS[args.workload]
    scene, cam = build_scene(rt, args.workload)
    eng = rt.Engine(rt.default_config(width=W, height=H, samples_per_axis=S, bounces=B, device=local_rank,
                                      max_batch_paths=args.batch_paths))
    eng.set_scene(scene)
    eng.set_camera(cam)
    tr = TileRenderer(eng, device, rank, world)

    def barrier():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)

    frame_ms = []
    for _ in range(args.warmup):
        tr.render_frame(SEED)
    barrier()
    eng.time_kernels(True)
    t0 = time.perf_counter()
    if args.workload == "c5":
        for f in range(args.steps):
            cam = rt.camera_input(cam, 3.0, 1.0, w=int(f % 240 < 120), d=int(f % 240 >= 120))
            eng.set_camera(cam)
            t1 = time.perf_counter()
            tr.render_frame(SEED + f)
            barrier()
            frame_ms.append((time.perf_counter() - t1) * 1e3)
    else:
        for _ in range(args.steps):
            tr.render_frame(SEED)
    barrier()
    dt = time.perf_counter() - t0
    kern_ms, kern_launches = eng.kernel_times()
    eng.time_kernels(False)
    st = eng.stats()                       # this rank's share of one frame (every step renders the same frame)

    t = torch.tensor([dt], dtype=torch.float64, device=red_device)
    counts = torch.tensor([st["rays"], st["paths"]], dtype=torch.float64, device=red_device)
    kt = torch.tensor([kern_ms], dtype=torch.float64, device=red_device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        dist.all_reduce(kt, op=dist.ReduceOp.MAX)
    dt = float(t[0])
    rays_frame, paths_frame = float(counts[0]), float(counts[1])
    if rank == 0:
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            threads = args.cpu_threads or min(16, len(os.sched_getaffinity(0)))
            cpu = cpu_baseline(args.workload, W, H, S, B, threads)
        flop_per_ray = cpu["intersect_flop_per_ray"] if cpu else FALLBACK_FLOP_PER_RAY[args.workload]
        scene_byte_per_ray = cpu["scene_byte_per_ray"] if cpu else FALLBACK_SCENE_BYTE_PER_RAY[args.workload]
        value = rays_frame * args.steps / dt / 1e6
        # dominant kernel = intersect_kernel (one launch per bounce level); the launches of the last timed frame on the
        # slowest rank, measured with HIP events on the render stream; every step renders the same frame
        kern_s = float(kt[0]) / 1e3
        rays_rank0 = st["rays"]
        ach_tflops = flop_per_ray * rays_rank0 / kern_s / 1e12 if kern_s > 0 else None
        stream_bytes = HIT_WRITE_BYTES * st["rays"] + RAY_READ_BYTES * (st["rays"] - st["paths"])
        ach_gbs = stream_bytes / kern_s / 1e9 if kern_s > 0 else None
        # measured HBM traffic of the same kernel: PMC passes cannot run inside this process, so the figure comes from the
        # committed summary of tools/profile.sh runs of this command (C2, one GPU) and is null for anything else
        traffic, lanes = None, None
        if args.workload == "c2" and world == 1 and kern_launches:
            per_frame = pmc_traffic(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r1_c2_pmc_summary.txt"),
                                    "intersect_kernel")
            traffic = per_frame / kern_launches if per_frame else None
            lanes = pmc_active_lanes(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r1_c2_pmc_summary.txt"),
                                     "intersect_kernel")
        out = {
            "metric": "Mrays/s", "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic" if not args.rehearse_on_one_gpu else "synthetic (rehearsal: all ranks on one GPU, not a measurement)",
            "config": {"workload": desc, "width": W, "height": H, "spp": S * S, "bounces": B, "rays_per_frame": rays_frame,
                       "paths_per_frame": paths_frame, "parallelism": f"tiles{world}", "batch_paths": eng.cfg.max_batch_paths},
            "ms_per_frame": dt / args.steps * 1e3,
            "frame_latency_ms": ({"p50": float(np.percentile(frame_ms, 50)), "p99": float(np.percentile(frame_ms, 99)),
                                  "max": float(max(frame_ms))} if frame_ms else None),
            "roofline": {"bound": "valu", "achieved": ach_tflops, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": (ach_tflops / PEAK_FP32_TFLOPS) if ach_tflops else None, "traffic": traffic,
                         # the reference's unfused, unpacked arithmetic (kept for bit-exact ids) can issue 1/4 of the peak
                         "frac_of_unfused_peak": (ach_tflops / (PEAK_FP32_TFLOPS / 4)) if ach_tflops else None,
                         "kernel": "intersect_kernel", "launches": kern_launches,
                         "avg_launch_ms": (float(kt[0]) / kern_launches) if kern_launches else None,
                         "active_lane_fraction": lanes, "flop_per_ray": flop_per_ray, "rays_per_launch": rays_rank0 / kern_launches if kern_launches else None},
            # the same kernel against HBM: ray records in, hit records out (the scene itself is read from LDS)
            "roofline_hbm": {"bound": "hbm", "achieved": ach_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                             "frac": (ach_gbs / PEAK_HBM_GBS) if ach_gbs else None, "traffic": traffic,
                             "traffic_unit": "bytes per launch (PMC, profiles/r1_c2_pmc_summary.txt)",
                             "algorithmic_bytes_per_launch": stream_bytes / kern_launches if kern_launches else None,
                             "stream_byte_per_ray": stream_bytes / rays_rank0 if rays_rank0 else None,
                             "scene_byte_per_ray_from_lds": scene_byte_per_ray},
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
