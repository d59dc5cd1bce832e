#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on MI355X: Mrays/s (whole job) + frames/s, Cornell Box 1080p 4 spp / 8 bounces.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one frame of the hot path on synthetic inputs already resident in HBM:
GBufferGeneration::Render + Raytracing::Render (Source/App.cpp:1196-1328) and, for N > 1, the RCCL gather
of the per-rank row bands to rank 0 + de-interleave. The frame is sharded by 16-row bands (band b -> rank b % N),
total work is fixed as N grows => "scaling": "strong". Rays = primary (G-buffer) + secondary rays actually
traced, counted on the device. One JSON line is printed by rank 0.

Extra objects in the line (DESIGN.md "Measurement"):
  roofline      dominant kernel of the frame: algorithmic bytes / HIP-event time of its launches vs 8 TB/s HBM
  cpu_baseline  the CPU oracle (oracle/, OpenMP) timed on this host on a bounded row slab of the same frame
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec
BAND = 16

WORKLOADS = {
    # name: (scene factory kwargs, width, height, spp, bounces, description)
    "c2": ("cornell", 1920, 1080, 4, 8, "Cornell Box 1920x1080 4spp 8 bounces full GGX metallic-roughness (BASELINE configs[1])"),
    "c3": ("sponza", 1920, 1080, 1, 8, "Sponza-scale ~250k tris 1920x1080 1spp 8 bounces + RR (BASELINE configs[2])"),
    "c4": ("cornell", 3840, 2160, 16, 16, "Cornell Box 3840x2160 16spp 16 bounces (BASELINE configs[3])"),
    "c5": ("grid", 1920, 1080, 4, 8, "10k instances two-level BVH 1920x1080 4spp 8 bounces (BASELINE configs[4])"),
    "c1": ("cornell_lambert", 256, 256, 1, 2, "Cornell Box 256x256 1spp 2 bounces Lambertian only (BASELINE configs[0])"),
}


def make_scene(kind, aspect, S):
    if kind == "cornell":
        return S.cornell_box(aspect=aspect, variant="ggx"), 0
    if kind == "cornell_lambert":
        return S.cornell_box(aspect=aspect, variant="diffuse"), 1
    if kind == "sponza":
        return S.sponza_scale(aspect=aspect), 0
    if kind == "grid":
        return S.instanced_grid(n=100, aspect=aspect), 0
    raise ValueError(kind)


def cpu_baseline(scene, gs, W, H, L, budget_s=15.0):
    """Time the oracle (kind "port": the C restatement, OpenMP over scanlines, its own BVH) on a row slab."""
    oracle = ge.load_oracle()
    osc = oracle.OracleScene(scene, accel_mode=1)
    gb = {k: np.zeros((H, W, c), dt) for k, (dt, c) in L.GBUFFER_FORMATS.items()}
    consts = np.zeros((), L.GBUFFER_CONSTANTS)
    consts["RenderSize"] = (W, H); consts["Flags"] = L.GBufferFlags.DefaultNoDenoiser
    cores = os.cpu_count() or 1

    def slab(rows):
        y0 = max(0, H // 2 - rows // 2); y1 = min(H, y0 + rows)
        t = time.perf_counter()
        rays = osc.gbuffer(consts, gb, rows=(y0, y1))
        rays += osc.raytrace(gs, gb, rows=(y0, y1))
        return rays, time.perf_counter() - t, (y0, y1)

    rays, dt, _ = slab(8)                                   # calibration (also warms the threads up)
    rows = int(min(H, max(8, 8 * budget_s / max(dt, 1e-4))))
    rays, dt, (y0, y1) = slab(rows)
    reps = 1
    while dt < 0.66 * budget_s and reps < 64:               # fast hosts: repeat the slab until ~budget_s of CPU work
        r2, d2, _ = slab(rows)
        rays += r2; dt += d2; reps += 1
    osc.close()
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"rows {y0}..{y1} of the {W}x{H} frame x{reps} ({rays} rays in {dt:.2f} s), oracle/pt_oracle.c with its own BVH, OpenMP"}


def self_launch(n):
    """Run this script under torch.distributed.run with one rank per GPU; stdout of the children is scanned for the
    JSON result line (rank 0 prints exactly one), everything else they print goes to our stderr."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")            # dmabuf IPC only on this pool (RCCL needs it across processes)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in p.stdout:
        if out.lstrip().startswith("{") and out.rstrip().endswith("}"):
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = p.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        rc = 1
        sys.stderr.write("bench.py: the ranks exited without a result line\n")
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    ap.add_argument("--inflight", type=int, default=3,
                    help="frames in flight per GPU: independent contexts on separate HIP streams, so the launch-bound tail "
                         "rounds of one frame overlap the wide rounds of the next")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="developer aid: skip the per-launch HIP events (and with them the roofline object), so the timed region replays hipGraphs")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="developer aid: render only rank 0's share of an N-rank sharding on one GPU (no gather), to see the per-rank frame time")
    ap.add_argument("--unfused", action="store_true",
                    help="developer aid: a round as two launches (k_shade + k_extend2) instead of the fused k_round, to profile the halves separately")
    ap.add_argument("--rehearse-collective", action="store_true",
                    help="developer aid for a 1-GPU box: run the N > 1 code path (RCCL process group, gather to rank 0, de-interleave) with world size 1")
    ap.add_argument("--launch-check", action="store_true",
                    help="developer aid / CPU test: stop after the process group is up (gloo, no GPU touched) and print the world size")
    args = ap.parse_args()

    # `python bench.py --gpus N` with N > 1 and no torchrun environment: start the N ranks ourselves, as a CHILD process
    # (never exec: nothing here has touched the GPU yet, and nothing may before this point), relay rank 0's one JSON line
    # and leave with the child's return code.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    if args.launch_check:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        t = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"launch_check": True, "world_size": dist.get_world_size(), "ranks_seen": int(t[0])}), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    collective = world > 1 or args.rehearse_collective
    # RCCL writes a version banner to stdout when the first communicator comes up; the contract is ONE JSON line on stdout,
    # so everything until the result is printed goes to stderr at the file-descriptor level (native prints included)
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    if collective:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=device)

    ge.load_package()
    import dxpbrt_amd.layouts as L
    import dxpbrt_amd.ptamd as P
    import dxpbrt_amd.scenes as S
    import dxpbrt_amd.sharding as SH

    kind, W, H, spp, bounces, desc = WORKLOADS[args.workload]
    scene, ext = make_scene(kind, W / H, S)
    max_rows = max(P.local_rows(H, r, world, BAND) for r in range(world))
    offsets = (np.arange(world, dtype=np.uint64) * np.uint64(max_rows * W * 8))

    class Lane:
        """One frame in flight: its own HIP stream, library context (queues, BVH copy) and G-buffer textures."""
        def __init__(self):
            self.stream = torch.cuda.Stream(device)
            with torch.cuda.stream(self.stream):
                self.ctx = P.DeviceContext(local_rank, stream=self.stream.cuda_stream)
                self.ctx.set_sharding(rank, args.emulate_world if args.emulate_world else world, BAND)
                self.scene = P.Scene(self.ctx, scene, device)
                self.renderer = P.Renderer(self.ctx, self.scene, W, H)
                if collective:                                  # equal-sized gather pieces
                    self.renderer.textures["Radiance"] = torch.zeros((max_rows, W, 4), dtype=torch.int16, device=device)
                    for op in (self.renderer.gbuffer, self.renderer.raytracing):
                        op.Textures = self.renderer.textures
                self.full = torch.zeros((H, W, 4), dtype=torch.int16, device=device) if rank == 0 else None
                self.gathered = torch.zeros((world, max_rows, W, 4), dtype=torch.int16, device=device) if (rank == 0 and collective) else None
            self.stream.synchronize()

    lanes = [Lane() for _ in range(max(1, args.inflight))]
    ctx = lanes[0].ctx
    BASE_FLAGS = 0x10 if args.unfused else 0                   # PT_DEBUG_UNFUSED_ROUNDS
    for lane in lanes:
        lane.ctx.set_debug_flags(BASE_FLAGS)

    def step(frame_index):
        lane = lanes[frame_index % len(lanes)]
        gs = S.graphics_settings(W, H, spp=spp, bounces=bounces, frame_index=frame_index, ext_flags=ext)
        with torch.cuda.stream(lane.stream):
            lane.renderer.render(gs)
            if collective:
                SH.gather_to_root(lane.renderer.textures["Radiance"], rank, world, dist, out=lane.gathered)
                if rank == 0:
                    lane.ctx.check(lane.ctx.lib.pt_deinterleave_bands(lane.ctx.handle, lane.full.data_ptr(), lane.gathered.data_ptr(),
                                                                      offsets.ctypes.data, world, BAND, W, H, 8))
        return gs

    def barrier():
        torch.cuda.synchronize(device)
        if collective:
            dist.barrier()
        torch.cuda.synchronize(device)

    for i in range(args.warmup):
        step(i)
    barrier()
    timing = world == 1 and not collective and not args.emulate_world and not args.no_kernel_timing          # HIP events around every extend / shade launch (library side)
    for lane in lanes:
        lane.ctx.reset_counters()
        if timing:
            lane.ctx.enable_kernel_timing(True)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        gs = step(args.warmup + i)
    enqueue_s = time.perf_counter() - t0
    barrier()
    elapsed = time.perf_counter() - t0
    counters = [lane.ctx.counters() for lane in lanes]
    kts = [lane.ctx.kernel_timing() for lane in lanes] if timing else None
    for lane in lanes:
        lane.ctx.enable_kernel_timing(False)
    primary = sum(c.PrimaryRays for c in counters); secondary = sum(c.SecondaryRays for c in counters)

    rays_local = float(primary + secondary)
    t = torch.tensor([elapsed, rays_local, float(secondary)], dtype=torch.float64, device=device)
    if collective:
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed, rays_total, secondary_total = float(tmax[0]), float(tsum[1]), float(tsum[2])
    else:
        rays_total, secondary_total = rays_local, float(secondary)

    result = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        result = {
            "metric": "Mrays/s per GPU + frames/s at 1080p, Cornell Box 4spp/8bounce",
            "value": rays_total / elapsed / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "frames_per_s": args.steps / elapsed, "mrays_per_s_per_gpu": rays_total / elapsed / 1e6 / world,
            "rays_per_frame": rays_total / args.steps, "host_enqueue_ms_per_step": enqueue_s / args.steps * 1e3,
            "config": {"workload": desc, "width": W, "height": H, "spp": spp, "bounces": bounces, "frames_in_flight": len(lanes),
                       "russian_roulette": True, "triangles": scene.triangle_count, "instances": len(scene.objects),
                       "sharding": f"{BAND}-row bands, band b -> rank b % {world}" + (", RCCL gather to rank 0" if world > 1 else ""),
                       "rccl_world_size": dist.get_world_size() if collective else 1,
                       "parity": "bit-identical to oracle on this scene (tests/test_gpu_parity.py)"},
        }

    # ---- roofline of the dominant kernel (N = 1): one extra frame with traversal statistics for B_bvh
    if rank == 0 and timing:
        kt = {k: sum(x[k] for x in kts) for k in kts[0]}
        ctx.set_debug_flags(1 | BASE_FLAGS)
        ctx.reset_counters()
        with torch.cuda.stream(lanes[0].stream):
            lanes[0].renderer.render(gs)
        cs = ctx.counters()
        ctx.set_debug_flags(BASE_FLAGS)
        # one more instrumented pass on ONE lane: with several frames in flight an event pair around a launch also spans
        # the other lanes' kernels, so the per-launch durations above are upper bounds. Here each launch has the GPU alone.
        ctx.reset_counters(); ctx.enable_kernel_timing(True)
        with torch.cuda.stream(lanes[0].stream):
            for i in range(args.steps):
                lanes[0].renderer.render(S.graphics_settings(W, H, spp=spp, bounces=bounces, frame_index=args.warmup + i, ext_flags=ext))
        torch.cuda.synchronize(device)
        ks = ctx.kernel_timing(); serial_secondary = float(ctx.counters().SecondaryRays)
        ctx.enable_kernel_timing(False)
        rays_frame = max(1, cs.SecondaryRays)
        node_b, tri_b = 64, 48
        bvh_bytes_per_ray = (cs.NodesVisited * node_b + cs.TrianglesTested * tri_b) / float(cs.PrimaryRays + cs.SecondaryRays)
        sec_rays = secondary_total
        # algorithmic bytes per secondary ray (DESIGN.md / SURVEY 8d). Fused round (the product path): ray read 32 + path state
        # 48 r + 48 w + ray write 32 + hit geometry 108 + B_bvh = 268 + B_bvh (the 16 B hit record stays in registers).
        # Two-kernel form (--unfused): extend = ray read 32 + hit write 16 + B_bvh; shade = hit read 16 + ray dir 16 + state 96 + ray write 32 + geometry 108
        STATE = {"k_round": 268.0, "k_extend": 48.0, "k_shade": 252.0}
        WITH_BVH = {"k_round", "k_extend"}

        def kernel_table(t, rays):
            tab = {}
            for name, key in (("k_round", "round"), ("k_extend", "extend"), ("k_shade", "shade")):
                if t.get(key + "_launches", 0):
                    per_ray = STATE[name] + (bvh_bytes_per_ray if name in WITH_BVH else 0.0)
                    tab[name] = {"ms": t[key + "_ms"], "launches": t[key + "_launches"], "bytes": rays * per_ray, "state_bytes": rays * STATE[name]}
            return tab

        def gbps(nbytes, ms):
            return nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0

        kernels = kernel_table(kt, sec_rays)
        dom = max(kernels, key=lambda k: kernels[k]["ms"])
        kd = kernels[dom]
        achieved = gbps(kd["bytes"], kd["ms"])
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get(args.workload, {}).get(dom)
            except Exception:
                traffic = None
        result["roofline"] = {
            "bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "state_only": {"achieved": gbps(kd["state_bytes"], kd["ms"]), "frac": gbps(kd["state_bytes"], kd["ms"]) / HBM_PEAK_GBS,
                           "note": "queue/state bytes only, without B_bvh (BVH bytes of an LDS- or cache-resident scene never reach HBM)"},
            "avg_launch_ms": kd["ms"] / max(1, kd["launches"]), "launches_timed": kd["launches"],
            "algorithmic_bytes_per_launch": kd["bytes"] / max(1, kd["launches"]),
            "bvh_bytes_per_ray": bvh_bytes_per_ray, "nodes_per_ray": cs.NodesVisited / float(cs.PrimaryRays + cs.SecondaryRays),
            "tris_per_ray": cs.TrianglesTested / float(cs.PrimaryRays + cs.SecondaryRays),
            "other_kernel": {k: {"ms_total": v["ms"], "GBps": gbps(v["bytes"], v["ms"])} for k, v in kernels.items() if k != dom},
            "note": "HIP-event time summed over every launch of the timed steps (all frames in flight, so a launch shares the GPU with "
                    "the other lanes' kernels); bytes = algorithmic bytes per secondary ray x rays (DESIGN.md)",
        }
        # the bound that actually binds: VALU issue. Instruction count per frame from the committed PMC
        # profile, issue peak = CUs x 4 SIMDs x clock / 4 cycles per wave64 instruction
        try:
            vj = json.load(open(tpath)).get(args.workload + "_valu")
        except Exception:
            vj = None
        if vj:
            props = torch.cuda.get_device_properties(device)
            peak_issue = props.multi_processor_count * 4 * 2.4e9 / 4.0
            issued = vj["wave_instructions_per_frame"] * (args.steps / elapsed)
            result["roofline"]["valu_issue"] = {"wave_instructions_per_frame": vj["wave_instructions_per_frame"], "issued_per_s": issued,
                                                 "peak_per_s": peak_issue, "frac": issued / peak_issue,
                                                 "note": "SQ_INSTS_VALU from profiles/ x frames/s of this run; 2.4 GHz engine clock"}
        serial = {}
        for name, v in kernel_table(ks, serial_secondary).items():
            g = gbps(v["bytes"], v["ms"]); so = gbps(v["state_bytes"], v["ms"])
            serial[name] = {"avg_launch_ms": v["ms"] / max(1, v["launches"]), "launches_timed": v["launches"], "achieved": g, "frac": g / HBM_PEAK_GBS,
                            "state_only": {"achieved": so, "frac": so / HBM_PEAK_GBS}}
        result["roofline"]["one_frame_in_flight"] = dict(serial, note="same K steps repeated on a single stream after the timed region: "
                                                         "per-launch durations without other lanes' kernels inside the event pair "
                                                         "(these are the figures rocprofv3's per-kernel averages agree with)")
        del rays_frame
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.emulate_world:
        result["cpu_baseline"] = cpu_baseline(scene, gs, W, H, L, args.cpu_budget)

    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    os.close(saved_stdout)
    if rank == 0:
        print(json.dumps(result), flush=True)
    os.dup2(2, 1)                                                 # teardown chatter (process group, contexts) stays off stdout too
    for lane in lanes:
        lane.ctx.close()
    if collective:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
