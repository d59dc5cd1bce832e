#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on MI355X: Mrays/s (whole job) + frames/s, Cornell Box 1080p 4 spp / 8 bounces.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one frame of the hot path on synthetic inputs already resident in HBM:
GBufferGeneration::Render + Raytracing::Render (Source/App.cpp:1196-1328) and, for N > 1, the library's RCCL gather
(pt_gather_bands: grouped ncclSend / ncclRecv of the row bands straight into rank 0's full frame, on the frame's own
stream). The frame is sharded by 16-row bands (band b -> rank b % N),
total work is fixed as N grows => "scaling": "strong". Rays = primary (G-buffer) + secondary rays actually
traced, counted on the device. One JSON line is printed by rank 0.

Extra objects in the line (DESIGN.md "Measurement"):
  roofline      the dominant kernel: algorithmic HBM bytes per launch (nothing that LDS serves) / its average launch duration (per-launch
                HIP events on its stream, single stream) vs 8 TB/s, PMC traffic beside it; the whole frame alone and pipelined; BVH bytes by
                where they are served from; VALU issue against the chip's best and against the ceiling at the kernels' occupancy
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
    # BASELINE configs[2] says "Sponza-scale glTF": the same mesh with what a glTF import brings -- UVs, tangents, three 1024^2 textures per material, alpha-masked strips
    "c3t": ("sponza_textured", 1920, 1080, 1, 8, "Sponza-scale ~250k tris, 24 textured materials (base colour + normal + metallic-roughness, 1024^2 RGBA8 each), 11.9 % of the triangles alpha-masked, 1920x1080 1spp 8 bounces + RR"),
    "c4": ("cornell", 3840, 2160, 16, 16, "Cornell Box 3840x2160 16spp 16 bounces (BASELINE configs[3])"),
    "c5": ("grid", 1920, 1080, 4, 8, "10k instances two-level BVH 1920x1080 4spp 8 bounces (BASELINE configs[4])"),
    "c1": ("cornell_lambert", 256, 256, 1, 2, "Cornell Box 256x256 1spp 2 bounces Lambertian only (BASELINE configs[0])"),
    # not a BASELINE config: the per-frame cost of a dynamic scene (skinning, in-place BLAS refit, TLAS rebuild), Scene.ixx:233-280,327-380
    "dynamic": ("dynamic", 1920, 1080, 1, 4, "1024 static instances + one skinned 2048-triangle mesh, 1920x1080 1spp 4 bounces, structures updated every frame"),
}


def source_hash():
    """hash of the kernel sources a libptamd.so is built from: PMC figures in profiles/traffic.json are only quoted for the same sources"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "directx-physically-based-raytracer_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")) or f == "Makefile":
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def make_scene(kind, aspect, S):
    if kind == "cornell":
        return S.cornell_box(aspect=aspect, variant="ggx"), 0
    if kind == "cornell_lambert":
        return S.cornell_box(aspect=aspect, variant="diffuse"), 1
    if kind == "sponza":
        return S.sponza_scale(aspect=aspect), 0
    if kind == "sponza_textured":
        return S.sponza_scale(aspect=aspect, textured=True), 0
    if kind == "grid":
        return S.instanced_grid(n=100, aspect=aspect), 0
    if kind == "dynamic":
        return S.dynamic_instanced(n=32, aspect=aspect), 0
    raise ValueError(kind)


def host_cpu_share():
    """What this process may really use of the host: the affinity mask, the cgroup CPU quota, and what OpenMP will start."""
    info = {"os_cpu_count": os.cpu_count() or 1, "sched_affinity": len(os.sched_getaffinity(0)), "cgroup_cpu_max": None, "cgroup_cpus": None}
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                info["cgroup_cpu_max"] = " ".join(txt)
                if txt[0] != "max":
                    info["cgroup_cpus"] = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0]); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                info["cgroup_cpu_max"] = f"{int(q)} {int(per)}"
                if q > 0:
                    info["cgroup_cpus"] = q / per
            break
        except (OSError, ValueError, IndexError):
            continue
    return info


def cpu_baseline(scene, gs, W, H, L, budget_s=15.0):
    """Time the oracle (kind "port": the C restatement, OpenMP over scanlines, its own BVH) on a row slab: with as many threads as the
    process may use (affinity mask and cgroup quota, not os.cpu_count()), and with ONE thread on a smaller slab."""
    import ctypes
    oracle = ge.load_oracle()
    osc = oracle.OracleScene(scene, accel_mode=1)
    gb = {k: np.zeros((H, W, c), dt) for k, (dt, c) in L.GBUFFER_FORMATS.items()}
    consts = np.zeros((), L.GBUFFER_CONSTANTS)
    consts["RenderSize"] = (W, H); consts["Flags"] = L.GBufferFlags.DefaultNoDenoiser
    share = host_cpu_share()
    usable = share["sched_affinity"]
    if share["cgroup_cpus"]:
        usable = max(1, min(usable, int(share["cgroup_cpus"] + 0.5)))
    omp = None
    try:
        omp = ctypes.CDLL("libgomp.so.1")                    # the copy the oracle is linked against
        omp.omp_get_max_threads.restype = ctypes.c_int
        share["omp_max_threads_default"] = int(omp.omp_get_max_threads())
    except OSError:
        share["omp_max_threads_default"] = None

    def slab(rows, threads=usable):
        y0 = max(0, H // 2 - rows // 2); y1 = min(H, y0 + rows)
        t = time.perf_counter()
        rays = osc.gbuffer(consts, gb, rows=(y0, y1), threads=threads)
        rays += osc.raytrace(gs, gb, rows=(y0, y1), threads=threads)
        return rays, time.perf_counter() - t, (y0, y1)

    rays, dt, _ = slab(8)                                   # calibration (also warms the threads up)
    rows = int(min(H, max(8, 8 * budget_s / max(dt, 1e-4))))
    rays, dt, (y0, y1) = slab(rows)
    reps = 1
    while dt < 0.66 * budget_s and reps < 64:               # fast hosts: repeat the slab until ~budget_s of CPU work
        r2, d2, _ = slab(rows)
        rays += r2; dt += d2; reps += 1
    # the same oracle on ONE thread, ~3 s: what a core does, next to what the host does
    r1, d1, _ = slab(2, 1)
    rows1 = int(min(H, max(2, 2 * 3.0 / max(d1, 1e-4))))
    r1, d1, (a0, a1) = slab(rows1, 1)
    one = {"value": r1 / d1 / 1e6, "unit": "Mrays/s", "cores": 1, "sample": f"rows {a0}..{a1} ({r1} rays in {d1:.2f} s)"}
    osc.close()
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": int(usable), "kind": "port",
            "sample": f"rows {y0}..{y1} of the {W}x{H} frame x{reps} ({rays} rays in {dt:.2f} s), oracle/pt_oracle.c with its own BVH, OpenMP",
            "threads": int(usable), "one_thread": one, "host": share,
            "parallel_efficiency": (rays / dt / 1e6) / (one["value"] * usable) if one else None}


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
    ap.add_argument("--chains", type=int, default=0,
                    help="pt_set_round_chains: independent chains of launches per frame (0 = the library's choice: 3 for a scene beyond LDS when a frame has the GPU "
                         "to itself, else 1). --chains 1 for a kernel trace in which every launch ran alone")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="developer aid: skip the per-launch HIP events (and with them the roofline object), so the timed region replays hipGraphs")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="developer aid: render only rank 0's share of an N-rank sharding on one GPU (no gather), to see the per-rank frame time")
    ap.add_argument("--unfused", action="store_true",
                    help="developer aid: a round as two launches (k_shade + k_extend2) instead of the fused k_round, to profile the halves separately")
    ap.add_argument("--rehearse-collective", action="store_true",
                    help="developer aid for a 1-GPU box: run the N > 1 code path (process group, one RCCL communicator per lane, pt_gather_bands) with world size 1; "
                         "the frame's 68 bands travel by grouped ncclSend / ncclRecv to the rank itself (PT_DEBUG_GATHER_SELF_EXCHANGE)")
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

    kind, W, H, spp, bounces, desc = WORKLOADS[args.workload]
    scene, ext = make_scene(kind, W / H, S)
    dynamic = kind == "dynamic"
    if dynamic:
        args.inflight = 1                                       # every frame depends on the structures the previous one updated

    class Lane:
        """One frame in flight: its own HIP stream, library context (path queues, counters) and G-buffer textures. The scene -- vertex
        and index buffers, acceleration structures, traversal copy -- is built ONCE, by lane 0; the others view it (pt_share_scene)."""
        def __init__(self, owner=None):
            self.stream = torch.cuda.Stream(device)
            with torch.cuda.stream(self.stream):
                self.ctx = P.DeviceContext(local_rank, stream=self.stream.cuda_stream)
                self.ctx.set_sharding(rank, args.emulate_world if args.emulate_world else world, BAND)
                self.ctx.set_frames_in_flight(max(1, args.inflight))
                self.ctx.set_round_chains(args.chains)
                self.scene = P.Scene(self.ctx, scene, device) if owner is None else P.SharedScene(self.ctx, owner.scene)
                self.renderer = P.Renderer(self.ctx, self.scene, W, H)
                self.full = torch.zeros((H, W, 4), dtype=torch.int16, device=device) if (rank == 0 and collective) else None
            self.stream.synchronize()

    lanes = [Lane()]
    lanes += [Lane(lanes[0]) for _ in range(max(1, args.inflight) - 1)]
    ctx = lanes[0].ctx
    if collective:
        # one RCCL communicator per lane (= per stream: frames in flight never share one), made by the library itself; the 128-byte
        # id of each comes from rank 0 through the process group torch.distributed.run set up
        for lane in lanes:
            uid = [P.DeviceContext.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            lane.ctx.comm_init(uid[0], rank, world)
    BASE_FLAGS = 0x10 if args.unfused else 0                   # PT_DEBUG_UNFUSED_ROUNDS
    if collective and world == 1:
        BASE_FLAGS |= 0x80                                      # PT_DEBUG_GATHER_SELF_EXCHANGE: the rehearsal really issues ncclSend / ncclRecv (to itself), one pair per band
    for lane in lanes:
        lane.ctx.set_debug_flags(BASE_FLAGS)
    dyn_events = []
    gather_tail = [None]                                        # the event behind the last gather enqueued by this rank

    def step(frame_index):
        lane = lanes[frame_index % len(lanes)]
        gs = S.graphics_settings(W, H, spp=spp, bounces=bounces, frame_index=frame_index, ext_flags=ext)
        with torch.cuda.stream(lane.stream):
            if dynamic:                                         # Scene::Tick + SkinSkeletalMeshes + CreateAccelerationStructures, every frame
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                bar = scene.nodes[-1].meshes[0]
                ev[0].record(lane.stream)
                lane.scene.SkinSkeletalMeshes(bar, S.bar_pose(30.0 * np.sin(0.2 * frame_index), 0.05 * (frame_index % 7)))
                ev[1].record(lane.stream)
                lane.scene.UpdateAccelerationStructures(len(scene.nodes) - 1)
                ev[2].record(lane.stream)
                lane.renderer.render(gs)
                ev[3].record(lane.stream)
                dyn_events.append(ev)
            else:
                lane.renderer.render(gs)
            if collective:
                # one gather at a time per rank, in step order on every rank: the lanes have a communicator each (RCCL allows their concurrent use),
                # but two grouped exchanges in flight on different streams could start in different orders on different ranks and wait for each other
                # while the render kernels hold the CUs (VERDICT r3 weak point 10). An event chains them; the frames' render kernels still overlap.
                if gather_tail[0] is not None:
                    lane.stream.wait_event(gather_tail[0])
                lane.ctx.gather_bands(lane.renderer.textures["Radiance"], lane.full, W, H, 8, 0)
                gather_tail[0] = torch.cuda.Event()
                gather_tail[0].record(lane.stream)
        return gs

    def barrier():
        torch.cuda.synchronize(device)
        if collective:
            dist.barrier()
        torch.cuda.synchronize(device)

    for i in range(args.warmup):
        step(i)
    barrier()
    dyn_events.clear()
    for lane in lanes:
        lane.ctx.reset_counters()
    barrier()
    # ---- the timed region: K steps, frames in flight on their own streams, hipGraph replay (no per-launch events here: an event pair
    # around a launch on one stream also spans the other lanes' kernels; kernel durations are measured in the single-stream pass below)
    t0 = time.perf_counter()
    for i in range(args.steps):
        gs = step(args.warmup + i)
    enqueue_s = time.perf_counter() - t0
    barrier()
    elapsed = time.perf_counter() - t0
    counters = [lane.ctx.counters() for lane in lanes]
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
        acc = ctx.accel_stats()
        result = {
            "metric": "Mrays/s per GPU + frames/s at 1080p, Cornell Box 4spp/8bounce",
            "value": rays_total / elapsed / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "frames_per_s": args.steps / elapsed, "mrays_per_s_per_gpu": rays_total / elapsed / 1e6 / world,
            "rays_per_frame": rays_total / args.steps, "host_enqueue_ms_per_step": enqueue_s / args.steps * 1e3,
            "config": {"workload": desc, "width": W, "height": H, "spp": spp, "bounces": bounces, "frames_in_flight": len(lanes), "round_chains": args.chains or "library's choice",
                       "scene_copies_per_gpu": 1, "russian_roulette": True, "triangles": scene.triangle_count, "instances": len(scene.objects),
                       "bvh": {"node_bytes": acc.NodeSizeBytes, "nodes_total_bytes": acc.NodeBytes, "triangles_total_bytes": acc.TriangleBytes,
                               "bottom_level_depth": acc.MaxBottomLevelDepth, "top_level_depth": acc.TopLevelDepth, "traversal_copy_bytes": acc.BlobBytes},
                       "sharding": f"{BAND}-row bands, band b -> rank b % {world}" + (", pt_gather_bands (grouped ncclSend / ncclRecv into rank 0's frame)" if collective else ""),
                       "rccl_world_size": dist.get_world_size() if collective else 1,
                       "parity": "bit-identical to oracle on this scene (tests/test_gpu_parity.py, tests/test_gpu_fullscale.py)"},
        }
        if dynamic and dyn_events:
            ms = np.array([[e[k].elapsed_time(e[k + 1]) for k in range(3)] for e in dyn_events[-args.steps:]])
            result["dynamic"] = {"skin_ms": float(ms[:, 0].mean()), "update_and_top_level_ms": float(ms[:, 1].mean()), "render_ms": float(ms[:, 2].mean()),
                                 "note": "HIP events on the frame's stream: pt_skin_mesh | pt_update_bottom_level (in-place refit) + pt_build_top_level | G-buffer + path tracer"}

    # ---- roofline (N = 1). Everything below runs AFTER the timed region, on lane 0 alone.
    instrumented = world == 1 and not collective and not args.emulate_world and not args.no_kernel_timing and not dynamic
    if rank == 0 and instrumented:
        ctx.set_frames_in_flight(1)                              # from here on lane 0 has the GPU to itself

        def frames_on_lane0(n, first):
            with torch.cuda.stream(lanes[0].stream):
                for i in range(n):
                    lanes[0].renderer.render(S.graphics_settings(W, H, spp=spp, bounces=bounces, frame_index=first + i, ext_flags=ext))
            torch.cuda.synchronize(device)

        # (1) one frame at a time, graph replay, the GPU to itself: the latency figure next to the pipelined throughput
        frames_on_lane0(2, args.warmup)
        t1 = time.perf_counter(); frames_on_lane0(args.steps, args.warmup); latency_ms = (time.perf_counter() - t1) / args.steps * 1e3
        # (2) traversal statistics of one frame (nodes / triangles actually fetched)
        ctx.set_debug_flags(1 | BASE_FLAGS); ctx.reset_counters()
        frames_on_lane0(1, args.warmup + args.steps - 1)
        cs = ctx.counters()
        ctx.set_debug_flags(BASE_FLAGS)
        # (3) per-launch HIP events (recorded by the library on the stream the kernels run on), K frames, single stream
        ctx.reset_counters(); ctx.enable_kernel_timing(True)
        frames_on_lane0(args.steps, args.warmup)
        ks = ctx.kernel_timing(); c1 = ctx.counters()
        ctx.enable_kernel_timing(False)
        solo_secondary, solo_primary = float(c1.SecondaryRays), float(c1.PrimaryRays)

        all_rays = float(cs.PrimaryRays + cs.SecondaryRays)
        bvh_bytes_per_ray = (cs.NodesVisited * acc.NodeSizeBytes + cs.TrianglesTested * acc.TriangleSizeBytes) / all_rays
        acc1 = ctx.accel_stats()                                  # after the first frames: the normal records exist now
        blob_in_lds = acc.BlobBytes <= 40 * 1024                 # kBlobLdsMax: the traversal copy is staged into LDS by every block
        geometry_in_lds = blob_in_lds and acc1.RoundObjectsInLds > 0 and acc1.RoundRecordsInLds > 0
        # Algorithmic HBM bytes: what the data layout forces through HBM, nothing that is served from LDS.
        # Queue bytes of one secondary ray: ray read 32 + path state 48 r + 48 w + next ray written 32 = 160 (the hit stays in registers in
        # the fused round; the two-kernel form adds hit record 16 w + 16 r: traversal 32 + 16, shading 16 + ray direction 16 + 96 + 32).
        # Hit geometry, SURVEY 8(d)'s 3 x 32 B vertices + 12 B indices = 108 B: counted ONLY when a hit's shading fetches it from memory --
        # the fused round kernel on a small scene holds the traversal copy, the object table and the frame's normal records in LDS and
        # fetches nothing (counting them credited the kernel with bytes it never moved).
        QUEUE, GEOM, CONTRACT = 160.0, 108.0, 300.0              # CONTRACT: SURVEY 8(d)'s B_state per ray (adds the 32 B hit record round trip), reported apart
        geom = 0.0 if geometry_in_lds else GEOM
        STATE = {"k_round": QUEUE + geom, "k_extend": 48.0, "k_shade": QUEUE + GEOM}

        def gbps(nbytes, ms):
            return nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0

        # Pixels whose primary ray hit: counted on the device (finite Position.w), not assumed -- the cameras of C3 / C5 see sky.
        hit_pixels = float(torch.isfinite(lanes[0].renderer.textures["Position"].view(torch.float32).reshape(-1, 4)[:, 3]).sum().item())
        # the shading kernel of a round (k_round, or k_shade in the two-kernel form) also restarts the paths whose sample ended: one fresh
        # entry per primary-hit pixel and sample after the first over a frame -- primary-surface record 48 r, state 48 r + 48 w, first ray 32 w
        FRESH = 48.0 + 48.0 + 48.0 + 32.0
        fresh_entries = hit_pixels * (spp - 1) * args.steps     # over the timed single-stream pass
        # BVH bytes of a traversal launch that HBM has to deliver: what the rays touch, but never more than the traversal copy itself -- a byte of it
        # that L2 / Infinity Cache hold is fetched from HBM at most once per launch, however many rays read it (C3: 438 k rays x 1.42 KB touch
        # 620 MB of an 18.6 MB copy). SURVEY 8(d)'s B_bvh, every touched byte, is what contract_frac prices.
        def bvh_hbm_bytes(launches):
            if blob_in_lds or not launches:
                return 0.0
            return launches * min(solo_secondary / launches * bvh_bytes_per_ray, float(acc.BlobBytes))

        kernels = {}
        for name, key in (("k_round", "round"), ("k_extend", "extend"), ("k_shade", "shade")):
            n = ks.get(key + "_launches", 0)
            if n:
                traversal = name in ("k_round", "k_extend")
                fresh = fresh_entries * FRESH if name in ("k_round", "k_shade") else 0.0
                kernels[name] = {"ms": ks[key + "_ms"], "launches": n, "bytes": solo_secondary * STATE[name] + (bvh_hbm_bytes(n) if traversal else 0.0) + fresh,
                                 "contract_bytes": solo_secondary * ((CONTRACT if name != "k_extend" else STATE[name]) + (bvh_bytes_per_ray if traversal else 0.0)) + fresh}
        dom = max(kernels, key=lambda k: kernels[k]["ms"])
        kd = kernels[dom]
        # step level: every byte a frame has to move through HBM by construction of the data layout
        shade_per_ray = (QUEUE + geom) if "k_round" in kernels else (QUEUE + 32.0 + GEOM)     # two-kernel form: + hit record w + r
        frame_bytes = (W * H * 63.0 + hit_pixels * (47.0 + 48.0 + 48.0 + 32.0)          # G-buffer stores; k_pt_first: G-buffer read, primary-surface record written, the first sample's first bounce shaded in place: state + ray written
                       + solo_secondary / args.steps * shade_per_ray                     # traced entries
                       + bvh_hbm_bytes(ks.get("extend_launches", 0) or ks.get("round_launches", 0)) / args.steps   # BVH bytes HBM must deliver (0: LDS-resident)
                       + hit_pixels * (spp - 1) * FRESH                                  # fresh entries of the later samples
                       + W * H * 8.0)                                                    # radiance out
        contract_bytes = solo_secondary / args.steps * (CONTRACT + bvh_bytes_per_ray) + W * H * (63.0 + 8.0 + 32.0)
        step_achieved = gbps(frame_bytes, latency_ms)
        traffic, traffic_note, valu, lanes_per_instr = None, None, None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                ent = tj.get(args.workload, {})
                if ent.get("source_hash") == source_hash():
                    traffic = ent.get(dom); valu = ent.get("valu"); lanes_per_instr = ent.get("lanes_per_instruction")
                else:
                    traffic_note = "profiles/traffic.json was taken from other kernel sources than this build: PMC figures omitted"
            except Exception:
                traffic = None
        props = torch.cuda.get_device_properties(device)
        clock_hz = float(getattr(props, "clock_rate", 2400000)) * 1e3
        dom_ms = kd["ms"] / max(1, kd["launches"])
        result["roofline"] = {
            "bound": "hbm", "kernel": dom, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            # the contract's fields are the DOMINANT KERNEL's: algorithmic bytes per launch / its average launch duration (HIP events on its stream)
            "achieved": gbps(kd["bytes"], kd["ms"]), "frac": gbps(kd["bytes"], kd["ms"]) / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_frac": (traffic / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
            "contract_frac": (gbps(kd["contract_bytes"], kd["ms"]) / HBM_PEAK_GBS) if kd.get("contract_bytes") else None,
            "definition": "achieved = algorithmic HBM bytes per launch of the dominant kernel / its average launch duration (per-launch HIP events on the "
                          "kernel's stream, single stream, after the timed region); bytes served from LDS are not counted (geometry_served_from); "
                          "traffic = PMC (2 x FETCH_SIZE + WRITE_SIZE) per launch from profiles/traffic.json, traffic_frac = that / the same duration; "
                          "contract_frac = SURVEY 8(d)'s B_state (300 B per ray) + B_bvh (every BVH byte the rays touch, wherever it is served from: LDS on small scenes, "
                          "mostly L2 on large ones -- it can exceed 1); frame.* = the same for a whole frame",
            "geometry_served_from": "LDS (object table + normal records staged per launch): not counted" if geometry_in_lds else "memory: 108 B per ray counted",
            "dominant_kernel": {"name": dom, "avg_launch_ms": dom_ms, "launches_timed": kd["launches"],
                                "algorithmic_bytes_per_launch": kd["bytes"] / max(1, kd["launches"]),
                                "rays_per_launch": solo_secondary / max(1, kd["launches"]),
                                "fresh_entries_per_launch": (fresh_entries / max(1, kd["launches"])) if dom in ("k_round", "k_shade") else 0.0,
                                "achieved": gbps(kd["bytes"], kd["ms"]), "frac": gbps(kd["bytes"], kd["ms"]) / HBM_PEAK_GBS},
            "other_kernels": {k: {"avg_launch_ms": v["ms"] / max(1, v["launches"]), "launches_timed": v["launches"], "achieved": gbps(v["bytes"], v["ms"]),
                                  "frac": gbps(v["bytes"], v["ms"]) / HBM_PEAK_GBS}
                              for k, v in kernels.items() if k != dom},
            "frame": {"hbm_bytes_per_frame": frame_bytes, "contract_bytes_per_frame": contract_bytes, "primary_hit_pixels": hit_pixels,
                      "latency_ms_one_frame": latency_ms, "achieved": step_achieved, "frac": step_achieved / HBM_PEAK_GBS,
                      "contract_frac": gbps(contract_bytes, latency_ms) / HBM_PEAK_GBS,
                      "mrays_per_s_one_frame_at_a_time": (solo_secondary + solo_primary) / args.steps / (latency_ms * 1e-3) / 1e6},
            "pipelined": {"frames_in_flight": len(lanes), "ms_per_step": ms_per_step, "achieved": gbps(frame_bytes, ms_per_step),
                          "frac": gbps(frame_bytes, ms_per_step) / HBM_PEAK_GBS},
            "bvh": {"bytes_per_ray": bvh_bytes_per_ray, "nodes_per_ray": cs.NodesVisited / all_rays, "tris_per_ray": cs.TrianglesTested / all_rays,
                    "longest_walk_nodes": cs.MaxNodesPerRay or None,
                    "traversal_copy_bytes": acc.BlobBytes,
                    "served_from": "LDS (the traversal copy is staged by every block): not HBM traffic" if blob_in_lds
                                   else "L2 / Infinity Cache / HBM: per launch, min(bytes touched, traversal_copy_bytes) counted as HBM bytes; every touched byte in contract_frac"},
        }
        if traffic_note:
            result["roofline"]["traffic_note"] = traffic_note
        # VALU issue: wave64 instructions per second against what the chip was MEASURED to issue (tools/valu_peak.hip ->
        # profiles/r03_valu_peak.json). Two ceilings: independent v_fma_f32 at 8 waves per SIMD (the chip's best), and the mixed stream
        # (cvt / fma / min / max / cmp / cndmask, the node test's proportions) at the 4 waves per SIMD the render kernels hold -- the
        # ceiling these kernels can actually reach at their occupancy.
        if valu:
            nominal = props.multi_processor_count * 4 * clock_hz / 2.0
            peak_issue, peak_occ, peak_src = nominal, None, "nominal: CUs x 4 SIMD-32 x clock / 2 cycles per wave64 instruction"
            vp = next((p for p in (os.path.join(ROOT, "profiles", n) for n in ("r04_valu_peak.json", "r03_valu_peak.json")) if os.path.exists(p)), "")
            if os.path.exists(vp):
                try:
                    vj = json.load(open(vp))
                    peak_issue = vj["fma_indep"]["waves_per_simd_8"] * 1e9
                    peak_occ = vj["mixed"]["waves_per_simd_4"] * 1e9
                    peak_src = "measured, profiles/r03_valu_peak.json: fma_indep at 8 waves per SIMD | mixed at 4 waves per SIMD, all CUs"
                except Exception:
                    pass
            per_frame = valu["wave_instructions_per_frame"]
            alone, piped = per_frame / (latency_ms * 1e-3), per_frame / (ms_per_step * 1e-3)
            result["roofline"]["valu_issue"] = {
                "wave_instructions_per_frame": per_frame, "peak_per_s": peak_issue, "peak_per_s_at_occupancy": peak_occ, "peak_source": peak_src,
                "nominal_peak_per_s": nominal, "clock_hz": clock_hz,
                "one_frame_at_a_time": {"issued_per_s": alone, "frac": alone / peak_issue, "frac_at_occupancy": alone / peak_occ if peak_occ else None},
                "pipelined": {"issued_per_s": piped, "frac": piped / peak_issue, "frac_at_occupancy": piped / peak_occ if peak_occ else None},
                "lanes_per_instruction": lanes_per_instr,
                "note": "SQ_INSTS_VALU of every kernel of a frame (profiles/, same kernel sources) / frame time; lanes_per_instruction = "
                        "SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) per kernel from the same PMC files"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.emulate_world and not dynamic:
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
