"""Experiment (potential only, not parity): what would splitting the long thin triangles of C3 buy? The scene GENERATOR subdivides every
triangle whose longest edge exceeds L by longest-edge bisection (real triangles, so images and ray counts differ slightly from C3's) and
the product library renders it; node visits per ray, longest walk and Mrays/s tell what a reference-splitting builder could reach at best.
usage: tools/experiments/presplit_potential.py [L ...]     (L = 0 : the scene as it is)"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as ge
ge.load_package()
import dxpbrt_amd.ptamd as P, dxpbrt_amd.scenes as S, dxpbrt_amd.layouts as L_
import bench


def subdivide(mesh, L):
    pos = mesh.vertices["Position"].astype(np.float64); nrm_src = mesh.vertices
    idx = mesh.indices.reshape(-1, 3).astype(np.int64)
    extra_pos, extra_from = [], []
    nv = len(pos)
    done = []
    work = idx
    allpos = pos
    while len(work):
        p = allpos[work]                                       # [n,3,3]
        e = np.stack([np.linalg.norm(p[:, 1] - p[:, 0], axis=1), np.linalg.norm(p[:, 2] - p[:, 1], axis=1), np.linalg.norm(p[:, 0] - p[:, 2], axis=1)], 1)
        k = e.argmax(1); long_ = e.max(1) > L
        done.append(work[~long_])
        w = work[long_]; k = k[long_]
        if not len(w): break
        a = w[np.arange(len(w)), k]; b = w[np.arange(len(w)), (k + 1) % 3]; c = w[np.arange(len(w)), (k + 2) % 3]
        mid = 0.5 * (allpos[a] + allpos[b])
        m = len(allpos) + np.arange(len(w))
        allpos = np.concatenate([allpos, mid]); extra_from.append(a)
        work = np.concatenate([np.stack([a, m, c], 1), np.stack([m, b, c], 1)])
    tri = np.concatenate(done)
    src = np.concatenate([np.arange(nv)] + extra_from) if extra_from else np.arange(nv)
    # attributes of a new vertex: copied from one end of the split edge (an end that is itself new points further back: resolve the chains)
    for _ in range(64):
        nxt = src[src]
        if np.array_equal(nxt, src): break
        src = nxt
    v = mesh.vertices[src].copy(); v["Position"] = allpos.astype(np.float32)
    return S.Mesh(v, S.make_indices(tri.reshape(-1)), mesh.has_normals, mesh.material)


for Lstr in sys.argv[1:] or ["0", "0.5", "0.25"]:
    Lv = float(Lstr)
    kind, W, H, spp, bounces, desc = bench.WORKLOADS["c3"]
    scene, ext = bench.make_scene(kind, W / H, S)
    if Lv > 0:
        scene.nodes[0].meshes = [subdivide(m, Lv) for m in scene.nodes[0].meshes]
        scene.finalize()
    lanes = []
    for _ in range(3):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            ctx = P.DeviceContext(0, stream=st.cuda_stream); ctx.set_frames_in_flight(3)
            g = P.Scene(ctx, scene); lanes.append((st, ctx, g, P.Renderer(ctx, g, W, H)))
    def frame(i):
        st, ctx, g, r = lanes[i % 3]
        with torch.cuda.stream(st):
            r.render(S.graphics_settings(W, H, spp=spp, bounces=bounces, frame_index=i, ext_flags=ext))
    for i in range(9): frame(i)
    torch.cuda.synchronize()
    for _, ctx, _, _ in lanes: ctx.reset_counters()
    t = time.perf_counter()
    for i in range(30): frame(100 + i)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    rays = sum(c.PrimaryRays + c.SecondaryRays for c in (ctx.counters() for _, ctx, _, _ in lanes))
    st_, ctx, g, r = lanes[0]
    with torch.cuda.stream(st_):
        ctx.set_debug_flags(1); ctx.reset_counters()
        r.render(S.graphics_settings(W, H, spp=spp, bounces=bounces, frame_index=7, ext_flags=ext)); ctx.sync()
        c = ctx.counters(); n = c.PrimaryRays + c.SecondaryRays
    acc = ctx.accel_stats()
    print(f"L={Lv}: {scene.triangle_count} triangles, nodes {acc.NodeBytes // 80}, depth {acc.MaxBottomLevelDepth} | {rays / dt / 1e6:.0f} Mrays/s | "
          f"node visits/ray {c.NodesVisited / n:.2f}, triangle tests/ray {c.TrianglesTested / n:.2f}, longest walk {c.MaxNodesPerRay} nodes", flush=True)
    for _, ctx, _, _ in lanes: ctx.close()
