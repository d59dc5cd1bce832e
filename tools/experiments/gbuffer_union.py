"""Experiment: what a wave-shared G-buffer traversal would have to visit. For some 8 x 8 pixel squares of a workload
the 64 primary rays are walked one by one through the library's logged one-lane walk (pt_debug_trace_ray); printed per square: node visits
and triangle tests of the mean and the longest lane -- what the per-lane walk of k_gbuffer costs a wave is its longest lane -- and the UNION
over the 64 lanes, which is what a walk with one shared stack executes (every lane tests every node and triangle any lane needs).
usage: tools/experiments/gbuffer_union.py [c3 c5]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
ge.load_package()
import dxpbrt_amd.ptamd as P, dxpbrt_amd.scenes as S
import bench


def ray(cam, px, py, W, H):
    u, v = (px + 0.5) / W, (py + 0.5) / H
    nx, ny = u * 2 - 1, 1 - v * 2
    d = cam["ForwardDirection"].astype(np.float64) + nx * cam["RightDirection"] + ny * cam["UpDirection"]
    d /= np.linalg.norm(d)
    return np.array([*cam["Position"], 0.0, *d, 1e30], np.float32)


for w in sys.argv[1:] or ["c3", "c5"]:
    kind, W, H, spp, bounces, desc = bench.WORKLOADS[w]
    scene, ext = bench.make_scene(kind, W / H, S)
    ctx = P.DeviceContext(0); g = P.Scene(ctx, scene); tlas = g.GetTopLevelAccelerationStructure(); ctx.sync()
    log = np.zeros(4 * 4096, np.uint32)
    rng = np.random.default_rng(7)
    rows = []
    for _ in range(24):
        x0, y0 = int(rng.integers(0, W // 8)) * 8, int(rng.integers(H // 3 // 8, H // 8)) * 8      # the lower two thirds of the frame: geometry, not sky
        nodes, tris, per_n, per_t = set(), set(), [], []
        for k in range(64):
            r = ray(scene.camera, x0 + k % 8, y0 + k // 8, W, H)
            log[:] = 0
            ctx.check(ctx.lib.pt_debug_trace_ray(ctx.handle, C.c_void_p(r.ctypes.data), C.c_void_p(log.ctypes.data), log.size))
            e = log.reshape(-1, 4)
            mine_n, mine_t = set(), set()
            inst = 0xFFFFFFFF
            for code, a, b, sp in e:
                if code == 0: break
                if code == 1: inst = int(a)
                elif code == 6: inst = 0xFFFFFFFF
                elif code == 2: mine_t.add((int(a), int(b)))
                elif code == 4: mine_n.add((inst, int(a), int(b)))          # (child base, triangle base) of the node just visited: its identity
            nodes |= mine_n; tris |= mine_t; per_n.append(len(mine_n)); per_t.append(len(mine_t))
        rows.append((np.mean(per_n), max(per_n), len(nodes), np.mean(per_t), max(per_t), len(tris)))
    a = np.array(rows)
    print(f"{w}: over {len(rows)} squares of 8 x 8 pixels: nodes per lane {a[:,0].mean():.1f}, longest lane {a[:,1].mean():.1f}, union of the 64 lanes {a[:,2].mean():.1f} | "
          f"triangles per lane {a[:,3].mean():.1f}, longest {a[:,4].mean():.1f}, union {a[:,5].mean():.1f}", flush=True)
    ctx.close()
