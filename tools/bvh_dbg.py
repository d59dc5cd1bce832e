"""Developer helper: structural check of the built BVH + device brute force vs BVH on bounce rays; dumps blob + first mismatch."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
ge.load_package()
import dxpbrt_amd.layouts as L, dxpbrt_amd.ptamd as P, dxpbrt_amd.scenes as S
import bvh_check

def run(scene, W, H, spp, bounces, tag):
    gs = S.graphics_settings(W, H, spp=spp, bounces=bounces)
    ctx = P.DeviceContext(0)
    g = P.Scene(ctx, scene)
    r = P.Renderer(ctx, g, W, H, with_f32=True)
    st = ctx.accel_stats()
    print(scene.name, "instances", st.InstanceCount, "blas depth", st.MaxBottomLevelDepth, "tlas depth", st.TopLevelDepth, "blob", st.BlobBytes, flush=True)
    lay, buf = ctx.download_blob()
    np.savez_compressed(f"gpurun_out/blob_{tag}.npz", buf=buf, lay=np.array([getattr(lay, n) for n, _ in lay._fields_], np.uint32))
    ctx.set_debug_flags(2); ctx.reset_counters()
    r.render(gs); ctx.sync()
    c = ctx.counters()
    print("  rays", c.SecondaryRays, "mismatches", c.BvhMismatches, "overflows", c.StackOverflows, flush=True)
    if c.BvhMismatches:
        m = np.zeros(16, np.float32)
        ctx.lib.pt_debug_read_mismatch(ctx.handle, m.ctypes.data)
        u = m.view(np.uint32)
        print("  ray o", m[0:3], "tmin", m[3], "d", m[4:7], "tmax", m[7])
        print("  bvh inst", u[8], "slot", u[9], "t", m[10], "| brute inst", u[12], "slot", u[13], "t", m[14])
        np.save(f"gpurun_out/mismatch_{tag}.npy", m)
    ctx.set_debug_flags(0)
    ctx.close()

if __name__ == "__main__":
    pass

def trace_log(scene, ray8, n=400):
    ctx = P.DeviceContext(0)
    g = P.Scene(ctx, scene)
    r = P.Renderer(ctx, g, 16, 16)
    r.render(S.graphics_settings(16, 16, spp=1, bounces=1)); ctx.sync()      # object data / heap bound
    log = np.zeros(4 * n, np.uint32)
    ray = np.asarray(ray8, np.float32)
    ctx.check(ctx.lib.pt_debug_trace_ray(ctx.handle, ray.ctypes.data, log.ctypes.data, log.size))
    for row in log.reshape(-1, 4):
        if row[0] == 0: break
        print("   ", ["", "enter", "tri", "visit", "after", "pop", "restore", "result"][row[0]], hex(row[1]), hex(row[2]), "sp", row[3])
    ctx.close()

if __name__ == "__main__" and os.path.exists("tools/_mismatch_grid20.npy"):
    m = np.load("tools/_mismatch_grid20.npy")
    trace_log(S.instanced_grid(n=20), m[:8])
