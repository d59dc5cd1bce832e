"""Developer aid: list visibility rays whose rgb differs between the device and the oracle."""
import sys, os, numpy as np, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.load_package()
import torch
import dxpbrt_amd.ptamd as P, dxpbrt_amd.scenes as S
orc = ge.load_oracle()
scene = S.cornell_box_textured(env=None)
rng = np.random.default_rng(5)
n = 20000
a = (rng.random((n, 3)) * 1.9 - 0.95).astype(np.float32); b = (rng.random((n, 3)) * 1.9 - 0.95).astype(np.float32)
d = b - a; ln = np.linalg.norm(d, axis=1, keepdims=True)
rays = np.zeros((n, 8), np.float32)
rays[:, 0:3] = a; rays[:, 3] = 1e-3; rays[:, 4:7] = d / ln; rays[:, 7] = np.maximum(0, ln[:, 0] - 2e-3)
osc = orc.OracleScene(scene, accel_mode=0)
ref = np.zeros((n, 4), np.float32)
orc.lib().or_trace_visibility(osc.handle, rays.ctypes.data, n, ref.ctypes.data)
P.load_library()
ctx = P.DeviceContext(0)
ctx.set_sharding(0, 1, 16)
g = P.Scene(ctx, scene)
dr = torch.from_numpy(rays).cuda(); dv = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
ctx.check(ctx.lib.pt_trace_visibility(ctx.handle, C.c_void_p(dr.data_ptr()), n, C.c_void_p(dv.data_ptr())))
ctx.sync()
got = dv.cpu().numpy()
bad = np.argwhere(~np.isclose(got[:, :3], ref[:, :3], rtol=1e-6, atol=0).all(1)).reshape(-1)
print("bad", len(bad), "of", n, "flagdiff", int((got[:, 3] != ref[:, 3]).sum()))
for i in bad[:10]:
    print(i, "gpu", got[i], "ref", ref[i], "ray", rays[i])
