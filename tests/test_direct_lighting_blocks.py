"""SURVEY 8f rank 4 building blocks: shadow-ray TraceRay with coloured visibility (RTXDIAppBridge.hlsli:418-439,
ShadingHelpers.hlsli:117-159) and the all-lobe BSDFSample::Evaluate / EvaluatePDF (BxDF.hlsli:247-285)."""
import ctypes as C

import numpy as np
import pytest

import __graft_entry__ as ge


def make_queries(n, seed=11):
    rng = np.random.default_rng(seed)
    q = np.zeros((n, 20), np.float32)
    q[:, 0:3] = rng.random((n, 3)); q[:, 3] = rng.choice([0.0, 0.3, 1.0], n); q[:, 4] = rng.random(n)
    q[:, 5] = 1.0 + rng.random(n); q[:, 6] = rng.choice([0.0, 0.0, 0.6, 1.0], n); q[:, 7] = rng.integers(0, 2, n)
    def unit(k):
        v = rng.standard_normal((n, 3)); return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    ns = unit(0); ng = ns + 0.1 * unit(1); ng /= np.linalg.norm(ng, axis=1, keepdims=True)
    q[:, 8:11] = ng; q[:, 11:14] = ns; q[:, 14:17] = unit(2); q[:, 17:20] = unit(3)
    return q


def test_all_lobe_evaluate_is_the_sum_of_single_lobes(oracle):
    lib = oracle.lib()
    q = make_queries(400)
    r = np.zeros((len(q), 8), np.float32)
    lib.or_bsdf_evaluate(q.ctypes.data, len(q), r.ctypes.data)
    assert np.isfinite(r).all() and (r[:, :7] >= 0).all()
    # opaque metals have no diffuse term; fully transmissive dielectrics only a "specular" (transmission) term
    metal = (q[:, 3] == 1.0)
    assert np.all(r[metal, 0:3] == 0)
    glass = (q[:, 6] == 1.0) & (q[:, 3] == 0.0)
    assert np.all(r[glass, 0:3] == 0) and np.all(r[glass, 6] > 0)
    # below the geometric surface of an opaque material everything vanishes
    ngf = np.where(q[:, 7:8] != 0, q[:, 8:11], -q[:, 8:11])
    below = ((ngf * q[:, 17:20]).sum(1) <= 0) & (q[:, 6] == 0.0)
    assert below.any() and np.all(r[below, :7] == 0)


@pytest.mark.gpu
def test_gpu_bsdf_evaluate_matches_oracle(gpu, oracle):
    import torch
    q = make_queries(5000)
    ref = np.zeros((len(q), 8), np.float32)
    oracle.lib().or_bsdf_evaluate(q.ctypes.data, len(q), ref.ctypes.data)
    dq = torch.from_numpy(q).cuda(); dr = torch.zeros((len(q), 8), dtype=torch.float32, device="cuda")
    gpu.check(gpu.lib.pt_bsdf_evaluate(gpu.handle, C.c_void_p(dq.data_ptr()), len(q), C.c_void_p(dr.data_ptr())))
    gpu.sync()
    assert np.array_equal(dr.cpu().numpy().view(np.uint32), ref.view(np.uint32))


@pytest.mark.gpu
def test_gpu_visibility_rays_match_oracle(gpu, ptamd, oracle, pkg):
    import torch
    S = pkg.scenes
    scene = S.cornell_box_textured(env=None)                    # opaque walls, an alpha-masked lattice, a transmissive pane, a metal floor
    rng = np.random.default_rng(5)
    n = 20000
    a = (rng.random((n, 3)) * 1.9 - 0.95).astype(np.float32); b = (rng.random((n, 3)) * 1.9 - 0.95).astype(np.float32)
    d = b - a; ln = np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3] = a; rays[:, 3] = 1e-3; rays[:, 4:7] = d / ln; rays[:, 7] = np.maximum(0, ln[:, 0] - 2e-3)      # CreateVisibilityRay
    osc = oracle.OracleScene(scene, accel_mode=0)
    ref = np.zeros((n, 4), np.float32)
    oracle.lib().or_trace_visibility(osc.handle, rays.ctypes.data, n, ref.ctypes.data)
    gpu.set_sharding(0, 1, 16)
    g = ptamd.Scene(gpu, scene)
    dr = torch.from_numpy(rays).cuda(); dv = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    gpu.check(gpu.lib.pt_trace_visibility(gpu.handle, C.c_void_p(dr.data_ptr()), n, C.c_void_p(dv.data_ptr())))
    gpu.sync()
    got = dv.cpu().numpy()
    assert np.array_equal(got[:, 3], ref[:, 3])                  # occluded / unoccluded: exact
    assert np.allclose(got[:, :3], ref[:, :3], rtol=1e-6, atol=0)     # products of >= 3 transmittances may associate differently
    frac_clear = ref[:, 3].mean()
    assert 0.2 < frac_clear < 0.95
    partial = (ref[:, 3] == 1) & (ref[:, :3].max(1) < 1)
    assert partial.any()                                         # rays through the pane: coloured, not binary, visibility
    osc.close()
