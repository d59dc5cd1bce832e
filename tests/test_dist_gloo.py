"""N > 1 path on CPU: world_size 2 over gloo. Each rank renders only its own row bands (the oracle stands in for
the per-rank renderer here), the bands are gathered to rank 0 and de-interleaved with the same sharding rules
bench.py uses on GPUs; the result must equal the single-process frame bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, BAND, WORLD = 48, 40, 8, 2


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, out_path):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    ge.load_package()
    import dxpbrt_amd.layouts as L
    import dxpbrt_amd.scenes as S
    import dxpbrt_amd.sharding as SH
    oracle = ge.load_oracle()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    scene = S.cornell_box(aspect=W / H, glass_sphere=True)
    gs = S.graphics_settings(W, H, spp=2, bounces=4, frame_index=3)
    gb = S.alloc_gbuffer(W, H)
    consts = np.zeros((), L.GBUFFER_CONSTANTS); consts["RenderSize"] = (W, H); consts["Flags"] = L.GBufferFlags.DefaultNoDenoiser
    osc = oracle.OracleScene(scene, accel_mode=1)
    rays = 0
    for y0, y1, _ in SH.rank_bands(H, rank, world, BAND):       # only this rank's rows
        rays += osc.gbuffer(consts, gb, rows=(y0, y1), threads=1)
        rays += osc.raytrace(gs, gb, rows=(y0, y1), threads=1)
    max_rows = max(SH.local_rows(H, r, world, BAND) for r in range(world))
    local = np.zeros((max_rows, W, 4), np.uint16)
    mine = SH.extract_local(gb["Radiance"], rank, world, BAND)
    local[:mine.shape[0]] = mine
    pieces = SH.gather_to_root(torch.from_numpy(local.view(np.int16)), rank, world, dist)
    total = torch.tensor([rays], dtype=torch.int64)
    dist.all_reduce(total)
    if rank == 0:
        frame = SH.deinterleave([p.numpy().view(np.uint16) for p in pieces], H, BAND)
        np.savez(out_path, frame=frame, rays=int(total[0]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_band_sharding_equals_single_process(tmp_path, oracle, pkg):
    import torch.multiprocessing as mp
    out = str(tmp_path / "frame.npz")
    mp.spawn(_worker, args=(WORLD, _free_port(), out), nprocs=WORLD, join=True)
    got = np.load(out)
    S, L = pkg.scenes, pkg.layouts
    scene = S.cornell_box(aspect=W / H, glass_sphere=True)
    gb, rays, _ = oracle.render(scene, S.graphics_settings(W, H, spp=2, bounces=4, frame_index=3), accel_mode=1, layouts=L)
    assert int(got["rays"]) == rays
    assert np.array_equal(got["frame"], gb["Radiance"])
