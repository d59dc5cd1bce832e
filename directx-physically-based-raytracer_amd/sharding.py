"""Row-band framebuffer sharding across ranks (the multi-GPU design of SURVEY.md 8e; not a reference feature).

Band b (BandHeight rows) belongs to rank b % world; a rank stores its bands contiguously. The same mapping
is implemented in the library by pt_set_sharding / pt_gather_bands / pt_gather_plan (include/ptamd.h, csrc/pt_comm.hip:
the RCCL exchange that bench.py and the C++ host use); this module is the host-side statement of it, used by the tests:
the expectations of the GPU sharding tests, the check of pt_gather_plan, and the 2-process gloo test of the N > 1 path,
where gather_to_root stands in for the RCCL exchange.
"""
import numpy as np


def rank_bands(height, rank, world, band):
    """[(global_y0, global_y1, local_y0)] of the bands owned by `rank`."""
    out, local = [], 0
    b = rank
    while b * band < height:
        y0 = b * band
        y1 = min(height, y0 + band)
        out.append((y0, y1, local))
        local += y1 - y0
        b += world
    return out


def local_rows(height, rank, world, band):
    return sum(y1 - y0 for y0, y1, _ in rank_bands(height, rank, world, band))


def extract_local(full, rank, world, band):
    """rows of `full` ([H, ...]) owned by `rank`, in local order."""
    parts = [full[y0:y1] for y0, y1, _ in rank_bands(full.shape[0], rank, world, band)]
    return np.concatenate(parts, 0) if parts else full[:0]


def deinterleave(pieces, height, band):
    """inverse of extract_local: pieces[r] = rank r's local buffer (may be padded with extra rows)."""
    world = len(pieces)
    out = np.zeros((height,) + pieces[0].shape[1:], pieces[0].dtype)
    for r in range(world):
        for y0, y1, l0 in rank_bands(height, r, world, band):
            out[y0:y1] = pieces[r][l0:l0 + (y1 - y0)]
    return out


def gather_to_root(local, rank, world, dist, out=None):
    """One exchange step of the multi-GPU path: every rank's (equal-sized, possibly row-padded) local buffer
    goes to rank 0 (RCCL gather over xGMI on GPUs, gloo in the CPU tests). Returns the list of pieces on rank 0;
    `out` ([world, *local.shape], same dtype) makes them views of one contiguous buffer."""
    import torch
    raw = local.contiguous().view(torch.uint8)       # bytes on the wire: fp16 / snorm texels have no collective dtype
    if rank == 0 and out is not None:
        pieces = [out[r].view(torch.uint8) for r in range(world)]
    else:
        pieces = [torch.empty_like(raw) for _ in range(world)] if rank == 0 else None
    dist.gather(raw, pieces, dst=0)
    return [p.view(local.dtype).view(local.shape) for p in pieces] if rank == 0 else None
