"""Decoder + structural checker of the compressed 8-wide BVH the device builder produces (include/ptamd.h pt_debug_download_blob).

Used by the GPU tests: after a build the blob is downloaded and every invariant the traversal relies on is checked on the CPU --
child boxes (decoded exactly as the kernel decodes them) contain what hangs below them, child / triangle references stay inside
their tree, every triangle and every instance is reachable exactly once, depth fits the traversal stack."""
import numpy as np

NODE_DT = np.dtype([("origin", "<f4", 3), ("exp", "u1", 3), ("imask", "u1"), ("childBase", "<u4"), ("triBase", "<u4"), ("meta", "u1", 8),
                    ("qlox", "u1", 8), ("qloy", "u1", 8), ("qloz", "u1", 8), ("qhix", "u1", 8), ("qhiy", "u1", 8), ("qhiz", "u1", 8)])
INST_DT = np.dtype([("worldToObject", "<f4", 12), ("boxLo", "<f4", 3), ("nodeBase", "<u4"), ("boxHi", "<f4", 3), ("triBase", "<u4"),
                    ("mask", "<u4"), ("triCount", "<u4"), ("instanceID", "<u4"), ("instanceIndex", "<u4"), ("objectToWorld", "<f4", 12)])
TRI_DT = np.dtype([("v0", "<f4", 3), ("geom", "<u4"), ("v1", "<f4", 3), ("prim", "<u4"), ("v2", "<f4", 3), ("flags", "<u4")])
assert NODE_DT.itemsize == 80 and INST_DT.itemsize == 144 and TRI_DT.itemsize == 48


def split(layout, buf):
    inst = buf[layout.InstanceOffset16 * 16:][:layout.InstanceCount * 144].view(INST_DT)
    nodes = buf[layout.NodeOffset16 * 16:][:layout.NodeCount * 80].view(NODE_DT)
    tris = buf[layout.TriangleOffset16 * 16:][:layout.TriangleCount * 48].view(TRI_DT)
    leaf = buf[layout.LeafInstanceOffset16 * 16:][:layout.InstanceCount * 144].view(INST_DT)
    order = leaf["instanceIndex"].astype(np.uint32)             # the TLAS's items: the instance records once more, in leaf order
    if len(inst):
        assert np.array_equal(inst["instanceIndex"], np.arange(len(inst), dtype=np.uint32)), "instance records are not in API order"
        if sorted(order.tolist()) == list(range(len(inst))):
            assert leaf.tobytes() == inst[order].tobytes(), "leaf-order instance records differ from the API-order ones"
    return inst, nodes, tris, order


def child_boxes(n):
    """decoded child boxes of one node: lo[8,3], hi[8,3] in float64 (exact: origin + q * 2^e)"""
    scale = np.ldexp(1.0, n["exp"].astype(np.int64) - 127)
    qlo = np.stack([n["qlox"], n["qloy"], n["qloz"]], -1).astype(np.float64)
    qhi = np.stack([n["qhix"], n["qhiy"], n["qhiz"]], -1).astype(np.float64)
    o = n["origin"].astype(np.float64)
    return o + qlo * scale, o + qhi * scale


def walk(nodes, node_base, leaf_box, leaf_count, single_leaf, max_depth=64):
    """Walks the tree whose root is nodes[node_base]. leaf_box(first, count) -> (lo, hi) of items [first, first + count).
    Returns (visited item indices, node count, depth, problems)."""
    problems, seen = [], []
    if single_leaf:
        return list(range(leaf_count)), 0, 0, problems
    stack = [(0, 1, None, None)]
    visited_nodes, depth = 0, 0
    while stack:
        idx, d, plo, phi = stack.pop()
        depth = max(depth, d)
        if d > max_depth:
            problems.append(f"node {idx} deeper than {max_depth}"); continue
        n = nodes[node_base + idx]
        visited_nodes += 1
        lo, hi = child_boxes(n)
        if plo is not None:          # a ray reaches a slot only through every ancestor slot: what counts is the intersection of them all
            lo = np.maximum(lo, plo); hi = np.minimum(hi, phi)       # (a child's own grid may round its slots a quantum beyond the parent's slot)
        rank = 0
        for s in range(8):
            m = int(n["meta"][s])
            if m == 0:
                continue
            if (m & 0x1F) >= 24:             # internal
                if (m >> 5) != 1 or (m & 0x1F) != 24 + s or not (n["imask"] >> s) & 1:
                    problems.append(f"node {idx} slot {s}: malformed internal meta {m:#x} / imask {int(n['imask']):#x}")
                stack.append((int(n["childBase"]) + rank, d + 1, lo[s], hi[s]))
                rank += 1
            else:
                if (n["imask"] >> s) & 1:
                    problems.append(f"node {idx} slot {s}: leaf slot flagged internal")
                cnt = {1: 1, 3: 2, 7: 3}.get(m >> 5)
                if cnt is None:
                    problems.append(f"node {idx} slot {s}: malformed leaf meta {m:#x}"); continue
                first = int(n["triBase"]) + (m & 0x1F)
                if first + cnt > leaf_count:
                    problems.append(f"node {idx} slot {s}: items {first}..{first + cnt} beyond {leaf_count}"); continue
                blo, bhi = leaf_box(first, cnt)
                if (blo < lo[s]).any() or (bhi > hi[s]).any():
                    problems.append(f"node {idx} slot {s}: items {first}+{cnt} stick out of the slot box by {max((lo[s] - blo).max(), (bhi - hi[s]).max()):.3e}")
                seen.extend(range(first, first + cnt))
    return seen, visited_nodes, depth, problems


def check_blob(layout, buf, leaf_tris_rule):
    """Full check. leaf_tris_rule(triCount) -> triangles per leaf (pt_trace.hpp blas_leaf_tris). Returns a dict of statistics; raises
    AssertionError listing the problems found."""
    inst, nodes, tris, order = split(layout, buf)
    problems = []
    stats = {"instances": len(inst), "blas_depth": 0, "tlas_depth": 0, "blas_nodes": 0}
    checked = {}
    for i, it in enumerate(inst):
        key = (int(it["nodeBase"]), int(it["triBase"]), int(it["triCount"]))
        if key in checked or key[2] == 0:
            continue
        tb, tc = key[1], key[2]
        t = tris[tb:tb + tc]
        v = np.stack([t["v0"], t["v1"], t["v2"]], 1).astype(np.float64)            # [tc, 3, 3]
        tlo, thi = v.min(1), v.max(1)

        def leaf_box(first, cnt, tlo=tlo, thi=thi):
            return tlo[first:first + cnt].min(0), thi[first:first + cnt].max(0)
        seen, nn, depth, pr = walk(nodes, key[0], leaf_box, tc, tc <= leaf_tris_rule(tc))
        problems += [f"BLAS@{key[0]}: {p}" for p in pr[:5]]
        if sorted(seen) != list(range(tc)):
            problems.append(f"BLAS@{key[0]}: {tc} triangles but the tree reaches {len(seen)} ({len(set(seen))} distinct)")
        if len(np.unique(np.stack([t["geom"], t["prim"]], -1), axis=0)) != tc:
            problems.append(f"BLAS@{key[0]}: duplicate (geometry, primitive) pairs among the packets")
        checked[key] = depth
        stats["blas_depth"] = max(stats["blas_depth"], depth); stats["blas_nodes"] += nn
    if len(inst):
        ilo, ihi = inst["boxLo"].astype(np.float64), inst["boxHi"].astype(np.float64)
        # the world box of an instance (built from two levels of its BLAS's child boxes, pt_bvh.hip instance_world_box) must contain
        # every triangle of the BLAS pushed through ObjectToWorld: checked on the vertices themselves, not on any box of them
        worst = 0.0
        for key in {(int(it["triBase"]), int(it["triCount"])) for it in inst if int(it["triCount"])}:
            t = tris[key[0]:key[0] + key[1]]
            v = np.concatenate([t["v0"], t["v1"], t["v2"]], 0).astype(np.float64)     # [3 tc, 3] object space
            ids = np.nonzero((inst["triBase"] == key[0]) & (inst["triCount"] == key[1]))[0]
            for c0 in range(0, len(ids), 512):
                sel = ids[c0:c0 + 512]
                M = inst["objectToWorld"][sel].astype(np.float64).reshape(-1, 3, 4)
                w = np.einsum("iak,vk->iva", M[:, :, :3], v) + M[:, None, :, 3]            # [n, 3 tc, 3] world space
                out = np.maximum(ilo[sel][:, None, :] - w, w - ihi[sel][:, None, :]).max((1, 2))
                bad = np.nonzero(out > 0)[0]
                worst = max(worst, float(out.max()))
                for b in bad[:3]:
                    problems.append(f"instance {int(sel[b])}: a vertex of its BLAS sticks out of its world box by {out[b]:.3e}")
        stats["instance_box_slack_min"] = -worst
        if sorted(order.tolist()) != list(range(len(inst))):
            problems.append("instance order list is not a permutation")

        def inst_box(first, cnt):
            ids = order[first:first + cnt]
            ok = ilo[ids, 0] <= ihi[ids, 0]                 # empty instances (no triangles) have inverted boxes and need no cover
            if not ok.any():
                return np.full(3, np.inf), np.full(3, -np.inf)
            return ilo[ids][ok].min(0), ihi[ids][ok].max(0)
        seen, nn, depth, pr = walk(nodes, 0, inst_box, len(inst), len(inst) == 1)
        problems += [f"TLAS: {p}" for p in pr[:8]]
        if sorted(seen) != list(range(len(inst))):
            problems.append(f"TLAS: {len(inst)} instances but the tree reaches {len(seen)} ({len(set(seen))} distinct)")
        stats["tlas_depth"] = depth; stats["tlas_nodes"] = nn
    assert not problems, "\n".join(problems[:20])
    return stats


# ---------------------------------------------------------------------------------------------
# CPU emulation of the kernels' box traversal over a downloaded blob (float32 arithmetic as in wide_node_hits): which
# (instance, triangle slot) pairs does a ray reach? Used to tell a builder fault from a traversal fault.
# ---------------------------------------------------------------------------------------------
def _node_hits(n, o, idir, tmin, tmax):
    f = np.float32
    scale = np.ldexp(f(1.0), n["exp"].astype(np.int32) - 127).astype(f)
    a = (scale * idir).astype(f)
    b = ((n["origin"].astype(f) - o) * idir).astype(f)
    qlo = np.stack([n["qlox"], n["qloy"], n["qloz"]], -1).astype(f)
    qhi = np.stack([n["qhix"], n["qhiy"], n["qhiz"]], -1).astype(f)
    neg = idir < 0
    near = np.where(neg, qhi, qlo); far = np.where(neg, qlo, qhi)
    tn = (near * a + b).astype(f); tf = (far * a + b).astype(f)
    tn = np.maximum(tn.max(-1), f(tmin)); tf = np.minimum(tf.min(-1), f(tmax))
    return tn <= tf * f(1.000001)


def reach(layout, buf, o, d, tmin=0.0, tmax=np.inf, leaf_tris_rule=lambda n: 2 if n <= 32 else 1):
    """[(instance, [triangle slots the box traversal hands to the triangle test])] for the ray, with no closest-hit culling."""
    f = np.float32
    inst, nodes, tris, order = split(layout, buf)
    o = np.asarray(o, f); d = np.asarray(d, f)

    def safe_inv(v):
        v = np.where(np.abs(v) < 1e-20, np.copysign(f(1e-20), v), v).astype(f)
        return (f(1.0) / v).astype(f)

    def items(node_base, single, count, ro, rd):
        if single:
            return list(range(count))
        idir = safe_inv(rd)
        out, stack = [], [0]
        while stack:
            n = nodes[node_base + stack.pop()]
            hit = _node_hits(n, ro, idir, tmin, tmax)
            rank = 0
            for s in range(8):
                m = int(n["meta"][s])
                if m == 0:
                    continue
                if (m & 0x1F) >= 24:
                    if hit[s]:
                        stack.append(int(n["childBase"]) + rank)
                    rank += 1
                elif hit[s]:
                    cnt = {1: 1, 3: 2, 7: 3}[m >> 5]
                    first = int(n["triBase"]) + (m & 0x1F)
                    out.extend(range(first, first + cnt))
        return out

    res = []
    for pos in items(0, len(inst) == 1, len(inst), o, d):
        x = int(order[pos]); it = inst[x]
        if not (int(it["mask"]) & 0xFF) or int(it["triCount"]) == 0:
            continue
        W = it["worldToObject"].reshape(3, 4).astype(f)
        ro = (W[:, :3] @ o + W[:, 3]).astype(f); rd = (W[:, :3] @ d).astype(f)
        tc = int(it["triCount"])
        res.append((x, items(int(it["nodeBase"]), tc <= leaf_tris_rule(tc), tc, ro, rd)))
    return res
