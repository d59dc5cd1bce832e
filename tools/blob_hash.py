"""Developer helper: digests of the sections of the traversal copy (instances, nodes, triangles, leaf instances) of some workloads, for
comparing two builds of the library (PROF_LIB=<name> selects build/ab/libptamd_<name>.so): a change of the builder that is meant to keep
the trees must keep these digests."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.load_package()
import dxpbrt_amd.ptamd as P, dxpbrt_amd.scenes as S
if os.environ.get("PROF_LIB"):
    P.LIB_PATH = os.path.join(ROOT, "build", "ab", "libptamd_%s.so" % os.environ["PROF_LIB"])
import bench
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bvh_check as B
import numpy as np


def canonical(nodes, base, items, single):
    """digest of one tree that does not depend on how its nodes and items are numbered: depth-first in slot order, per node its
    frame and per slot the quantised box, kind and item identities"""
    h = hashlib.sha256()
    if single:
        h.update(np.ascontiguousarray(items).tobytes()); return h.hexdigest()[:12]
    stack = [0]
    while stack:
        n = nodes[base + stack.pop()]
        h.update(n["origin"].tobytes()); h.update(n["exp"].tobytes()); h.update(bytes([int(n["imask"])]))
        rank = 0; kids = []
        for s in range(8):
            m = int(n["meta"][s])
            h.update(bytes([int(n[k][s]) for k in ("qlox", "qloy", "qloz", "qhix", "qhiy", "qhiz")]))
            if m == 0:
                h.update(b"e")
            elif (m & 0x1F) >= 24:
                h.update(b"i"); kids.append(int(n["childBase"]) + rank); rank += 1
            else:
                cnt = {1: 1, 3: 2, 7: 3}[m >> 5]; first = int(n["triBase"]) + (m & 0x1F)
                h.update(b"l"); h.update(np.ascontiguousarray(items[first:first + cnt]).tobytes())
        stack.extend(reversed(kids))
    return h.hexdigest()[:12]
for w in sys.argv[1:] or ["c2", "c3", "c5", "dynamic"]:
    kind, W, H, spp, bounces, desc = bench.WORKLOADS[w]
    scene, ext = bench.make_scene(kind, W / H, S)
    ctx = P.DeviceContext(0); g = P.Scene(ctx, scene); ctx.sync()
    lay, buf = ctx.download_blob()
    cuts = [lay.InstanceOffset16 * 16, lay.NodeOffset16 * 16, lay.TriangleOffset16 * 16, lay.LeafInstanceOffset16 * 16, lay.Bytes]
    order = sorted(range(4), key=lambda i: cuts[i])
    names = ["instances", "nodes", "triangles", "leaf_instances"]
    out = []
    for j, i in enumerate(order):
        end = cuts[order[j + 1]] if j + 1 < 4 else lay.Bytes
        out.append("%s %s" % (names[i], hashlib.sha256(buf[cuts[i]:end].tobytes()).hexdigest()[:12]))
    inst, nodes, tris, order = B.split(lay, buf)
    canon = ["tlas " + canonical(nodes, 0, order, len(inst) == 1)]
    done = set()
    for it in inst:
        key = (int(it["nodeBase"]), int(it["triBase"]), int(it["triCount"]))
        if key in done or not key[2] or len(done) >= 4: continue
        done.add(key)
        t = tris[key[1]:key[1] + key[2]]
        canon.append("blas@%d %s" % (key[0], canonical(nodes, key[0], np.stack([t["geom"], t["prim"]], -1), key[2] <= (2 if key[2] <= 32 else 1))))
    out = canon + out
    st = ctx.accel_stats()
    print(w, "nodes", lay.NodeCount, "tris", lay.TriangleCount, "depth", st.MaxBottomLevelDepth, st.TopLevelDepth, "|", " | ".join(out), flush=True)
    ctx.close()
