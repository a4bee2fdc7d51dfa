"""Data-parallel reformulation of DistributeOctTree -- numpy MODEL of the HIP
kernel in rgbd_pl_slam_amd/csrc/orb_octree.hip.  Every step is an array operation that maps to a
workgroup-parallel primitive (per-key map, per-node map, LDS atomic histogram, block scan, sort);
test_models.py checks it order-exactly against the sequential oracle, and the HIP kernel
mirrors it line by line.  Test/model code only -- not imported by the product."""
import numpy as np


def _divide(nodes, sel, kx, ky, node_of_key):
    """Split every node in `sel` (indices into the node arrays).  Returns child rects [len(sel),4,4]
    (x0,y0,x1,y1), quadrant per key (-1 for keys not in a split node) and child key counts."""
    x0, y0, x1, y1 = nodes["x0"][sel], nodes["y0"][sel], nodes["x1"][sel], nodes["y1"][sel]
    halfX = np.ceil((x1 - x0).astype(np.float32) / np.float32(2)).astype(np.int32)
    halfY = np.ceil((y1 - y0).astype(np.float32) / np.float32(2)).astype(np.int32)
    mx, my = x0 + halfX, y0 + halfY
    rects = np.zeros((len(sel), 4, 4), np.int32)
    rects[:, 0] = np.stack([x0, y0, mx, my], 1)
    rects[:, 1] = np.stack([mx, y0, x1, my], 1)
    rects[:, 2] = np.stack([x0, my, mx, y1], 1)
    rects[:, 3] = np.stack([mx, my, x1, y1], 1)
    slot = np.full(len(nodes["x0"]), -1, np.int64)
    slot[sel] = np.arange(len(sel))
    ks = slot[node_of_key]                      # per key: which split slot (or -1)
    quad = np.full(len(kx), -1, np.int64)
    a = ks >= 0
    left = kx[a] < mx[ks[a]].astype(np.float32)
    top = ky[a] < my[ks[a]].astype(np.float32)
    quad[a] = np.where(left, np.where(top, 0, 2), np.where(top, 1, 3))
    cnt = np.zeros((len(sel), 4), np.int64)
    np.add.at(cnt, (ks[a], quad[a]), 1)        # LDS atomic histogram on the GPU
    return rects, ks, quad, cnt


def distribute_octree(kx, ky, resp, minX, maxX, minY, maxY, N):
    """kx, ky, resp: float32 arrays (coordinates relative to minX/minY).  Returns indices of the kept keys."""
    nk = len(kx)
    if nk == 0:
        return np.zeros(0, np.int64)
    nIni = int(np.round(np.float32(maxX - minX) / np.float32(maxY - minY)))  # roundf (half away from zero)
    v = np.float32(maxX - minX) / np.float32(maxY - minY)
    nIni = int(np.floor(v + np.float32(0.5))) if v >= 0 else int(np.ceil(v - np.float32(0.5)))
    if nIni < 1:
        return np.zeros(0, np.int64)
    hX = np.float32(maxX - minX) / np.float32(nIni)
    # roots, list order = index order
    ii = np.arange(nIni)
    nodes = dict(x0=(hX * ii.astype(np.float32)).astype(np.int32), y0=np.zeros(nIni, np.int32),
                 x1=(hX * (ii + 1).astype(np.float32)).astype(np.int32), y1=np.full(nIni, maxY - minY, np.int32))
    node_of_key = np.clip((kx / hX).astype(np.int64), 0, nIni - 1)
    cnt = np.bincount(node_of_key, minlength=nIni)
    keep = cnt > 0                               # erase empty roots (order preserved)
    remap = np.cumsum(keep) - 1
    for k in nodes:
        nodes[k] = nodes[k][keep]
    nodes["n"] = cnt[keep].astype(np.int64)
    node_of_key = remap[node_of_key]
    # candidates for phase 2: (node index, creation rank)
    finish = False
    cand = np.zeros(0, np.int64)
    while not finish:
        L = len(nodes["n"])
        prev = L
        split = np.nonzero(nodes["n"] > 1)[0]    # noMore == (n == 1); processed in list order
        rects, ks, quad, ccnt = _divide(nodes, split, kx, ky, node_of_key)
        nonempty = ccnt > 0
        c_per = nonempty.sum(1)                  # children per split node
        # new list: children of the LAST processed node first; inside a node n4..n1; then old noMore nodes
        S = len(split)
        after = np.concatenate((np.cumsum(c_per[::-1])[::-1][1:], [0])) if S else np.zeros(0, np.int64)
        # rank of child q among the node's non-empty children in reverse order (n4 first)
        rev_rank = np.cumsum(nonempty[:, ::-1], 1)[:, ::-1] - 1
        child_pos = after[:, None] + rev_rank
        total_children = int(c_per.sum())
        stay = np.nonzero(nodes["n"] == 1)[0]
        new = {k: np.zeros(total_children + len(stay), np.int32) for k in ("x0", "y0", "x1", "y1")}
        new["n"] = np.zeros(total_children + len(stay), np.int64)
        si, qi = np.nonzero(nonempty)
        pos = child_pos[si, qi]
        for j, k in enumerate(("x0", "y0", "x1", "y1")):
            new[k][pos] = rects[si, qi, j]
            new[k][total_children:] = nodes[k][stay]
        new["n"][pos] = ccnt[si, qi]
        new["n"][total_children:] = 1
        # keys follow their child / their (moved) noMore node
        stay_pos = np.full(L, -1, np.int64)
        stay_pos[stay] = total_children + np.arange(len(stay))
        a = ks >= 0
        nk_new = np.where(a, 0, stay_pos[node_of_key])
        nk_new[a] = child_pos[ks[a], quad[a]]
        node_of_key = nk_new
        # candidates recorded in creation order: (processing order of parent, q = n1..n4), only n > 1
        big = ccnt > 1
        bi, bq = np.nonzero(big)                 # row-major == creation order
        cand = child_pos[bi, bq]
        nToExpand = len(cand)
        nodes = new
        L = len(nodes["n"])
        if L >= N or L == prev:
            finish = True
        elif L + 3 * nToExpand > N:
            while not finish:
                prev = L
                # sort ascending by (n, creation rank); iterate from the back
                order = np.lexsort((np.arange(len(cand)), nodes["n"][cand]))
                work = cand[order][::-1]         # processing order
                rects, ks, quad, ccnt = _divide(nodes, work, kx, ky, node_of_key)
                nonempty = ccnt > 0
                c_per = nonempty.sum(1)
                sizes_after = prev + np.cumsum(c_per - 1)
                hit = np.nonzero(sizes_after >= N)[0]
                P = (hit[0] + 1) if len(hit) else len(work)   # number of nodes really divided
                work_p = work[:P]
                nonempty_p, ccnt_p, rects_p = nonempty[:P], ccnt[:P], rects[:P]
                c_p = c_per[:P]
                after = np.concatenate((np.cumsum(c_p[::-1])[::-1][1:], [0])) if P else np.zeros(0, np.int64)
                rev_rank = np.cumsum(nonempty_p[:, ::-1], 1)[:, ::-1] - 1
                child_pos = after[:, None] + rev_rank
                total_children = int(c_p.sum())
                removed = np.zeros(L, bool)
                removed[work_p] = True
                stay = np.nonzero(~removed)[0]
                new = {k: np.zeros(total_children + len(stay), np.int32) for k in ("x0", "y0", "x1", "y1")}
                new["n"] = np.zeros(total_children + len(stay), np.int64)
                si, qi = np.nonzero(nonempty_p)
                pos = child_pos[si, qi]
                for j, k in enumerate(("x0", "y0", "x1", "y1")):
                    new[k][pos] = rects_p[si, qi, j]
                    new[k][total_children:] = nodes[k][stay]
                new["n"][pos] = ccnt_p[si, qi]
                new["n"][total_children:] = nodes["n"][stay]
                stay_pos = np.full(L, -1, np.int64)
                stay_pos[stay] = total_children + np.arange(len(stay))
                a = (ks >= 0) & (ks < P)
                nk_new = np.where(a, 0, stay_pos[node_of_key])
                nk_new[a] = child_pos[ks[a], quad[a]]
                node_of_key = nk_new
                big = ccnt_p > 1
                bi, bq = np.nonzero(big)
                cand = child_pos[bi, bq]
                nodes = new
                L = len(nodes["n"])
                if L >= N or L == prev:
                    finish = True
    # best key per node: max response, first in key order wins ties
    L = len(nodes["n"])
    out = np.full(L, -1, np.int64)
    best = np.full(L, -np.inf)
    for i in range(nk):                          # GPU: atomicMax on (response << 32 | ~index) per node
        n = node_of_key[i]
        if resp[i] > best[n]:
            best[n] = resp[i]
            out[n] = i
    return out
