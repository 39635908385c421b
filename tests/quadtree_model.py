"""numpy model of the data-parallel quadtree-cull formulation used by the HIP kernel
(orb-slam2_amd/csrc/orbx_extract.hip: k_quadtree).  It exists to check, on CPU and against the
sequential oracle (oracle_distribute_octtree, which follows src/ORBextractor.cc:617-915 with a
linked list), the order algebra the kernel relies on:

  * node id == position in the list;
  * a sweep splits a set of nodes in a processing order; children of the r-th processed node are
    pushed to the list front in order n1..n4, so new list = reverse(all children in processing
    order) ++ (unsplit nodes in old order);
  * phase 1 processes every node with >1 point in list order; phase 2 processes nodes with >1 point
    sorted by (count desc, list position asc) and stops after the split that reaches N leaves;
  * per leaf the kept point is argmax(response), lowest original index on ties.
"""
import numpy as np


def quadtree_model(px, py, resp, min_x, max_x, min_y, max_y, N):
    px = np.asarray(px, np.int64); py = np.asarray(py, np.int64); resp = np.asarray(resp, np.int64)
    n = len(px)
    if n == 0:
        return np.zeros(0, np.int64)
    width, height = max_x - min_x, max_y - min_y
    n_ini = int(np.floor(np.float32(width) / np.float32(height) + np.float32(0.5)))  # roundf for positives
    hx = np.float32(width) / np.float32(n_ini)
    root = (px.astype(np.float32) / hx).astype(np.int64)
    root = np.clip(root, 0, n_ini - 1)
    rx0 = (hx * np.arange(n_ini, dtype=np.float32)).astype(np.int64)
    rx1 = (hx * np.arange(1, n_ini + 1, dtype=np.float32)).astype(np.int64)
    cnt_root = np.bincount(root, minlength=n_ini)
    alive = np.nonzero(cnt_root > 0)[0]
    remap = -np.ones(n_ini, np.int64); remap[alive] = np.arange(len(alive))
    nid = remap[root]
    x0, x1 = rx0[alive], rx1[alive]
    y0, y1 = np.zeros(len(alive), np.int64), np.full(len(alive), height, np.int64)
    cnt = cnt_root[alive]

    def apply_split(split_order):
        """split_order: node ids to split, in processing order"""
        nonlocal nid, x0, x1, y0, y1, cnt
        m = len(cnt)
        hxn = (x1 - x0 + 1) // 2
        hyn = (y1 - y0 + 1) // 2
        c = (px >= (x0 + hxn)[nid]).astype(np.int64) + 2 * (py >= (y0 + hyn)[nid]).astype(np.int64)
        cc = np.zeros((m, 4), np.int64)
        np.add.at(cc, (nid, c), 1)
        is_split = np.zeros(m, bool); is_split[split_order] = True
        nc = (cc[split_order] > 0).sum(axis=1)
        S = int(nc.sum())
        s_off = np.concatenate([[0], np.cumsum(nc)[:-1]]) if len(nc) else np.zeros(0, np.int64)
        unsplit = np.nonzero(~is_split)[0]
        m2 = S + len(unsplit)
        nx0 = np.zeros(m2, np.int64); nx1 = nx0.copy(); ny0 = nx0.copy(); ny1 = nx0.copy(); ncnt = nx0.copy()
        child_pos = -np.ones((m, 4), np.int64)
        for r, k in enumerate(split_order):
            j = 0
            for ch in range(4):
                if cc[k, ch] == 0:
                    continue
                pos = S - 1 - (s_off[r] + j); j += 1
                child_pos[k, ch] = pos
                nx0[pos] = x0[k] + hxn[k] if ch & 1 else x0[k]
                nx1[pos] = x1[k] if ch & 1 else x0[k] + hxn[k]
                ny0[pos] = y0[k] + hyn[k] if ch & 2 else y0[k]
                ny1[pos] = y1[k] if ch & 2 else y0[k] + hyn[k]
                ncnt[pos] = cc[k, ch]
        upos = S + np.arange(len(unsplit))
        nx0[upos], nx1[upos], ny0[upos], ny1[upos], ncnt[upos] = x0[unsplit], x1[unsplit], y0[unsplit], y1[unsplit], cnt[unsplit]
        keep_pos = -np.ones(m, np.int64); keep_pos[unsplit] = upos
        nid = np.where(is_split[nid], child_pos[nid, c], keep_pos[nid])
        x0, x1, y0, y1, cnt = nx0, nx1, ny0, ny1, ncnt
        return cc

    def child_gain(k_ids):
        hxn = (x1 - x0 + 1) // 2
        hyn = (y1 - y0 + 1) // 2
        c = (px >= (x0 + hxn)[nid]).astype(np.int64) + 2 * (py >= (y0 + hyn)[nid]).astype(np.int64)
        cc = np.zeros((len(cnt), 4), np.int64)
        np.add.at(cc, (nid, c), 1)
        return (cc[k_ids] > 0).sum(axis=1) - 1

    finish = False
    while not finish:
        prev = len(cnt)
        order = np.nonzero(cnt > 1)[0]
        apply_split(order)
        n_to_expand = int((cnt > 1).sum())  # every node with >1 point is a child created in this sweep
        size = len(cnt)
        if size >= N or size == prev:
            finish = True
        elif size + 3 * n_to_expand > N:
            while not finish:
                prev = len(cnt)
                cand = np.nonzero(cnt > 1)[0]
                cand = cand[np.lexsort((cand, -cnt[cand]))]  # count desc, position asc
                gain = child_gain(cand)
                G = np.cumsum(gain)
                reach = np.nonzero(prev + G >= N)[0]
                ksplit = (reach[0] + 1) if len(reach) else len(cand)
                apply_split(cand[:ksplit])
                size = len(cnt)
                if size >= N or size == prev:
                    finish = True
    # leaf winners
    key = resp * (1 << 24) + ((1 << 24) - 1 - np.arange(n))
    best = np.zeros(len(cnt), np.int64)
    np.maximum.at(best, nid, key)
    return (1 << 24) - 1 - (best & ((1 << 24) - 1))
