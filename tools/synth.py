"""Seeded synthetic inputs for tests and bench: no datasets exist offline.

Images: low-frequency gradient + random-contrast rectangles / rotated rectangles / discs
+ N(0, 2^2) noise, uint8.  Stereo: right eye = left shifted by a per-row-band integer
disparity in [2, 80] px + independent noise.  BoW: a seeded two-level slice of a k=10
vocabulary tree (only tree level L-4=2 matters for DBoW2::FeatureVector with levelsup=4,
reference src/Frame.cc:464, Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1218-1259).
"""
import numpy as np


def _texture(rng, h, w, nshapes):
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = 96.0 + 48.0 * np.sin(xx / w * 2.1 + 0.3) * np.cos(yy / h * 1.7 + 0.9)
    for _ in range(nshapes):
        kind = rng.integers(0, 3)
        cx, cy = rng.uniform(0, w), rng.uniform(0, h)
        sx, sy = rng.uniform(3, 0.03 * w + 6), rng.uniform(3, 0.06 * h + 6)
        val = rng.uniform(10, 245)
        x0, x1 = int(max(cx - 1.5 * sx - 2, 0)), int(min(cx + 1.5 * sx + 2, w))
        y0, y1 = int(max(cy - 1.5 * sy - 2, 0)), int(min(cy + 1.5 * sy + 2, h))
        if x1 <= x0 or y1 <= y0:
            continue
        sub_x, sub_y = xx[y0:y1, x0:x1] - cx, yy[y0:y1, x0:x1] - cy
        if kind == 0:
            m = (np.abs(sub_x) < sx) & (np.abs(sub_y) < sy)
        elif kind == 1:
            t = rng.uniform(0, np.pi)
            c, s = np.cos(t), np.sin(t)
            m = (np.abs(c * sub_x + s * sub_y) < sx) & (np.abs(-s * sub_x + c * sub_y) < sy)
        else:
            r = min(sx, sy)
            m = sub_x * sub_x + sub_y * sub_y < r * r
        blk = img[y0:y1, x0:x1]
        a = rng.uniform(0.5, 1.0)
        blk[m] = (1 - a) * blk[m] + a * val
    return img


def _finish(rng, img, sigma=2.0):
    out = img + rng.normal(0, sigma, img.shape).astype(np.float32)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def image(seed, w, h, nshapes=1500):
    rng = np.random.Generator(np.random.PCG64(seed))
    return _finish(rng, _texture(rng, h, w, nshapes))


def sequence(seed, w, h, nframes, nshapes=1500):
    """frames translating 1-3 px/frame over one wide texture"""
    rng = np.random.Generator(np.random.PCG64(seed))
    pad = 3 * nframes + 4
    tex = _texture(rng, h + pad, w + pad, nshapes)
    out, ox, oy = [], 0, 0
    for _ in range(nframes):
        out.append(_finish(rng, tex[oy:oy + h, ox:ox + w]))
        ox += int(rng.integers(1, 4)); oy += int(rng.integers(0, 3))
    return np.stack(out)


def stereo_pair(seed, w, h, nshapes=1500, dmin=2, dmax=80):
    rng = np.random.Generator(np.random.PCG64(seed))
    tex = _texture(rng, h, w + dmax + 1, nshapes)
    nb = max(2, h // 47)
    edges = np.linspace(0, h, nb + 1).astype(int)
    disp = np.zeros(h, np.int64)
    for b in range(nb):
        disp[edges[b]:edges[b + 1]] = int(rng.integers(dmin, dmax + 1))
    # a scene point seen at column uL in the left eye is seen at uR = uL - d in the right eye
    # (positive disparity, src/Frame.cc:649):  right[y, u] = left[y, u + d]
    left = tex[:, :w]
    right = np.empty_like(left)
    for y in range(h):
        right[y] = tex[y, disp[y]:disp[y] + w]
    return _finish(rng, left), _finish(rng, right), disp


# ----------------------------------------------------------------------------- BoW

_POPCNT = np.array([bin(i).count("1") for i in range(256)], np.uint8)


def hamming_matrix(a, b):
    """[na,32] x [nb,32] uint8 -> [na,nb] int"""
    x = a[:, None, :] ^ b[None, :, :]
    return _POPCNT[x].sum(axis=2).astype(np.int32)


class Vocab2:
    """Levels 1 and 2 of a seeded k=10 vocabulary tree; node ids in breadth-first order
    (root 0, level-1 nodes 1..10, level-2 nodes 11..110)."""

    def __init__(self, seed, k=10):
        rng = np.random.Generator(np.random.PCG64(seed))
        self.k = k
        self.l1 = rng.integers(0, 256, (k, 32), dtype=np.uint8)
        self.l2 = rng.integers(0, 256, (k, k, 32), dtype=np.uint8)

    def seed_from(self, desc, rng):
        """re-seed node descriptors from data so that features spread over nodes"""
        k = self.k
        self.l1 = desc[rng.choice(len(desc), k, replace=False)].copy()
        self.l2 = desc[rng.choice(len(desc), k * k, replace=False)].reshape(k, k, 32).copy()

    def node_of(self, desc):
        c1 = hamming_matrix(desc, self.l1).argmin(axis=1)  # first minimum, strict <
        out = np.empty(len(desc), np.uint32)
        for c in range(self.k):
            m = np.nonzero(c1 == c)[0]
            if len(m):
                c2 = hamming_matrix(desc[m], self.l2[c]).argmin(axis=1)
                out[m] = 1 + self.k + c * self.k + c2
        return out

    def feature_vector(self, desc):
        """CSR FeatureVector: (node_id ascending u32, node_off i32, feat ascending-in-node u32)"""
        node = self.node_of(desc)
        order = np.argsort(node, kind="stable")
        ids, counts = np.unique(node, return_counts=True)
        off = np.zeros(len(ids) + 1, np.int32)
        off[1:] = np.cumsum(counts)
        return ids.astype(np.uint32), off, order.astype(np.uint32)


def flip_bits(rng, desc, p):
    bits = np.unpackbits(desc, axis=1)
    bits ^= (rng.random(bits.shape) < p).astype(np.uint8)
    return np.packbits(bits, axis=1)


def vocab_tree(seed, k=10, L=3, stop_frac=0.02, data=None):
    """Seeded full k-ary vocabulary tree in DBoW2 id order (node ids assigned breadth-first, as
    saveToTextFile / loadFromTextFile keep them): returns parent[], is_leaf[], desc[], weight[] for
    nodes 1..N (the root is implicit).  Leaf weights are positive idf-like values, a fraction is 0
    ("stopped" words, TemplatedVocabulary.h:1157).  Node descriptors are random, or sampled from
    `data` descriptors (+ bit noise) so that real features spread over the tree."""
    rng = np.random.Generator(np.random.PCG64(seed))
    parent, leaf = [], []
    level_nodes = [0]
    next_id = 1
    for lvl in range(1, L + 1):
        new = []
        for p in level_nodes:
            for _ in range(k):
                parent.append(p); leaf.append(1 if lvl == L else 0); new.append(next_id); next_id += 1
        level_nodes = new
    n = len(parent)
    if data is not None:
        desc = flip_bits(rng, data[rng.integers(0, len(data), n)], 0.1)
    else:
        desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    weight = np.where(np.array(leaf) == 1, rng.uniform(0.5, 9.0, n), 0.0)
    weight[(np.array(leaf) == 1) & (rng.random(n) < stop_frac)] = 0.0
    return np.array(parent, np.int32), np.array(leaf, np.uint8), desc, weight


def write_vocab_text(path, k, L, parent, is_leaf, desc, weight, scoring=0, weighting=0):
    """the ORBvoc.txt text format read by TemplatedVocabulary::loadFromTextFile (:1358-1445)"""
    with open(path, "w") as f:
        f.write(f"{k} {L} {scoring} {weighting}\n")
        for i in range(len(parent)):
            f.write(f"{parent[i]} {is_leaf[i]} " + " ".join(str(int(b)) for b in desc[i]) + f" {float(weight[i])!r}\n")
