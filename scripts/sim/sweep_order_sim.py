"""CPU simulation of the pre-filter's hit count per query under different STAGE ORDERS of the sweep
(round 4): the bench's GNN space (50k x 32), cell tree of depth 6 like build_cell_tree, image ordered by
cell, a stage = 16 tiles = 512 rows.  For a sample of queries: hits = values below the running threshold
(J-th smallest so far, J = 6, refreshed per tile; seeded from the first 64 tiles of the query's order)."""
import sys, numpy as np
sys.path.insert(0, ".")
from sknnr_amd import synth, transformers as T

n_ref, d, nq, J, TPS = 50_000, 32, 1024, 6, 16
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 6
x_ref = synth.make_features(n_ref, d, seed=0)
y = synth.make_targets(x_ref, t=40, kind="positive")
tr = T.CCATransformer().fit(x_ref, y)
ref = tr.transform(x_ref)
rng = np.random.default_rng(1)
q = tr.transform((rng.standard_normal((nq, d)) @ synth.mixing_matrix(d)))
mu = ref.mean(0)
c = ref - mu
val, vec = np.linalg.eigh(np.cov(c.T))
axes = vec[:, ::-1][:, :depth]
zr, zq = c @ axes, (q - mu) @ axes

def build(zr):
    code = np.zeros(len(zr), int); thr = {}
    for l in range(depth):
        new = np.zeros_like(code)
        for n in range(1 << l):
            m = code == n
            s = np.median(zr[m, l]) if m.any() else 0.0
            thr[(l, n)] = s
            new[m] = 2 * n + (zr[m, l] >= s)
        code = new
    return code, thr
code, thr = build(zr)
def assign(z):
    node = np.zeros(len(z), int)
    for l in range(depth):
        s = np.array([thr[(l, n)] for n in node])
        node = 2 * node + (z[:, l] >= s)
    return node
qcode = assign(zq)
h = (np.arange(n_ref, dtype=np.uint64) * 0x9E3779B1) % 1000003
perm = np.lexsort((h, code))
n_stage = (n_ref + 32 * TPS - 1) // (32 * TPS)
pad = n_stage * 32 * TPS
D = ((q ** 2).sum(1)[:, None] + (ref ** 2).sum(1)[None, :] - 2 * q @ ref.T)[:, perm]
D = np.concatenate([D, np.full((nq, pad - n_ref), np.inf)], 1).reshape(nq, n_stage, TPS * 32)
first = np.searchsorted(code[perm], np.arange((1 << depth) + 1))
mid_stage = ((first[:-1] + first[1:]) // 2) // (32 * TPS)
# stage centroids in the cell-axes space and in the full space
pos_stage = np.minimum(np.arange(n_ref) // (32 * TPS), n_stage - 1)
cent_full = np.stack([ref[perm][pos_stage == s].mean(0) for s in range(n_stage)])
cell_cent = np.stack([ref[code == cc].mean(0) if (code == cc).any() else mu for cc in range(1 << depth)])

def hits(order_fn, seed_stages=4, per_query=False):
    tot = 0; vis_tiles = 0
    for i in range(nq):
        order = order_fn(i)
        seq = D[i, order].reshape(-1)           # values in sweep order
        seedv = np.sort(seq[: seed_stages * TPS * 32])[J - 1]
        # running J-th smallest refreshed per tile
        tiles = seq.reshape(-1, 32)
        best = np.full(J, np.inf); thr_now = seedv; hcount = 0
        for t in tiles:
            m = t < thr_now
            k = int(m.sum())
            if k:
                hcount += k; vis_tiles += 1
                best = np.sort(np.concatenate([best, t[m]]))[:J]
                thr_now = min(thr_now, best[J - 1])
        tot += hcount
    return tot / nq, vis_tiles / nq

st0 = np.array([(mid_stage[cq] - 2) % n_stage for cq in qcode])
rot = lambda i: (st0[i] + np.arange(n_stage)) % n_stage
def alt(i):
    k = np.arange(n_stage); off = np.where(k % 2 == 1, (k + 1) // 2, -(k // 2))
    return (mid_stage[qcode[i]] + off) % n_stage
def by_cent_cell(i):   # what a workgroup could do: stages by distance from ITS cell's centroid (table per cell)
    dd = ((cent_full - cell_cent[qcode[i]]) ** 2).sum(1)
    return np.argsort(dd, kind="stable")
def by_cent_query(i):  # per query (not realisable per workgroup; upper bound of the centroid rule)
    dd = ((cent_full - q[i]) ** 2).sum(1)
    return np.argsort(dd, kind="stable")
def by_min(i):         # oracle: stages by their smallest value for this query
    return np.argsort(D[i].min(1), kind="stable")
rnd = lambda i: np.random.default_rng(i).permutation(n_stage)
for name, fn in [("rotation (shipped)", rot), ("alternating outward", alt), ("by centroid of own cell", by_cent_cell),
                 ("by centroid, per query", by_cent_query), ("oracle stage order", by_min), ("random stage order", rnd)]:
    hq, vt = hits(fn)
    print(f"depth {depth}  {name:28s} hits/query {hq:6.2f}   visited (tile, query) pairs {vt:6.2f} of {n_stage * TPS}")
