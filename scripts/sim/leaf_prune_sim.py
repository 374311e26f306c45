"""CPU simulation (round 4): tile-level pruning for low-dimensional spaces -- BASELINE C5: MSN(n_components=8), 100k references,
k = 1 (J = 2).  References ordered by a kd-tree (median split on the widest axis) down to leaves of 32 rows = one tile of the
pre-filter's image; a tile is summarised by its centroid and radius (ball) or its bounding box.  Queries are sorted by the
leaf they fall in; a wave = 64 consecutive queries.  A wave must sweep a tile when ANY of its queries has
lb(q, tile) < its threshold.  Thresholds: 'final' = the query's J-th smallest squared distance (what the sweep converges
to), 'seed' = the J-th smallest over the 2,048 rows around the query's own leaf (what the seeding pass would give).
usage: python scripts/sim/leaf_prune_sim.py [n_waves] [nq_all] [dims]"""
import sys, numpy as np
sys.path.insert(0, ".")
from sknnr_amd import synth, transformers as T

n_waves = int(sys.argv[1]) if len(sys.argv) > 1 else 200
nq_all = int(sys.argv[2]) if len(sys.argv) > 2 else 400_000
ncomp = int(sys.argv[3]) if len(sys.argv) > 3 else 8
n_ref, d_in, J, LEAF = 100_000, 32, 2, 32
x_ref = synth.make_features(n_ref, d_in, seed=0)
y = synth.make_targets(x_ref, t=40, kind="linear")
tr = T.CCorATransformer(ncomp).fit(x_ref, y)
ref = tr.transform(x_ref)
rng = np.random.default_rng(1)
q_all = tr.transform(rng.standard_normal((nq_all, d_in)) @ synth.mixing_matrix(d_in))
d = ref.shape[1]

# kd order: recursive median split on the widest axis, sizes kept multiples of LEAF
order = np.arange(n_ref)
splits = []  # (lo, hi, axis, value) for assigning queries
def build(lo, hi, node):
    n = hi - lo
    if n <= LEAF:
        return {"leaf": lo // LEAF}
    rows = order[lo:hi]
    ax = int(np.argmax(ref[rows].max(0) - ref[rows].min(0)))
    half = ((n // LEAF + 1) // 2) * LEAF
    part = np.argpartition(ref[rows, ax], half - 1)
    order[lo:hi] = rows[part]
    val = ref[order[lo + half - 1], ax]
    return {"ax": ax, "val": val, "l": build(lo, lo + half, 2 * node), "r": build(lo + half, hi, 2 * node + 1)}
sys.setrecursionlimit(10000)
tree = build(0, n_ref, 1)
refp = ref[order]
n_tiles = (n_ref + LEAF - 1) // LEAF
cent = np.stack([refp[t * LEAF:(t + 1) * LEAF].mean(0) for t in range(n_tiles)])
rad = np.array([np.sqrt(((refp[t * LEAF:(t + 1) * LEAF] - cent[t]) ** 2).sum(1).max()) for t in range(n_tiles)])
blo = np.stack([refp[t * LEAF:(t + 1) * LEAF].min(0) for t in range(n_tiles)])
bhi = np.stack([refp[t * LEAF:(t + 1) * LEAF].max(0) for t in range(n_tiles)])

def leaf_of(q):
    out = np.empty(len(q), int)
    stack = [(tree, np.arange(len(q)))]
    while stack:
        nd, idx = stack.pop()
        if "leaf" in nd:
            out[idx] = nd["leaf"]; continue
        m = q[idx, nd["ax"]] <= nd["val"]
        stack.append((nd["l"], idx[m])); stack.append((nd["r"], idx[~m]))
    return out
qleaf = leaf_of(q_all)
qorder = np.argsort(qleaf, kind="stable")
res = {k: [] for k in ("ball_final", "box_final", "ball_seed", "box_seed", "q_ball_final", "q_ball_seed", "q_box_final", "wavebox_final", "wavebox_seed",
                         "halfwavebox_final")}
for w in rng.choice(nq_all // 64, n_waves, replace=False):
    rows = qorder[w * 64:(w + 1) * 64]
    q = q_all[rows]
    d2 = ((q[:, None, :] - refp[None, :, :]) ** 2).sum(-1)
    thr_final = np.sort(d2, 1)[:, J - 1]
    t0 = int(np.clip(qleaf[rows[32]] - 32, 0, n_tiles - 64))
    thr_seed = np.sort(d2[:, t0 * LEAF:(t0 + 64) * LEAF], 1)[:, J - 1]
    dc = np.sqrt(((q[:, None, :] - cent[None]) ** 2).sum(-1))
    lb_ball = np.maximum(dc - rad[None], 0) ** 2
    gap = np.maximum(0, np.maximum(blo[None] - q[:, None], q[:, None] - bhi[None]))
    lb_box = (gap ** 2).sum(-1)
    for name, thr in (("final", thr_final), ("seed", thr_seed)):
        res["ball_" + name].append((lb_ball < thr[:, None] * 1.0001).any(0).mean())
        res["box_" + name].append((lb_box < thr[:, None] * 1.0001).any(0).mean())
        res["q_ball_" + name].append((lb_ball < thr[:, None] * 1.0001).mean())
        # the wave's own bounding box against the tile's box, one threshold per wave (the largest)
        wlo, whi = q.min(0), q.max(0)
        g2 = np.maximum(0, np.maximum(blo - whi[None], wlo[None] - bhi))
        res["wavebox_" + name].append(((g2 ** 2).sum(-1) < thr.max() * 1.0001).mean())
    res["q_box_final"].append((lb_box < thr_final[:, None] * 1.0001).mean())
    hw = []
    for h in range(2):
        qq, tt = q[h * 32:(h + 1) * 32], thr_final[h * 32:(h + 1) * 32]
        g2 = np.maximum(0, np.maximum(blo - qq.max(0)[None], qq.min(0)[None] - bhi))
        hw.append((g2 ** 2).sum(-1) < tt.max() * 1.0001)
    res["halfwavebox_final"].append((hw[0] | hw[1]).mean())
print(f"{ncomp}-D, {n_ref} refs in {n_tiles} kd leaves of {LEAF}; {nq_all} queries sorted by leaf, {n_waves} waves of 64 sampled: tiles a WAVE must sweep -- "
      + ", ".join(f"{k} {100 * np.mean(v):.1f} %" for k, v in res.items()))

# optimistic limit: a wave of the 64 MUTUALLY NEAREST queries (what perfect query sorting would give)
from scipy.spatial import cKDTree
kt = cKDTree(q_all)
lim_box, lim_ball = [], []
for c in rng.choice(nq_all, 40, replace=False):
    _, nb = kt.query(q_all[c], 64)
    q = q_all[nb]
    d2 = ((q[:, None, :] - refp[None, :, :]) ** 2).sum(-1)
    thr = np.sort(d2, 1)[:, J - 1]
    gap = np.maximum(0, np.maximum(blo[None] - q[:, None], q[:, None] - bhi[None]))
    lim_box.append((((gap ** 2).sum(-1)) < thr[:, None] * 1.0001).any(0).mean())
    dc = np.sqrt(((q[:, None, :] - cent[None]) ** 2).sum(-1))
    lim_ball.append((np.maximum(dc - rad[None], 0) ** 2 < thr[:, None] * 1.0001).any(0).mean())
print(f"waves of the 64 mutually nearest of {nq_all} queries: box_final {100 * np.mean(lim_box):.1f} %, ball_final {100 * np.mean(lim_ball):.1f} %")
