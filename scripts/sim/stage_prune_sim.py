"""CPU simulation (round 4): how many (wave, stage) pairs of the pre-filter's sweep could be skipped by a bounding-box test in a
low-dimensional space -- BASELINE C5: MSN(n_components=8), 100k references, k = 1 (J = 2).  References ordered by the cells of a
median-split tree of depth `depth` over the principal axes (stage = 512 rows), queries bucketed by the same tree, a wave = 64
consecutive queries of the bucketed order; a wave skips a stage when EVERY query of it has lb(q, box of the stage) >= its running
threshold (J-th smallest squared distance seen so far, seeded from the stages around its own cell).
usage: python scripts/sim/stage_prune_sim.py [depth] [n_waves]"""
import sys, numpy as np
sys.path.insert(0, ".")
from sknnr_amd import synth, transformers as T

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n_waves = int(sys.argv[2]) if len(sys.argv) > 2 else 48
qdepth = int(sys.argv[3]) if len(sys.argv) > 3 else depth  # the queries are bucketed by a deeper level of the same tree
n_ref, d_in, J, ROWS = 100_000, 32, 2, 512
x_ref = synth.make_features(n_ref, d_in, seed=0)
y = synth.make_targets(x_ref, t=40, kind="linear")
tr = T.CCorATransformer(8).fit(x_ref, y)
ref = tr.transform(x_ref)
rng = np.random.default_rng(1)
nq_all = int(sys.argv[4]) if len(sys.argv) > 4 else 200_000
q_all = tr.transform(rng.standard_normal((nq_all, d_in)) @ synth.mixing_matrix(d_in))
d = ref.shape[1]
mu = ref.mean(0)
val, vec = np.linalg.eigh(np.cov((ref - mu).T))
axes = vec[:, ::-1]

def build(z, depth):
    code = np.zeros(len(z), int); thr = {}
    for l in range(depth):
        new = np.zeros_like(code)
        for n in range(1 << l):
            m = code == n
            s = np.median(z[m, l % d]) if m.any() else 0.0
            thr[(l, n)] = s
            new[m] = 2 * n + (z[m, l % d] >= s)
        code = new
    return code, thr
zr, zq = (ref - mu) @ axes, (q_all - mu) @ axes
code_deep, thr = build(zr, max(depth, qdepth))
code = code_deep >> (max(depth, qdepth) - depth)
def assign(z, depth=max(depth, qdepth)):
    node = np.zeros(len(z), int)
    for l in range(depth):
        s = np.array([thr[(l, n)] for n in node])
        node = 2 * node + (z[:, l % d] >= s)
    return node
qdeep = assign(zq)
qcode = qdeep >> (max(depth, qdepth) - depth)
h = (np.arange(n_ref, dtype=np.uint64) * 0x9E3779B1) % 1000003
perm = np.lexsort((h, code))
refp = ref[perm]
n_stage = (n_ref + ROWS - 1) // ROWS
lo = np.stack([refp[s * ROWS:(s + 1) * ROWS].min(0) for s in range(n_stage)])
hi = np.stack([refp[s * ROWS:(s + 1) * ROWS].max(0) for s in range(n_stage)])
first = np.searchsorted(code[perm], np.arange((1 << depth) + 1))
mid_stage = np.minimum(((first[:-1] + first[1:]) // 2) // ROWS, n_stage - 1)
qorder = np.argsort(qdeep, kind="stable")
tot = skipped = q_needed = 0
for w in rng.choice(nq_all // 64, n_waves, replace=False):
    rows = qorder[w * 64:(w + 1) * 64]
    q = q_all[rows]
    st0 = (mid_stage[qcode[rows[32]]] - 2) % n_stage
    order = (st0 + np.arange(n_stage)) % n_stage
    D = ((q[:, None, :] - refp[None, :, :]) ** 2).sum(-1).reshape(64, n_stage, -1) if False else None
    best = np.full((64, J), np.inf)
    # seed: 4 stages around the cell
    for s in order[:4]:
        dd = ((q[:, None, :] - refp[s * ROWS:(s + 1) * ROWS][None]) ** 2).sum(-1)
        best = np.sort(np.concatenate([best, dd], 1), 1)[:, :J]
    thr_q = best[:, J - 1].copy(); best[:] = np.inf
    for s in order:
        gap = np.maximum(0, np.maximum(lo[s] - q, q - hi[s]))
        lb = (gap ** 2).sum(1)
        need = lb < thr_q
        tot += 1; q_needed += need.mean()
        if not need.any():
            skipped += 1
            continue
        dd = ((q[:, None, :] - refp[s * ROWS:(s + 1) * ROWS][None]) ** 2).sum(-1)
        best = np.sort(np.concatenate([best, dd], 1), 1)[:, :J]
        thr_q = np.minimum(thr_q, best[:, J - 1])
print(f"query depth {qdepth}; depth {depth}: {1 << depth} cells, {n_stage} stages of {ROWS} rows; waves sampled {n_waves}: stages skipped per WAVE {100 * skipped / tot:.1f} %, "
      f"needed per QUERY {100 * q_needed / tot:.1f} %")
