"""Offline estimate of the collision critic's candidate-test volume per design."""
import sys, numpy as np
from dddmr_navigation_amd import scenes
import oracle

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
cell = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
sc = scenes.bench_scene(cfg)
smp = oracle.samples(sc.theory, sc.tick)
if len(smp) > 4096:
    sel = np.random.default_rng(0).choice(len(smp), 4096, replace=False); smp = smp[np.sort(sel)]
pts = sc.cloud[:, :3]
pts = pts[(pts[:, 2] >= -0.02) & (pts[:, 2] <= 0.62)]
print("z-filtered points", len(pts))
# 2D grid (x,y) since z fastest & all z in run
x0, y0 = pts[:, 0].min(), pts[:, 1].min()
ix = np.floor((pts[:, 0] - x0) / cell).astype(int); iy = np.floor((pts[:, 1] - y0) / cell).astype(int)
nx, ny = ix.max() + 1, iy.max() + 1
cnt = np.zeros((ny, nx), dtype=np.int64); np.add.at(cnt, (iy, ix), 1)
# prefix sum along x for row-run counts
cs = np.concatenate([np.zeros((ny, 1), np.int64), np.cumsum(cnt, axis=1)], axis=1)
tot_cand = 0; zero_pairs = 0; pairs = 0; per_pair = []
inside_tot = 0
for s in smp:
    poses, cub, mm = oracle.generate(sc.theory, sc.tick, s)
    for k in range(len(poses)):
        lo = np.maximum(mm[k, 0, :2], poses[k, :2] - 1.0); hi = np.minimum(mm[k, 1, :2], poses[k, :2] + 1.0)
        cx0 = int(np.floor((lo[0] - x0) / cell)); cx1 = int(np.floor((hi[0] - x0) / cell))
        cy0 = int(np.floor((lo[1] - y0) / cell)); cy1 = int(np.floor((hi[1] - y0) / cell))
        cx0 = max(cx0, 0); cy0 = max(cy0, 0); cx1 = min(cx1, nx - 1); cy1 = min(cy1, ny - 1)
        c = 0
        if cx0 <= cx1 and cy0 <= cy1:
            c = int((cs[cy0:cy1 + 1, cx1 + 1] - cs[cy0:cy1 + 1, cx0]).sum())
        per_pair.append(c); pairs += 1
per_pair = np.array(per_pair)
print(cfg, "cell", cell, "pairs", pairs, "total candidates", per_pair.sum(), "mean", per_pair.mean(), "zero frac", (per_pair == 0).mean(),
      "p50", np.percentile(per_pair, 50), "p90", np.percentile(per_pair, 90), "p99", np.percentile(per_pair, 99), "max", per_pair.max())
S = sc.theory.bench_fixed_steps
pt = per_pair.reshape(-1, S).sum(1)
tile = 5
ntile = len(pt) // tile
tt = pt[: ntile * tile].reshape(ntile, tile).sum(1)
print("per-trajectory candidates: mean", pt.mean(), "p90", np.percentile(pt, 90), "max", pt.max())
print("per-tile(5) candidates: mean", tt.mean(), "p50", np.percentile(tt, 50), "p90", np.percentile(tt, 90), "p99", np.percentile(tt, 99), "max", tt.max(), " per-lane max", tt.max() / 256)
