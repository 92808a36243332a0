#!/usr/bin/env python3
"""Per-launch efficiency of gemm_tn_f64_kernel from a rocprofv3 kernel trace of bench.py (n = 50000):
replays the host schedule of chol.hip to label every launch with (M, N, K, tiles)."""
import collections
import csv
import glob
import sys

NB = 128


def outer_w(rows, has_update, slots=512):
    mt = (rows + NB - 1) // NB
    if not has_update or mt <= 6:
        return 1024
    best, best_eff = 8, -1.0
    for wt in range(6, min(20, mt) + 1):  # MADQP_CHOL_WMIN / WMAX defaults of chol.hip
        tiles = mt * wt - wt * (wt - 1) // 2
        rounds = (tiles + slots - 1) // slots
        eff = tiles / (rounds * slots)
        if eff > best_eff + 1e-9 or (eff > best_eff - 0.01 and wt > best and eff > 0.97):
            best_eff = max(eff, best_eff)
            best = wt
    return best * NB


def tiles_lower(M, N):
    tm, tn = (M + 127) // 128, (N + 127) // 128
    return sum(min(tm, tm - b) if False else (tm - b if b < tm else 0) for b in range(tn))


sched = []


def panel_update(n, row0, k0, width, kind="upd"):
    sched.append((kind, n - row0, width, row0 - k0, tiles_lower(n - row0, width)))


def factor_range(n, j0, w):
    if w <= NB:
        if j0 + w < n:
            below = n - j0 - w
            # chol.hip factor_block (round 4): panel_sub16_kernel, block substitution, 64 rows per workgroup
            sched.append(("trsm64", below, w, w, (below + 63) // 64))
        return
    h = ((w + NB - 1) // NB + 1) // 2 * NB
    factor_range(n, j0, h)
    panel_update(n, j0 + h, j0, w - h)
    factor_range(n, j0 + h, w - h)


n = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
J0 = 0
while J0 < n:
    W = min(outer_w(n - J0, J0 > 0), n - J0)
    if J0 > 0:
        panel_update(n, J0, 0, W, "outer")
    factor_range(n, J0, W)
    J0 += W
f = glob.glob(sys.argv[1])[0]
rows = [r for r in csv.DictReader(open(f)) if "gemm_tn" in r["Kernel_Name"] or "panel_sub16" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6 for r in rows]
grids = [int(r["Grid_Size_X"]) // 256 for r in rows]
tm_all = (n + 127) // 128
syrk_tiles = tm_all * (tm_all + 1) // 2
nsy, acc = 0, 0
while acc < syrk_tiles:  # the assembly may be cut into several launches (segments)
    acc += grids[nsy]
    nsy += 1
print(len(sched), "gemm-class launches per factorization in the host schedule;", len(rows), "in trace;", nsy, "assembly launches")
# a scheduled product may be several launches (64-round segments; whole rounds + the K-split tail of madqp_gemm_tn): consume
# launches until their tiles cover the scheduled ones, add their times
seg, pos = [], nsy
for item in sched:
    need, got, d = item[4], 0, 0.0
    while got < need and pos < len(rows):
        got += grids[pos]
        d += durs[pos]
        pos += 1
    seg.append((item, d, got))
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
mismatch = 0
for (kind, M, N, K, t), d, g in seg:
    mismatch += int(t != g)  # replayed schedule vs traced grid (labels are approximate if > 0)
    t = g
    if kind == "trsm64":  # 64 x 128 outputs per workgroup by block substitution: flops of the full product for comparison
        agg[kind][0] += 2.0 * t * 64 * 128 * K
        agg[kind][1] += d
        agg[kind][2] += 1
        continue
    agg[kind][0] += 2.0 * t * 128 * 128 * K
    agg[kind][1] += d
    agg[kind][2] += 1
sy_ms = sum(durs[:nsy])
print("syrk", round(sy_ms, 1), "ms", round(2.0 * syrk_tiles * 128 * 128 * 20000 / sy_ms * 1e-9, 1), "TF")
for k, (fl, d, c) in agg.items():
    print(k, c, "launches", round(d, 1), "ms", round(fl / d * 1e-9, 1), "TF (computed flops, full diagonal tiles)")
print("launches whose replayed tile count differs from the trace:", mismatch)
print("outer panels: M, W, K, tiles, ms, TF, tiles/512")
for (kind, M, N, K, t), d, g in seg:
    if kind == "outer":
        print(M, N, K, t, round(d, 2), round(2.0 * t * 128 * 128 * K / d * 1e-9, 1), round(t / 512, 2))
lev = collections.defaultdict(lambda: [0.0, 0.0, 0])
for (kind, M, N, K, t), d, g in seg:
    if kind == "upd":
        lev[K][0] += 2.0 * t * 128 * 128 * K
        lev[K][1] += d
        lev[K][2] += 1
for K in sorted(lev):
    print("upd K=", K, lev[K][2], "launches", round(lev[K][1], 1), "ms", round(lev[K][0] / lev[K][1] * 1e-9, 1), "TF")
