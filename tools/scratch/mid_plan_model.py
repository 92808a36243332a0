"""Cost model of the planned trailing updates of the mid-size factorisation (design aid for madqp_jl_amd/csrc/mid_plan.inc, never
part of the product): constant budget of R rounds per step, smallest R whose plan never needs more than qmax panels in a visit."""
import sys


def t_unit(q):
    return 4.0 + 15.0 * q


def plan(nblk, budget, qmax=2):
    upto = [0] * nblk
    out = []
    for k in range(nblk):
        visits, units = [], 0
        if k > 0:
            cand = []
            for j in range(k, nblk):
                pend = k - upto[j]
                if pend <= 0:
                    continue
                key = (-1000 + j) if j <= k + 1 else (j - k) - (pend + qmax - 1) // qmax
                cand.append((key, j, pend))
            cand.sort()
            for key, j, pend in cand:
                h = nblk - j - (1 if j == k else 0)
                forced = key <= 0
                q = pend if j <= k + 1 else min(qmax, pend)
                if not forced and (q < qmax or units + h > budget):
                    continue
                if h > 0:
                    visits.append((j, upto[j], q))
                units += h
                upto[j] += q
        out.append((k, visits, units))
    return out


def cost(p, nblk, cap, chain=36.5, panel=10.5):
    total = 0.0
    for k, visits, units in p:
        tmax = max([t_unit(q) for _, _, q in visits], default=0.0)
        rounds = max(1, -(-units // cap))
        total += max(chain if k > 0 else 25.0, tmax * rounds) + (panel if k + 1 < nblk else 0.0)
    return total


def best(nblk, cap=254, qmax=2):
    for R in range(1, 40):
        p = plan(nblk, R * cap, qmax)
        if all(q <= qmax for _, v, _ in p for _, _, q in v) and all(u <= R * cap for _, _, u in p):
            return R, p
    return None, None


if __name__ == "__main__":
    for nblk in (8, 16, 24, 32, 40, 48, 56, 63, 72, 79, 80):
        R, p = best(nblk)
        ideal = sum((nblk - j) * j for j in range(nblk)) * 14.7 / 256
        cur = 0.0  # today's two-panel schedule: every second column, K = 256, pairs beyond one round
        for k in range(nblk):
            rem = nblk - k
            n2 = (rem - 1) + sum(rem - 2 * c for c in range(1, rem) if 2 * c < rem)
            rounds = max(1, -(-n2 // 254))
            cur += max(36.5 if k else 25.0, (t_unit(2) if k > 1 else t_unit(1)) * rounds * (1.05 if rounds > 1 else 1.0)) + (10.5 if k + 1 < nblk else 0)
        print(f"nblk {nblk:2d}: R {R} lazy {cost(p, nblk, 254):7.0f} us   two-panel model {cur:7.0f} us   chain {nblk * 47:5d}   mfma-ideal {ideal:6.0f}")
