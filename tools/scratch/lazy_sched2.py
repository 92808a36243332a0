"""Second model: forced visits + pace-keeping rounds (design aid)."""
import sys


def t_unit(q):
    return 4.0 + 15.0 * q


def plan(nblk, cap=254, qmax=2):
    """Yields per step k the list of visits (col, first_panel, q)."""
    upto = [0] * nblk
    for k in range(nblk):
        visits = []
        units = 0
        if k > 0:
            work_left = sum((nblk - j) * (j - upto[j]) for j in range(k, nblk))
            steps_left = nblk - k
            rounds = max(1, -(-work_left // (qmax * steps_left * cap)))
            budget = rounds * cap
            cand = []
            for j in range(k, nblk):
                pend = k - upto[j]
                if pend <= 0:
                    continue
                due = j - k  # steps until the column must be complete (k+1: complete through k-1 one step early)
                if j == k or j == k + 1:
                    key = -1000 + j
                else:
                    key = due - (pend + qmax - 1) // qmax
                cand.append((key, j, pend))
            cand.sort()
            for key, j, pend in cand:
                h = nblk - j - (1 if j == k else 0)
                forced = key <= 0
                q = pend if j <= k + 1 else min(qmax, pend)
                if not forced:
                    if q < qmax:
                        continue
                    if units + h > budget:
                        continue
                if h > 0:
                    visits.append((j, upto[j], q))
                units += h
                upto[j] += q
        yield k, visits, units


def simulate(nblk, cap=254, qmax=2, chain=36.5, panel=10.5, verbose=False):
    total = 0.0
    for k, visits, units in plan(nblk, cap, qmax):
        tmax = max([t_unit(q) for _, _, q in visits], default=0.0)
        rounds = max(1, -(-units // cap))
        step = max(chain if k > 0 else 25.0, tmax * rounds) + (panel if k + 1 < nblk else 0.0)
        total += step
        if verbose:
            print(f"k={k:2d} units {units:4d} rounds {rounds} maxq {max([q for _, _, q in visits], default=0)} step {step:5.1f}")
    return total


if __name__ == "__main__":
    for nblk in (16, 24, 40, 63, 79):
        ideal = sum((nblk - j) * j for j in range(nblk)) * 14.7 / 256
        print(nblk, "total us", simulate(nblk, verbose=(len(sys.argv) > 1 and int(sys.argv[1]) == nblk)), "chain", nblk * 47, "mfma-ideal", round(ideal))
