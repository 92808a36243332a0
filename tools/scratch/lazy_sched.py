"""Model of lazy trailing-update schedules for the mid-size factorisation (tools/scratch: design aid, not product).
Columns j > k carry `upto[j]` = panels applied so far; a visit of column j at step k applies panels [upto[j], upto[j]+q)
to every tile (i, j), i >= j (one workgroup per tile, K = 128 q)."""
import sys


def t_unit(q):
    return 4.0 + 15.0 * q


def simulate(nblk, cap=254, qmax=2, chain=36.5, panel=10.5, verbose=False, policy="edf"):
    upto = [0] * nblk
    total = 0.0
    for k in range(nblk):
        visits = []  # (col, q)
        units = 0
        if k > 0:
            # mandatory: column k completely (its diagonal tile is the diagonal workgroup's: panel k-1 only, the
            # rest arrived with column k's visit as "k+1" in the step before)
            need = k - upto[k]
            if need > 0:
                visits.append((k, need, nblk - k - 1))
                units += nblk - k - 1
                upto[k] = k
            if k + 1 < nblk:
                need = k - upto[k + 1]
                if need > 0:
                    visits.append((k + 1, need, nblk - k - 1))
                    units += nblk - k - 1
                    upto[k + 1] = k
            # optional, by slack: steps left until the column is due minus the visits it still needs
            cand = []
            for j in range(k + 2, nblk):
                pend = k - upto[j]
                if pend <= 0:
                    continue
                h = nblk - j
                if policy == "edf":
                    key = (j - k) - (pend + qmax - 1) // qmax
                else:
                    key = -pend
                cand.append((key, j, pend, h))
            cand.sort()
            for key, j, pend, h in cand:
                q = min(qmax, pend)
                if q < qmax and key > 1:  # a single pending panel can wait for its partner unless the column is nearly due
                    continue
                if units + h > cap:
                    continue
                visits.append((j, q, h))
                units += h
                upto[j] += q
        tmax = max([t_unit(q) for _, q, _ in visits], default=0.0)
        rounds = max(1, -(-units // cap))
        step = max(chain if k > 0 else 25.0, tmax * rounds) + (panel if k + 1 < nblk else 0.0)
        total += step
        if verbose:
            print(f"k={k:2d} units {units:4d} maxq {max([q for _, q, _ in visits], default=0)} step {step:5.1f}  visits {[(j, q) for j, q, _ in visits][:12]}")
    return total


if __name__ == "__main__":
    nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    for qmax in (2, 3):
        print("qmax", qmax, "total us", simulate(nblk, qmax=qmax, verbose=(qmax == 2 and len(sys.argv) > 2)))
