#!/usr/bin/env python3
"""VERDICT round 2 item 5, measured instead of argued: what narrower neighbour rows would buy the walk.

Replays hnswlib's base-layer search (hnswalg.cpp:48-109) for a sample of queries on the bench's million-node graphs, with
the kernel's bookkeeping (FMODE 3: every linked neighbour's bound is judged each time it appears; a neighbour the bound
cannot reject has its 512-byte float row read and enters the visited set), and counts float rows read per expansion under
different LOWER BOUNDS of ||q - x|| -- all of them exact-safe (bound <= true distance, so results never change):
   b8      the shipped one: 8-bit rows, ||q' - c|| - errc                              128 B per neighbour
   b6, b4  the same with 6- / 4-bit components (96 / 64 B)
   b8r, b4r  8- / 4-bit rows with a PER-ROW error byte (in 1/16 steps) instead of the table's worst row (129 / 65 B)
   lead64  8-bit rows of the 64 highest-variance dims after a PCA rotation + the exact norm of the remaining 64 dims of x
           (4 B): sqrt(bound_lead^2 + (||q_R|| - ||x_R||)^2)                          68 B
From the survivors follows the HBM bytes per expansion: links + norms 256 B + degree x row bytes + survivors x 512 B.
usage: python tools/row_bits_probe.py [nq=300] [kind=sift|clustered] [nc=993127]"""
import heapq
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def quantised(x, bits):
    """(codes as float64, lo, step, errc): x ~ lo + step * c with c in 0 .. 2^bits - 1; errc = worst row error / step"""
    lo, hi = float(x.min()), float(x.max())
    step = (hi - lo) / (2 ** bits - 1)
    c = np.rint((x - lo) / step)
    rowerr = np.sqrt((((lo + step * c) - x).astype(np.float64) ** 2).sum(1)) / step
    return c.astype(np.float32), lo, step, rowerr.max(), np.ceil(rowerr * 16.0) / 16.0   # per row: one byte of 1/16 steps


def main():
    nq = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    kind = sys.argv[2] if len(sys.argv) > 2 else "sift"
    nc = int(sys.argv[3]) if len(sys.argv) > 3 else 993127
    import __graft_entry__ as ge
    import synth
    pkg = ge.load_pkg()
    d, ef = 128, 80
    tb = synth.make_throughput_tables(1234, nc, d, 16, 10 * nc, kind=kind)
    x = tb["centroids"]
    t0 = time.time()
    gb = pkg.GpuIndex(0)
    counts, links = gb.build_graph(x, 16, 32, 64)
    gb.close()
    links = links.reshape(nc, 32)
    print("[probe] %s table, %d nodes, graph in %.1fs, mean degree %.1f" % (kind, nc, time.time() - t0, counts.mean()), flush=True)
    rng = np.random.default_rng(1235)
    q_all = (x[rng.choice(nc, nq)] + rng.normal(0, 12.0, size=(nq, d))).astype(np.float32)

    # the candidate bounds
    forms = {}
    for bits in (8, 6, 4):
        c, lo, step, err, rowerr = quantised(x, bits)
        forms["b%d" % bits] = dict(c=c, lo=lo, step=step, err=err)
        if bits != 6:   # ... and with a per-row error byte instead of the table's worst row
            forms["b%dr" % bits] = dict(c=c, lo=lo, step=step, err=None, rowerr=rowerr)
    # PCA rotation from a sample; leading half in 8 bits, exact residual norm
    samp = x[rng.choice(nc, min(nc, 100000), replace=False)].astype(np.float64)
    mu = samp.mean(0)
    w, v = np.linalg.eigh(np.cov((samp - mu).T))
    R = v[:, ::-1].astype(np.float64)             # columns by falling variance
    energy = w[::-1].cumsum() / w.sum()
    xr = ((x.astype(np.float64) - mu) @ R).astype(np.float32)
    cl, lo_l, step_l, err_l, _ = quantised(xr[:, :64], 8)
    xres = np.sqrt((xr[:, 64:].astype(np.float64) ** 2).sum(1))
    print("[probe] PCA: leading 64 of 128 dims hold %.1f %% of the variance" % (100 * energy[63]), flush=True)

    def bounds(q, ids):
        out = {}
        for k_, f in forms.items():
            qp = (q.astype(np.float64) - f["lo"]) / f["step"]
            dist = np.sqrt(((qp[None, :] - f["c"][ids]) ** 2).sum(1))
            out[k_] = f["step"] * np.maximum(dist - (f["err"] if f["err"] is not None else f["rowerr"][ids]), 0.0)
        qr = (q.astype(np.float64) - mu) @ R
        qp = (qr[:64] - lo_l) / step_l
        lead = step_l * np.maximum(np.sqrt(((qp[None, :] - cl[ids]) ** 2).sum(1)) - err_l, 0.0)
        res = np.abs(np.sqrt((qr[64:] ** 2).sum()) - xres[ids])
        out["lead64"] = np.sqrt(lead ** 2 + res ** 2)
        return out

    names = ["b8", "b8r", "b6", "b4", "b4r", "lead64"]
    rowbytes = dict(b8=128, b8r=129, b6=96, b4=64, b4r=65, lead64=68)
    surv = {n: 0 for n in names}
    n_exp = n_nb = 0
    for qi in range(nq):
        q = q_all[qi]
        # one replay per bound (the visited set depends on what the bound lets through; the RESULT does not)
        res_ref = None
        for n in names:
            d0 = float(np.sqrt(((q - x[0]).astype(np.float64) ** 2).sum()))
            visited = {0}
            cand = [(d0, 0)]
            top = [(-d0, 0)]
            nsurv = nexp = nnb = 0
            while cand:
                dc, c = heapq.heappop(cand)
                if dc > -top[0][0]:
                    break
                nexp += 1
                ids = links[c, :counts[c]].astype(np.int64)
                nnb += len(ids)
                fresh = np.array([i for i in ids if int(i) not in visited], dtype=np.int64)
                if len(fresh) == 0:
                    continue
                if len(top) >= ef:
                    lb = bounds(q, fresh)[n]
                    fresh = fresh[lb < -top[0][0]]   # the kernel's rejection: bound >= max(topResults) settles the row
                nsurv += len(fresh)
                if len(fresh) == 0:
                    continue
                dd = np.sqrt(((q[None, :].astype(np.float64) - x[fresh]) ** 2).sum(1))
                for i, dist in zip(fresh.tolist(), dd.tolist()):
                    visited.add(i)
                    if len(top) < ef or dist < -top[0][0]:
                        heapq.heappush(cand, (dist, i))
                        heapq.heappush(top, (-dist, i))
                        if len(top) > ef:
                            heapq.heappop(top)
            found = sorted((-a, b) for a, b in top)
            if res_ref is None:
                res_ref = found
                n_exp += nexp
                n_nb += nnb
            else:
                assert [b for _, b in found] == [b for _, b in res_ref], "a bound changed the result"
            surv[n] += nsurv
        if (qi + 1) % 50 == 0:
            print("[probe] %d queries" % (qi + 1), flush=True)
    deg = n_nb / n_exp
    print("%s table: %.1f expansions per query, %.1f linked neighbours per expansion" % (kind, n_exp / nq, deg))
    print("| bound | bytes per neighbour row | float rows read per expansion | HBM bytes per expansion | vs shipped |")
    print("|---|---|---|---|---|")
    base = None
    for n in names:
        s = surv[n] / n_exp
        byt = 256 + deg * rowbytes[n] + s * 512
        base = base or byt
        print("| %s | %d | %.2f | %.0f | %.2f |" % (n, rowbytes[n], s, byt, byt / base))


if __name__ == "__main__":
    main()
