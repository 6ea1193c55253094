#!/usr/bin/env python
"""Where one bench step's wall time goes, from a rocprofv3 --kernel-trace database: busy time per kernel and the idle gaps
between consecutive kernels (launch latency, host work), over the kernels between the first and the last MH launch.
usage: timeline.py <results.db> [skip_first_n_mh_launches]"""
import collections
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    mh = [i for i, r in enumerate(rows) if "mh_steps" in r[0] or "mh_dense" in r[0]]
    if len(mh) <= skip + 1:
        raise SystemExit("too few MH launches")
    lo, hi = mh[skip], mh[-1]
    win = rows[lo:hi]                                   # whole steps: from an MH launch up to (not including) the last one
    nsteps = len([i for i in mh if lo <= i < hi])
    busy, gaps = collections.Counter(), collections.Counter()
    calls = collections.Counter()
    for k, (name, s, e) in enumerate(win):
        short = name.split("(")[0].replace("void ", "")[:60]
        busy[short] += e - s
        calls[short] += 1
        nxt = rows[lo + k + 1]
        gaps[short + " -> " + nxt[0].split("(")[0].replace("void ", "")[:40]] += max(0, nxt[1] - e)
    total = rows[hi][1] - rows[lo][1]
    print("%d steps, %.3f ms per step on the device timeline" % (nsteps, total / nsteps * 1e-6))
    for n, t in busy.most_common():
        print("  busy %-62s %8.3f us/step  (%d calls)" % (n, t / nsteps * 1e-3, calls[n]))
    print("  idle between kernels %8.3f us/step" % (sum(gaps.values()) / nsteps * 1e-3))
    for n, t in gaps.most_common(12):
        print("     gap %-100s %8.3f us/step" % (n, t / nsteps * 1e-3))


if __name__ == "__main__":
    main()
