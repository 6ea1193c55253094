"""One covariance epoch of the config-4 bench on the device timeline: the step launches with what ran beside them (developer tool).
usage: python tools/c4_epoch_trace.py <rocprofv3 results.db>"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
mh = [r for r in rows if "mh_steps" in r[0]]
t0 = mh[30][1]
sel = [r for r in rows if t0 <= r[1] <= mh[min(42, len(mh) - 1)][1]]
big = [r for r in sel if r[2] - r[1] > 200e3 or "mh_steps" in r[0] or "sytrd" in r[0]]
for name, s, e in big:
    print("%9.3f ms  +%8.3f ms  %s" % ((s - t0) * 1e-6, (e - s) * 1e-6, name.split("(")[0].replace("void ", "")[:60]))
