#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd (.db) file: per-kernel duration statistics and, when the run
collected counters, per-kernel PMC means.  Usage: rocpd_summary.py <results.db> [out.txt] [--window N:PATTERN]

--window N:PATTERN summarises the TIMED REGION only: everything from the start of the N-th last dispatch of a kernel whose name
contains PATTERN (bench.py --steps N with --ess-window 0 --also off: the timed region's first launch of the dominant kernel; the
process ends with the timed region) -- the warm-up launches, with their other cycle composition and cold clocks, stay out of the
averages."""
import sqlite3
import sys


def main():
    args = [a for a in sys.argv[1:]]
    window = None
    if "--window" in args:
        i = args.index("--window")
        n, pat = args[i + 1].split(":", 1)
        window = (int(n), pat)
        del args[i:i + 2]
    sys.argv = [sys.argv[0]] + args
    db = sys.argv[1]
    c = sqlite3.connect(db)
    out = []
    where, t0 = "", None
    if window:
        starts = [r[0] for r in c.execute("select start from kernels where name like ? order by start", ("%" + window[1] + "%",))]
        if len(starts) >= window[0]:
            t0 = starts[-window[0]]
            where = " where start >= %d" % t0
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     "from kernels%s group by name order by 3 desc" % where).fetchall()
    if window:
        span = c.execute("select min(start), max(end) from kernels%s" % where).fetchone()
        out.append("# timed region only: from the %d-th last dispatch of *%s* on (%s); it spans %.3f ms, kernels busy %.3f ms" % (
            window[0], window[1], "found" if t0 is not None else "NOT FOUND: whole trace", (span[1] - span[0]) / 1e6 if span[0] else 0.0,
            sum(r[2] for r in rows) / 1e6))
    tot = sum(r[2] for r in rows) or 1
    out.append("# kernel-trace summary of %s" % db.split("/")[-1])
    out.append("# (a Cijk_* row of several hundred calls of about 0.25 ms = bench.py's --preheat matrix products: before the warmup steps, outside the timed region)")
    out.append("%-86s %7s %14s %14s %12s %12s %6s" % ("Name", "Calls", "TotalNs", "AvgNs", "MinNs", "MaxNs", "Pct"))
    for n, k, t, a, mn, mx in rows:
        out.append("%-86s %7d %14d %14.0f %12d %12d %6.2f" % (n[:86], k, t, a, mn, mx, 100.0 * t / tot))
    try:
        meta = c.execute("select name, max(grid_x), max(workgroup_x), max(lds_size), max(scratch_size), max(vgpr_count), "
                         "max(accum_vgpr_count), max(sgpr_count) from kernels where name not like '%%at::native%%'%s group by name" % where.replace(" where", " and")).fetchall()
    except sqlite3.Error:
        meta = []
    if meta:
        out.append("")
        out.append("%-86s %10s %6s %8s %8s %6s %6s %6s" % ("Name", "grid_x", "wg_x", "lds", "scratch", "vgpr", "agpr", "sgpr"))
        for m in meta:
            out.append("%-86s %10s %6s %8s %8s %6s %6s %6s" % ((m[0][:86],) + tuple(m[1:])))
    pm = []
    try:
        ccols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
        if ccols:
            kcol = "kernel_name" if "kernel_name" in ccols else "name"
            pm = c.execute("select %s, counter_name, avg(value), count(*) from counters_collection group by 1,2" % kcol).fetchall()
            # (counter rows carry no timestamps in every rocprofv3 version: the window applies to the durations above only)
    except sqlite3.Error as e:
        out.append("# counters: %s" % e)
    if pm:
        out.append("")
        out.append("%-86s %-24s %20s %8s" % ("Name", "Counter", "MeanPerDispatch", "N"))
        for n, p, v, k in pm:
            out.append("%-86s %-24s %20.1f %8d" % (n[:86], p, v, k))
    text = "\n".join(out) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
