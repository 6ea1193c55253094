"""tools/rocpd_summary.py --window: the per-kernel summary of the TIMED REGION of a bench.py run (from the N-th last launch of the
dominant kernel on), which the committed profiles/r06_*_stats.txt are made with.  A synthetic rocpd database: warm-up launches with other
durations must stay out of the averages."""
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _db(path, rows):
    c = sqlite3.connect(path)
    c.execute("create table kernels (name text, start integer, end integer, grid_x integer, workgroup_x integer, lds_size integer, "
              "scratch_size integer, vgpr_count integer, accum_vgpr_count integer, sgpr_count integer)")
    c.executemany("insert into kernels values (?, ?, ?, 1, 64, 0, 0, 8, 0, 16)", rows)
    c.commit()
    c.close()


def _run(db, out, *extra):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocpd_summary.py"), db, out] + list(extra), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return open(out).read()


def test_window_summary_covers_the_timed_launches_only(tmp_path):
    rows, t = [], 0
    for i in range(5):                                   # warm-up: slow launches of the dominant kernel, a warm-up-only helper
        rows.append(("void mh_steps_kernel<4, 25>(KArgs)", t, t + 900, )); t += 1000
        rows.append(("warmup_only_kernel", t, t + 50)); t += 100
    for i in range(20):                                  # timed region: 20 launches of 700 ns, a swap behind each, two epochs
        rows.append(("void mh_steps_kernel<4, 25>(KArgs)", t, t + 700)); t += 800
        rows.append(("swap_fused_kernel", t, t + 20)); t += 30
        if i in (4, 14):
            rows.append(("pool_syrk_kernel", t, t + 1000)); t += 1100
    db, out = str(tmp_path / "r.db"), str(tmp_path / "r.txt")
    _db(db, rows)
    whole = _run(db, out)
    assert "warmup_only_kernel" in whole
    line = [ln for ln in whole.splitlines() if ln.startswith("void mh_steps_kernel")][0].split()
    assert int(line[-6]) == 25 and abs(float(line[-4]) - (5 * 900 + 20 * 700) / 25.0) < 1           # calls, average over everything
    win = _run(db, out, "--window", "20:mh_steps_kernel")
    assert "timed region only" in win and "(found)" in win and "warmup_only_kernel" not in win
    line = [ln for ln in win.splitlines() if ln.startswith("void mh_steps_kernel")][0].split()
    assert int(line[-6]) == 20 and float(line[-4]) == 700.0 and int(line[-3]) == 700 and int(line[-2]) == 700
    assert [ln for ln in win.splitlines() if ln.startswith("pool_syrk_kernel")][0].split()[1] == "2"
    assert [ln for ln in win.splitlines() if ln.startswith("swap_fused_kernel")][0].split()[1] == "20"
    # fewer launches than asked for: says so and summarises everything
    few = _run(db, out, "--window", "99:mh_steps_kernel")
    assert "NOT FOUND" in few and "warmup_only_kernel" in few
