"""profiles/r06_callback_traffic.json from the FETCH_SIZE / WRITE_SIZE summaries of tools/r6_callback.sh (the split path, one entry per
callback kind): HBM bytes per ITERATION of every chain = the accept + propose launch and the callback's kernels.
usage: python tools/make_callback_traffic.py"""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counters(path, name):
    """{kernel name: (mean per dispatch in KB, dispatches)}"""
    out = {}
    for line in open(path):
        m = re.match(r"(.*?)\s+%s\s+([0-9.]+)\s+(\d+)\s*$" % name, line)
        if m:
            out[m.group(1).strip()] = (float(m.group(2)), int(m.group(3)))
    return out


res = {}
for kind, tag in (("hip", "callback"), ("norm", "callbacktorch")):
    pf, pw = (os.path.join(ROOT, "profiles", "r06_%s_%s.txt" % (tag, k)) for k in ("fetch", "write"))
    if not (os.path.exists(pf) and os.path.exists(pw)):
        continue
    f, w = counters(pf, "FETCH_SIZE"), counters(pw, "WRITE_SIZE")
    step = [k for k in f if "split_rows_kernel<true, true" in k][0]
    n_step = f[step][1]
    per_iter, parts = 0.0, {}
    for k in f:
        # the kernels that run once per iteration: the step launch and the callback's (as many dispatches as the step launch, or one more per segment)
        if k == step or (abs(f[k][1] - n_step) <= 0.02 * n_step and k in w):
            b = (2.0 * f[k][0] + w[k][0]) * 1024.0
            parts[k[:90]] = {"FETCH_SIZE_KB": f[k][0], "WRITE_SIZE_KB": w[k][0], "bytes": b}
            per_iter += b
    res[kind] = {
        "workload": "ndim=100 ntemps=64 nwalkers=4096 mix=scam logl=iso, likelihood in a batched callback (%s), ptmi_accept_propose" % kind,
        "launches": "one", "callback_kind": kind, "kernels_per_iteration": parts,
        "traffic_bytes_per_iteration": per_iter,
        "correction": "gfx950: FETCH_SIZE reports half of a wide coalesced read stream (MI355X_MICROARCH.md, HBM section): x2; WRITE_SIZE uncorrected",
        "source": ["profiles/r06_%s_fetch.txt" % tag, "profiles/r06_%s_write.txt" % tag],
        "command": "bash tools/r6_callback.sh (rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py --callback)",
    }
json.dump(res, open(os.path.join(ROOT, "profiles", "r06_callback_traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
