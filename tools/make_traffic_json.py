"""profiles/<tag>_traffic.json from the FETCH_SIZE / WRITE_SIZE summaries of tools/gpu_profile.sh.
usage: python tools/make_traffic_json.py r01"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"


def counter(path, name):
    for line in open(path):
        m = re.match(r"(.*mh_steps_kernel<[^>]*>)\(KArgs\)\s+%s\s+([0-9.]+)\s+(\d+)" % name, line)
        if m:
            return m.group(1).replace("void ", "").strip(), float(m.group(2))
    raise SystemExit("no %s line for mh_steps_kernel in %s" % (name, path))


pf, pw = (os.path.join(ROOT, "profiles", "%s_scam_%s.txt" % (tag, k)) for k in ("fetch", "write"))
kern, fetch_kb = counter(pf, "FETCH_SIZE")
_, write_kb = counter(pw, "WRITE_SIZE")
out = {
    "round": int(tag[1:]),
    "kernel": kern,
    "workload": "ndim=100 ntemps=64 nwalkers=4096 mix=scam logl=iso (am_mode rle: the rank-0 rows of accepted steps only)",
    "steps_per_launch": 100,
    "pick": "chain",
    "FETCH_SIZE_KB_per_dispatch": fetch_kb,
    "WRITE_SIZE_KB_per_dispatch": write_kb,
    "correction": "gfx950: FETCH_SIZE reports half of a wide coalesced read stream (MI355X_MICROARCH.md, HBM section): x2; WRITE_SIZE uncorrected",
    "traffic_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
    "source": ["profiles/%s_scam_fetch.txt" % tag, "profiles/%s_scam_write.txt" % tag],
    "command": "bash tools/gpu_profile.sh %s  (rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes)" % tag,
}
json.dump(out, open(os.path.join(ROOT, "profiles", "%s_traffic.json" % tag), "w"), indent=1)
print(json.dumps(out, indent=1))
