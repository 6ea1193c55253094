#!/bin/bash
# usage: tools/codeobj.sh <object.o> [out.co]  -- the gfx950 code object of a hipcc object file; prints per-kernel register / scratch / LDS metadata
set -e
O=$1; OUT=${2:-/tmp/$(basename ${O%.o}).co}
B=/opt/rocm/lib/llvm/bin
$B/llvm-objcopy --dump-section .hip_fatbin=$OUT.fb $O
$B/clang-offload-bundler --unbundle --type=o --input=$OUT.fb --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$OUT
rm -f $OUT.fb
$B/llvm-readelf --notes $OUT | python3 -c '
import sys, re
txt = sys.stdin.read()
for blk in txt.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
    print("%-110s vgpr %s agpr %s sgpr %s spill %s scratch %s lds %s" % (g("name")[:110], g("vgpr_count"), blk.split()[0], g("sgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
'
