#!/usr/bin/env python3
"""loop_mix.py FILE.s SUBSTRING  -- static instruction mix of the largest loops of the kernels whose mangled name contains
SUBSTRING (device assembly from `hipcc -S --cuda-device-only`).  A quick way to see what a change adds to a hot loop."""
import re
import sys
from collections import Counter

lines = open(sys.argv[1]).read().split("\n")
sub = sys.argv[2]
starts = [i for i, l in enumerate(lines) if re.match(r"_Z\w+:", l)]
for si, a in enumerate(starts):
    name = lines[a].split(":")[0]
    if sub not in name:
        continue
    b = starts[si + 1] if si + 1 < len(starts) else len(lines)
    end = next((i for i in range(a, b) if ".Lfunc_end" in lines[i]), b)
    body = lines[a:end]
    labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"(\.LBB\d+_\d+):", l))}
    loops = []
    for i, l in enumerate(body):
        m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((i - labels[m.group(1)], labels[m.group(1)], i))
    loops.sort(reverse=True)
    tot = [l.split()[0] for l in body if re.match(r"\s+(s_|v_|ds_|global_|buffer_|scratch_)", l)]
    print(name, "instructions:", len(tot))
    for ln, x, y in loops[:4]:
        seg = [l.split()[0] for l in body[x:y + 1] if re.match(r"\s+(s_|v_|ds_|global_|buffer_|scratch_)", l)]
        c = Counter("valu" if s.startswith("v_") else "salu" if s.startswith("s_") else "lds" if s.startswith("ds_") else "vmem"
                    for s in seg)
        print("   loop of", len(seg), "instructions:", dict(c), "waitcnt", seg.count("s_waitcnt"), "branches",
              sum(1 for s in seg if "branch" in s))
