#!/usr/bin/env python3
"""Summarise a gfx950 .s file produced by `hipcc --cuda-device-only -S`: per kernel VGPR/SGPR/scratch
and an instruction histogram.  Usage: isa_summary.py file.s [kernel-substring]"""
import collections
import re
import sys

text = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else None
# kernel bodies
bodies = {}
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\s*s_endpgm", text, re.S | re.M):
    bodies[m.group(1)] = m.group(2)
meta = {}
for m in re.finditer(r"\.name:\s+(\S+)\n\s+\.private_segment_fixed_size:\s+(\d+)\n\s+\.sgpr_count:\s+(\d+)\n\s+\.sgpr_spill_count:\s+(\d+)\n.*?\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", text, re.S):
    meta[m.group(1)] = dict(scratch=int(m.group(2)), sgpr=int(m.group(3)), sspill=int(m.group(4)), vgpr=int(m.group(5)), vspill=int(m.group(6)))
for name, body in bodies.items():
    if pat and pat not in name:
        continue
    ins = [l.split()[0] for l in body.splitlines() if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    h = collections.Counter(ins)
    md = meta.get(name, {})
    print("%s\n   %s  total_instr=%d" % (name, md, len(ins)))
    if pat:
        groups = collections.Counter()
        for k, v in h.items():
            groups[k] += v
        for k, v in groups.most_common(40):
            print("      %-28s %d" % (k, v))
