#!/usr/bin/env python3
"""Per-basic-block instruction statistics of one kernel in a gfx950 .s file (scratch / flat / global ops, mads,
s_nops, branch targets): shows at a glance whether a hot loop touches scratch.  Usage: isa_blocks.py file.s kernel-substring"""
import re
import sys

t = open(sys.argv[1]).read()
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\s*s_endpgm", t, re.S | re.M):
    if sys.argv[2] not in m.group(1):
        continue
    print(m.group(1))
    blk = "entry"
    keys = ("n", "scratch", "flat", "global", "ds", "mad", "nop")
    stats = {blk: dict.fromkeys(keys, 0)}
    br = {blk: []}
    order = [blk]
    for l in m.group(2).splitlines():
        mm = re.match(r"^(\.LBB\d+_\d+):", l)
        if mm:
            blk = mm.group(1); order.append(blk); stats[blk] = dict.fromkeys(keys, 0); br[blk] = []
            continue
        if not l.startswith("\t") or l.strip().startswith((".", ";")):
            continue
        op = l.split()[0]
        s = stats[blk]
        s["n"] += 1
        for k in ("scratch", "flat", "global", "ds"):
            if op.startswith(k):
                s[k] += 1
        s["mad"] += op in ("v_mad_u64_u32", "v_mad_i64_i32")
        s["nop"] += op == "s_nop"
        if op.startswith("s_cbranch") or op == "s_branch":
            br[blk].append(l.split()[-1])
            # the fall-through code after a branch is its own block
            sub = blk.split("+")[0] + "+%d" % (sum(1 for b in order if b.split("+")[0] == blk.split("+")[0]))
            blk = sub; order.append(blk); stats[blk] = dict.fromkeys(keys, 0); br[blk] = []
    for b in order:
        if stats[b]["n"]:
            print("  %-10s %s -> %s" % (b, " ".join("%s=%d" % kv for kv in stats[b].items()), ",".join(br[b])))
