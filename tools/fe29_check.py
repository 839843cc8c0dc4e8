#!/usr/bin/env python3
"""value check of tools/fe29_bench.bin's dump: for the first 64 lanes of every run, the chain's result (f * g^iters, f^(2^iters))
must be the same field element in both limb forms.  The ladder-step chains are checked against each other (fe26 vs fe29: the
same step sequence on the same values, since both draw their inputs from the same stream masked to the limb widths -- which
are different values, so only mul / sqr chains are comparable against the big-integer model)."""
import sys
P = (1 << 255) - 19
def val(limbs, nl):
    if nl == 10:
        pos = [(51 * i + 1) // 2 for i in range(10)]
    else:
        pos = [29 * i for i in range(9)]
    return sum(v << p for v, p in zip(limbs, pos)) % P
lines = open(sys.argv[1] if len(sys.argv) > 1 else "fe29_check.txt").read().splitlines()
i = 0; bad = 0; checked = 0
while i < len(lines):
    name, nl, kind, iters = lines[i].rsplit(" ", 3); nl, kind, iters = int(nl), int(kind), int(iters); i += 1
    for t in range(64):
        rows = [[int(x) for x in lines[i + r].split()] for r in range(6)]; i += 6
        f, g, out = val(rows[0], nl), val(rows[1], nl), val(rows[5], nl)
        if kind == 0:
            want = f * pow(g, iters, P) % P
        elif kind == 1:
            want = pow(f, pow(2, iters, P - 1), P)
        else:
            continue
        checked += 1
        if out != want:
            bad += 1
            if bad < 5: print("MISMATCH", name, "lane", t)
print("checked %d chains, %d mismatches" % (checked, bad))
sys.exit(1 if bad else 0)
