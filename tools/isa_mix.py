#!/usr/bin/env python3
"""Dynamic instruction mix of a kernel from its gfx950 code object, without running it (CPU).

  python tools/isa_mix.py <object glob> <kernel substring> [--loops] [--json out.json] [--trip NAME=N ...]

The VALU-bound kernels of this library (ladders, scalar multiplications, fused curve kernels) have UNIFORM control flow: rolled
loops on scalar counters with compile-time trip counts, nothing that depends on lane data (tools/ct_audit.py checks exactly that).
So the number of times each instruction executes per pass of the kernel body is a static property:

  * loops = backward branches of the disassembly (interval [target, branch]; intervals with one target merged; nesting by containment);
  * trip count of a loop = (bound - init) / step of its scalar counter: the s_cmp / s_cmpk against an immediate inside the loop
    whose register is stepped by an `s_add_i32 sX, sX, imm` (or s_addk / s_sub) inside the loop, `init` from the last
    s_mov / s_movk of that register in front of the loop.  A loop with no such counter (the grid-stride loop over records, the
    `rounds` loops of the shared inversions) counts ONCE unless --trip <first 5 hex digits of its head address>=N names it;
  * an `if (i != 0)`-style region inside a loop (forward branch over part of the body) counts as executed every iteration
    (error 1 / trips on that region).

Each instruction is put into a class (multiplier = v_mad_u64_u32 / v_mad_i64_i32; mul32 = v_mul_lo / v_mul_hi / 24-bit forms, true 64-bit adds, carry adds, 64-bit shifts,
32-bit add, logic, select, move, compare, LDS, vector memory, scratch, scalar, wait / nop) and weighted by the product of the trip
counts around it.  Issue cycles per class per wave and SIMD: 5.0 for the multiplier class, 2.5 for other VALU (profiles/history/r01_valubench.log:
measured), listed per class so that a table like docs/kernels_field.md's "what is left in a ladder step" can be made for any kernel.

The result is validated where counters exist: sum(VALU) here against SQ_INSTS_VALU x 64 / records of a rocprofv3 --pmc pass
(profiles/history/r05_valu_pmc.json records both)."""
import fnmatch
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ct_audit  # noqa: E402  (disassemble())

MUL = ("v_mad_u64_u32", "v_mad_i64_i32")        # the 32 x 32 + 64 multiply-add: the instruction the field products are made of
MUL32 = ("v_mul_hi_u32", "v_mul_lo_u32", "v_mul_hi_i32", "v_mul_u32_u24", "v_mad_u32_u24", "v_mul_hi_u32_u24", "v_mad_u32_u16", "v_mul_i32_i24", "v_mad_i32_i24")
CYCLES = {"multiplier": 5.0}          # every other VALU class: 2.5 (the two-cost model of bench.py valu_roofline, rounds 2-5)


def classify(op):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if base in MUL:
        return "multiplier"
    if base in MUL32:
        return "mul32"
    if base.startswith(("v_lshl_add_u64", "v_add_u64", "v_sub_u64")):
        return "add_u64"
    if base.startswith(("v_add_co", "v_addc_co", "v_sub_co", "v_subb_co", "v_subrev_co", "v_subbrev_co")):
        return "add_carry"
    if base.startswith(("v_lshrrev_b64", "v_lshlrev_b64", "v_ashrrev_i64")):
        return "shift64"
    if base.startswith(("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_add3_u32", "v_lshl_add_u32", "v_add_lshl_u32", "v_add_nc", "v_sub_nc", "v_mad_u32")):
        return "add32"
    if base.startswith(("v_lshrrev_b32", "v_lshlrev_b32", "v_ashrrev_i32", "v_alignbit", "v_alignbyte", "v_bfe", "v_lshl_or", "v_perm")):
        return "shift32"
    if base.startswith(("v_and_or", "v_and", "v_or", "v_xor", "v_not", "v_bfi", "v_or3", "v_xad", "v_xnor")):
        return "logic"
    if base.startswith("v_cndmask"):
        return "select"
    if base.startswith(("v_mov", "v_accvgpr", "v_readfirstlane", "v_readlane", "v_writelane", "v_swap")):
        return "move"
    if base.startswith("v_cmp"):
        return "compare"
    if base.startswith("ds_"):
        return "lds"
    if base.startswith("scratch_"):
        return "scratch"
    if base.startswith(("global_", "flat_", "buffer_")):
        return "vmem"
    if base.startswith(("s_waitcnt", "s_nop")):
        return "wait"
    if base.startswith("s_"):
        return "scalar"
    if base.startswith("v_"):
        return "valu_other"
    return "other"


VALU_CLASSES = ("multiplier", "mul32", "add_u64", "add_carry", "shift64", "add32", "shift32", "logic", "select", "move", "compare", "valu_other")


def _imm(tok):
    tok = tok.strip()
    try:
        return int(tok, 0)
    except ValueError:
        return None


def _target(addr, ins):
    off = int(ins.split()[-1])
    if off >= 32768:
        off -= 65536
    return addr + 4 + 4 * off


def loops_of(code):
    """[(head, tail)] address intervals of the natural loops (backward branches), innermost last"""
    heads = {}
    for a, t in code:
        if a is None:
            continue
        if t.startswith(("s_cbranch", "s_branch")):
            tg = _target(a, t)
            if tg <= a:
                heads[tg] = max(heads.get(tg, a), a)
    # merge loops whose intervals cross (rotated loops with two back edges land on one head already)
    return sorted(heads.items(), key=lambda kv: (kv[0], -kv[1]))


def _s32(v):
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v >= (1 << 31) else v


def trip_count(code, head, tail, inner):
    """trip count of the loop [head, tail] by running its scalar counter, or None.  inner: intervals of nested loops (skipped).
    Counter = a register compared with an immediate (s_cmp / s_cmpk) whose compare decides a branch that LEAVES the loop (taken to
    an address outside it, or the loop's last instruction falling out of it), stepped inside the loop by `s_add sX, sX, imm` or
    through a temporary (`s_add sT, sX, imm` ... `s_mov sX, sT`); initial value from the last s_mov / s_movk in front of the head."""
    idx = {a: i for i, (a, _) in enumerate(code) if a is not None}
    body = [(a, t) for a, t in code if a is not None and head <= a <= tail and not any(h <= a <= e for h, e in inner)]
    adds, movs = {}, {}
    for k, (a, t) in enumerate(body):
        m = re.match(r"s_(add|sub)_[iu]32 (s\d+), (s\d+), (-?\w+)$", t)
        if m and _imm(m.group(4)) is not None:
            st = _s32(_imm(m.group(4)))
            adds[m.group(2)] = (m.group(3), -st if m.group(1) == "sub" else st, k)
            continue
        m = re.match(r"s_addk_i32 (s\d+), (-?\w+)$", t)
        if m and _imm(m.group(2)) is not None:
            v = _imm(m.group(2)) & 0xFFFF
            adds[m.group(1)] = (m.group(1), v - 65536 if v >= 32768 else v, k)
            continue
        m = re.match(r"s_mov_b32 (s\d+), (s\d+)$", t)
        if m:
            movs[m.group(1)] = (m.group(2), k)
    steps = {}                                  # counter register -> (step, position of its update in the body)
    for dst, (src, st, k) in adds.items():
        if dst == src:
            steps[dst] = (st, k)
    for dst, (tmp, k) in movs.items():
        if tmp in adds and adds[tmp][0] == dst and tmp != dst:
            steps[dst] = (adds[tmp][1], k)
    best = None
    for k, (a, t) in enumerate(body):
        m = re.match(r"s_cmpk?_(lg|eq|lt|gt|le|ge)_([iu])32 (s\d+), (-?\w+)$", t)
        if not m or m.group(3) not in steps or _imm(m.group(4)) is None:
            continue
        op, sign, reg, bound = m.group(1), m.group(2), m.group(3), _imm(m.group(4))
        if "cmpk" in t:
            bound = (bound & 0xFFFF) - (65536 if (bound & 0x8000) and sign == "i" else 0)
        bound = _s32(bound) if sign == "i" else bound & 0xFFFFFFFF
        # the branch this compare decides: the next scc branch in the body
        br = next(((a2, t2) for a2, t2 in body[k + 1:] if t2.startswith("s_cbranch_scc")), None)
        if br is None:
            continue
        tg, on = _target(br[0], br[1]), br[1].startswith("s_cbranch_scc1")
        if not (head <= tg <= tail):
            exit_when = on                      # taken = leaves the loop
        elif tg == head and br[0] == tail:
            exit_when = not on                  # the back edge itself: falling through leaves the loop
        else:
            continue                            # an if () inside the body
        init = None
        for j in range(idx[head] - 1, -1, -1):
            mm = re.match(r"s_movk?_[ib]32 %s, (-?\w+)$" % reg, code[j][1])
            if mm and _imm(mm.group(1)) is not None:
                init = _s32(_imm(mm.group(1)))
                break
            mm = re.match(r"s_mov_b64 s\[(\d+):(\d+)\], (-?\w+)$", code[j][1])
            if mm and _imm(mm.group(3)) is not None and int(mm.group(1)) <= int(reg[1:]) <= int(mm.group(2)):
                init = _s32(_imm(mm.group(3))) if int(reg[1:]) == int(mm.group(1)) else (0 if _imm(mm.group(3)) >= 0 else -1)
                break
            if re.match(r"s_\w+ %s[, ]" % reg, code[j][1]) and not code[j][1].startswith(("s_cmp", "s_cmpk")):
                break
        if init is None:
            continue
        step, upd = steps[reg]
        test = {"lg": lambda v: v != bound, "eq": lambda v: v == bound, "lt": lambda v: v < bound, "gt": lambda v: v > bound,
                "le": lambda v: v <= bound, "ge": lambda v: v >= bound}[op]
        v, n = init, 0
        while n < 1000000:
            n += 1
            if upd < k:
                v = _s32(v + step) if sign == "i" else (v + step) & 0xFFFFFFFF
            scc = test(v if sign == "i" else v & 0xFFFFFFFF)
            if upd > k:
                v = _s32(v + step) if sign == "i" else (v + step) & 0xFFFFFFFF
            if scc == exit_when:
                break
        else:
            continue
        if best is None or n > best:
            best = n
    return best


def analyse(code, trips_override=None, unknown_trips=1):
    """unknown_trips: what a loop without a compile-time counter counts for (the `rounds` loops of the shared inversions: pass rounds)"""
    loops = loops_of(code)
    info = []
    for h, e in loops:
        inner = [(h2, e2) for h2, e2 in loops if (h2, e2) != (h, e) and h <= h2 and e2 <= e]
        key = "%05x" % h
        n = (trips_override or {}).get(key)
        src = "--trip"
        if n is None:
            n = trip_count(code, h, e, inner)
            src = "counter"
        if n is None:
            # (a `continue` inside a loop is a second backward branch, to the latch: it shows up as an interval inside the loop's own;
            # only the outermost of nested counter-less intervals is given the caller's trip count)
            nested = any(h2 <= h and e <= e2 and (h2, e2) != (h, e) and li2["source"] != "counter" for (h2, e2), li2 in zip(loops, info))
            n = 1 if nested else unknown_trips
            src = "unknown (counted %d x)" % n
        info.append({"head": key, "tail": "%05x" % e, "trips": n, "source": src, "instructions": sum(1 for a, _ in code if a is not None and h <= a <= e)})
    counts, static = {}, {}
    for a, t in code:
        if a is None or not t:
            continue
        w = 1
        for (h, e), li in zip(loops, info):
            if h <= a <= e:
                w *= li["trips"]
        c = classify(t.split()[0])
        counts[c] = counts.get(c, 0) + w
        static[c] = static.get(c, 0) + 1
    valu = sum(counts.get(c, 0) for c in VALU_CLASSES)
    mad = counts.get("multiplier", 0)
    cycles = {c: counts.get(c, 0) * CYCLES.get(c, 2.5) for c in VALU_CLASSES}
    tot = sum(cycles.values())
    return {"loops": info, "dynamic": counts, "static": static, "valu": valu, "multiplier": mad,
            "non_multiplier_per_multiplier": (valu - mad) / mad if mad else None,
            "issue_cycles": cycles, "issue_cycles_total": tot, "issue_cost_of_mix_cycles": tot / valu if valu else None,
            "multiplier_share_of_issue_cycles": cycles["multiplier"] / tot if tot else None}


def kernels(obj_glob, pat):
    out = {}
    paths = glob.glob(obj_glob) or glob.glob(os.path.join(ROOT, "modarith_amd", "build", obj_glob)) or glob.glob(os.path.join(ROOT, "modarith_amd", "plugins", obj_glob))
    for obj in sorted(paths):
        funcs = ct_audit.disassemble(obj)
        if not funcs:
            continue
        names = subprocess.run(["c++filt"] + list(funcs), capture_output=True, text=True).stdout.splitlines()
        for sym, nm in zip(funcs, names):
            if pat in nm or fnmatch.fnmatch(nm, pat):
                out[(os.path.basename(obj), re.sub(r"^void ", "", nm))] = funcs[sym]
    return out


def main(argv):
    args = [a for a in argv if not a.startswith("--")]
    trips = {}
    for i, a in enumerate(argv):
        if a == "--trip":
            k, v = argv[i + 1].split("=")
            trips[k] = int(v)
            args.remove(argv[i + 1])
    res = {}
    for (obj, name), code in kernels(args[0], args[1]).items():
        r = analyse(code, trips)
        res["%s [%s]" % (name.split("(")[0], obj)] = r
        print("%s [%s]" % (name.split("(")[0], obj))
        if "--loops" in argv:
            for li in r["loops"]:
                print("    loop %s..%s  %6d instructions  x %-5d (%s)" % (li["head"], li["tail"], li["instructions"], li["trips"], li["source"]))
        print("    VALU %d   multiplier %d   other / multiplier %.2f   issue cycles %.0f (multiplier share %.2f)" % (
            r["valu"], r["multiplier"], r["non_multiplier_per_multiplier"] or 0, r["issue_cycles_total"], r["multiplier_share_of_issue_cycles"] or 0))
        print("    " + "  ".join("%s %d" % (c, r["dynamic"].get(c, 0)) for c in VALU_CLASSES + ("lds", "vmem", "scratch", "scalar", "wait")))
    if "--json" in argv:
        with open(argv[argv.index("--json") + 1], "w") as f:
            json.dump(res, f, indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
