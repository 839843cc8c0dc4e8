#!/usr/bin/env python3
"""Mechanical constant-time guard for the kernels that carry the reference's constant-time contract (pseudo.py:979-1048
modcsw / modcmv; edwards.c:382-401 select; the ladders of rfc7748.c:156-256; the fixed-window ecnXXXmul of edwards.c:435-482).

  python tools/ct_audit.py [--json out.json] [--verbose]

Disassembles the gfx950 code objects of modarith_amd/build/*.o, finds every conditional branch of the audited kernels and
classifies what its condition is made of, by tracing the condition register back through the instruction stream:

  scc   s_cbranch_scc0/1: the s_cmp / s_bitcmp / s_add that set SCC, and the producers of its SGPR operands.
        uniform   = loop counters, kernel arguments, constants (s_mov / s_add / s_load from the kernarg segment ...)
        lane-data = an operand that came out of the vector unit (v_readfirstlane / v_readlane / a v_cmp mask): a wave-level
                    decision on what the lanes hold -- a data-dependent branch
  vcc   s_cbranch_vccz/nz: the instruction that wrote VCC.  s_and / s_andn2 / s_or of EXEC with an SGPR pair that holds a
        constant or an s_cselect of SCC is the compiler's way of branching on a uniform condition (uniform); a v_cmp is a vote
        on what the lanes hold, and its VGPR operands are traced on: if everything they are computed from is the workitem id
        (VGPRs never written before) and uniform values, the vote is a `t < n` guard (lane-index: counted with the exec class);
        if a per-lane memory load feeds it, it is a vote on lane data (lane-data)
  exec  s_cbranch_execz/nz: divergence -- some lanes skip a region.  Constant-time only if the lane mask depends on the lane
        index and the batch size alone (the `t < n` guard and the grid-stride back-edge); cannot be told apart mechanically,
        so the NUMBER of such branches per kernel is pinned in the allow-list with the reviewed reason.

tools/ct_allowlist.json holds, per kernel pattern, the allowed count of exec branches and of lane-data branches (0 unless a
documented exception) with the justification; tests/test_ct_audit.py fails when a kernel exceeds its entry or an audited kernel
has no entry -- a regression (e.g. the compiler turning a predicated table scan into `if (eq) load`) cannot slip in silently.
No GPU needed.
"""
import fnmatch
import json
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALLOW = os.path.join(ROOT, "tools", "ct_allowlist.json")
# (object glob, demangled-name substring) of the audited kernels
AUDITED = [
    ("capi_X25519.o", "k_cond<"), ("capi_NIST256.o", "k_cond<"), ("capi_X448.o", "k_cond<"),
    ("capi_X25519.o", "k_x25519_fe26_xz"), ("capi_X25519.o", "k_x25519_fe26("), ("capi_X448.o", "k_x448_fe28_xz"), ("capi_X448.o", "k_x448_fe28("),
    ("capi_X25519.o", "k_fe_finish<"), ("capi_X448.o", "k_fe_finish<"),
    ("capi_X25519.o", "k_rfc7748<"), ("capi_X448.o", "k_rfc7748<"),
    ("capi_*_part1.o", "k_ed_mul<"), ("capi_*_part2.o", "k_ed_mul2<"),
    # the fused kernels that take a SECRET scalar (key generation, signing, key agreement): mul + get, gen + mul + get, base-point ladders.
    # (mul2_get / mulgen2_get are the verification patterns: public inputs, and the reference's own mul2 is variable time.)
    ("capi_NIST256F.o", "_mul_get"), ("capi_SECP256K1F.o", "_mul_get"),
    ("capi_ED25519G.o", "k_ed25519_mulgen<"), ("capi_ED448G.o", "k_ed448_mulgen<"), ("capi_NIST256G.o", "mulgen_get"), ("capi_SECP256K1G.o", "mulgen_get"),
    ("capi_ED25519G.o", "k_x25519_base"), ("capi_ED448G.o", "k_x448_base"),
    # round 5: the ladder form of the fused ED25519 multiplications (csrc/ed26l.h): prep, the two shared inversions, the ladder
    ("capi_ED25519F.o", "k_ed25519_lad("), ("capi_ED25519F.o", "k_edlad_prep<"), ("capi_ED25519F.o", "k_fe_batch_div<"),
    ("capi_ED448F.o", "k_ed448_lad("), ("capi_ED448F.o", "k_edlad_prep<"), ("capi_ED448F.o", "k_fe_batch_div<"), ("capi_ED448G.o", "k_fe_batch_div<"),
    # round 5: one scalar per lane on the fixed-base tables of P-256 / secp256k1, and the shared inversion + export behind the Weierstrass kernels
    ("capi_NIST256G.o", "k_nist256_mulgen("), ("capi_SECP256K1G.o", "k_secp256k1_mulgen("),
    ("capi_NIST256F.o", "k_wn_export<"), ("capi_SECP256K1F.o", "k_wn_export<"), ("capi_NIST256G.o", "k_wn_export<"), ("capi_SECP256K1G.o", "k_wn_export<"),
]


def disassemble(obj):
    """{mangled symbol: [(address, instruction text)]} of the gfx950 code object inside a host object"""
    with tempfile.TemporaryDirectory() as tmp:
        fb, co = os.path.join(tmp, "fb.bin"), os.path.join(tmp, "k.co")
        if subprocess.run([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fb], capture_output=True).returncode:
            return {}
        if subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fb, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co],
                          capture_output=True).returncode:
            return {}
        text = subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
    funcs, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = funcs.setdefault(m.group(1), [])
            continue
        if cur is not None and line.startswith("\t"):
            body, _, cm = line.partition("//")
            m = re.match(r"\s*([0-9A-Fa-f]+):", cm)
            cur.append((int(m.group(1), 16) if m else None, body.strip()))
    return funcs


def _regs(tok):
    """registers named by an operand token: 's4' -> {('s',4)}, 's[4:5]' -> {('s',4),('s',5)}, 'v[2:3]' -> {('v',2),('v',3)}, 'vcc' -> {('vcc',0)}"""
    tok = tok.strip().rstrip(",")
    m = re.match(r"^([sv])\[(\d+):(\d+)\]$", tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r"^([sv])(\d+)$", tok)
    if m:
        return {(m.group(1), int(m.group(2)))}
    if tok in ("vcc", "vcc_lo", "vcc_hi"):
        return {("vcc", 0)}
    if tok in EXEC:
        return {("exec", 0)}
    return set()


def _split(ins):
    parts = ins.split(None, 1)
    ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
    return parts[0], ops


# scalar instructions that do NOT write SCC (everything else in the SALU does: compares, arithmetic, logic, shifts)
NOT_SCC = ("s_cselect", "s_mul_i32", "s_mul_hi", "s_mov", "s_load", "s_waitcnt", "s_nop", "s_cbranch", "s_branch", "s_buffer", "s_movk", "s_cmov", "s_barrier", "s_endpgm", "s_sleep",
           "s_setprio", "s_getpc", "s_setpc", "s_swappc", "s_sext", "s_pack", "s_brev", "s_bcnt", "s_ff", "s_flbit", "s_bitset", "s_getreg", "s_setreg", "s_memtime", "s_memrealtime",
           "s_dcache", "s_icache", "s_store", "s_scratch", "s_atc", "s_code_end", "s_trap", "s_sendmsg", "s_inst_prefetch", "s_clause", "s_version", "s_ttrace", "s_wakeup", "s_sethalt",
           "s_set_gpr", "s_rfe", "s_incperf", "s_decperf")
CARRY_OUT = ("v_add_co", "v_sub_co", "v_subrev_co", "v_addc_co", "v_subb_co", "v_subbrev_co", "v_mad_u64_u32", "v_mad_i64_i32", "v_div_scale")
USES_SCC = ("s_cselect", "s_cmov", "s_addc", "s_subb")
UNIFORM_SOURCE = ("s_load", "s_buffer_load", "s_getpc", "s_memtime", "s_memrealtime", "s_getreg", "s_movk")
EXEC = ("exec", "exec_lo", "exec_hi")


class Function:
    """instruction list with a control-flow graph (from the branch offsets), for backward tracing of condition registers"""

    def __init__(self, ins):
        self.addr = [a for a, _ in ins]
        self.text = [t for _, t in ins]
        self.n = len(ins)
        index = {a: i for i, a in enumerate(self.addr) if a is not None}
        self.preds = [[] for _ in range(self.n)]
        for i, t in enumerate(self.text):
            op, ops = _split(t)
            falls = not op.startswith(("s_branch", "s_endpgm", "s_setpc"))
            if falls and i + 1 < self.n:
                self.preds[i + 1].append(i)
            if op.startswith(("s_branch", "s_cbranch_scc", "s_cbranch_vcc", "s_cbranch_exec")) and ops and self.addr[i] is not None:
                try:
                    off = int(ops[0], 0)
                except ValueError:
                    continue
                if off >= 32768:
                    off -= 65536
                j = index.get(self.addr[i] + 4 + 4 * off)
                if j is not None:
                    self.preds[j].append(i)

    def writes(self, i):
        """(registers written by instruction i, kind): kind in valu / load / uniform-source / salu / other"""
        op, ops = _split(self.text[i])
        dst = _regs(ops[0]) if ops else set()
        if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load", "ds_read", "ds_bpermute", "ds_permute", "ds_swizzle", "global_atomic", "buffer_atomic", "ds_add_rtn",
                          "ds_consume", "ds_append", "tbuffer_load", "image_")):
            return dst, "load"
        if op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "ds_write", "tbuffer_store")):
            return set(), "other"
        if op.startswith("v_"):
            if len(ops) > 1 and op.startswith(CARRY_OUT):
                dst |= _regs(ops[1])
            if op.startswith("v_cmp") and (op.endswith("_e32") or (ops and ops[0] in ("vcc",))):
                dst |= {("vcc", 0)}
            if op.startswith("v_cmpx"):
                dst |= {("exec", 0)}
            return dst, "valu"
        if op.startswith("s_"):
            if op.startswith(("s_cmp", "s_bitcmp")):
                return {("scc", 0)}, "salu"
            if not op.startswith(NOT_SCC):
                dst = dst | {("scc", 0)}
            if "saveexec" in op:
                dst = dst | {("exec", 0)}                     # s_and_saveexec_b64 sdst, ssrc: sdst = EXEC, EXEC = ssrc & EXEC
            if op.startswith(("s_cbranch", "s_branch", "s_waitcnt", "s_nop", "s_endpgm", "s_barrier", "s_setprio", "s_sleep")):
                return set(), "other"
            return dst, ("uniform-source" if op.startswith(UNIFORM_SOURCE) else "salu")
        return set(), "other"

    def sources(self, i, hit=None):
        """registers instruction i reads to produce the registers `hit` (None: any of its results)"""
        op, ops = _split(self.text[i])
        srcs = set()
        if "saveexec" in op and hit is not None and ("exec", 0) not in hit:
            return {("exec", 0)}                             # only the saved copy is wanted: sdst = the OLD mask, whatever ssrc holds
        if op == "s_or_b64" and len(ops) == 3 and ops[0] in EXEC and ops[1] in EXEC:
            return _regs(ops[2])                             # end of a structured region: EXEC |= the mask saved at its head, of which the
                                                             # current EXEC is a subset -- the result IS the saved mask
        first = 0 if op.startswith(("s_cmp", "s_bitcmp")) else 1
        # (llvm-objdump prints the destination of every v_cmp first -- "v_cmp_ge_u64_e32 vcc, s[12:13], v[2:3]" -- so sources start at 1 there too)
        writes_exec = "saveexec" in op or op.startswith("v_cmpx") or (ops and ops[0].strip() in EXEC)
        for o in ops[first:]:
            if o.strip() in EXEC and not writes_exec:
                continue                                     # EXEC read as a value (ballots, v_cndmask masks): the launched lanes
            srcs |= _regs(o)
        if "saveexec" in op or op.startswith("v_cmpx"):
            srcs |= {("exec", 0)}                            # the new mask narrows the old one
        if op.startswith("v_") and len(ops) > 1 and op.startswith(CARRY_OUT):
            srcs -= _regs(ops[1])                            # the carry-out operand is a destination
        if op.startswith(USES_SCC):
            srcs |= {("scc", 0)}
        return srcs

    RANK = {"uniform": 0, "lane-index": 1, "unknown": 2, "lane-data": 3}

    def trace(self, start, regs, depth=0, budget=None):
        """what `regs` hold on entry to instruction `start`, over all paths:
        'uniform' (scalar constants, arguments, counters), 'lane-index' (functions of the lane / workitem id and uniform values only),
        'lane-data' (something loaded from memory per lane) or 'unknown'"""
        if not regs:
            return "uniform"
        key = (start, frozenset(regs))
        active = self.__dict__.setdefault("active", set())
        if key in active:
            return "uniform"                                 # being traced further up the recursion: a loop-carried value adds nothing new
        if depth > 80:
            return "unknown"
        active.add(key)                                      # (no caching of verdicts: one reached through an open cycle is provisional)
        try:
            return self._trace(start, regs, depth, budget)
        finally:
            active.discard(key)

    def _trace(self, start, regs, depth, budget):
        budget = budget if budget is not None else [600000]
        seen = set()
        work = [(p, frozenset(regs)) for p in self.preds[start]]
        verdict = "uniform"
        if not self.preds[start] and any(r[0] == "v" for r in regs):
            verdict = "lane-index"

        def worse(a, b):
            return a if self.RANK[a] >= self.RANK[b] else b
        while work:
            i, pend = work.pop()
            if (i, pend) in seen:
                continue
            seen.add((i, pend))
            budget[0] -= 1
            if budget[0] < 0:
                return worse(verdict, "unknown")
            dst, kind = self.writes(i)
            hit = dst & pend
            if hit:
                if kind == "load":
                    return "lane-data"                       # per-lane contents of memory
                if kind in ("valu", "salu"):
                    op = self.text[i].split()[0]
                    if op.startswith("v_mbcnt"):
                        r = "lane-index"
                    else:
                        r = self.trace(i, self.sources(i, hit), depth + 1, budget)
                    if r == "lane-data":
                        return r
                    verdict = worse(verdict, r)
                elif kind != "uniform-source":
                    verdict = worse(verdict, "unknown")
                pend = pend - dst
                if not pend:
                    continue
            if not self.preds[i]:
                # function entry reached with registers never written: SGPRs are kernel arguments / launch constants (uniform),
                # VGPRs are the workitem ids the hardware provides (lane-index)
                if any(r[0] == "v" for r in pend):
                    verdict = worse(verdict, "lane-index")
            for p in self.preds[i]:
                work.append((p, pend))
        return verdict


def audit_function(ins):
    """ins: [(address, text)] or [text] (hand-made streams: addresses are synthesised, 4 bytes per instruction)"""
    if ins and not isinstance(ins[0], tuple):
        ins = [(4 * i, t) for i, t in enumerate(ins)]
    f = Function(ins)
    out = {"scc_uniform": 0, "scc_lane_data": 0, "vcc_uniform": 0, "vcc_lane_data": 0, "lane_index": 0, "exec": 0, "exec_lane_data": 0, "unknown": 0, "calls": 0, "detail": []}
    for i, t in enumerate(f.text):
        op, ops = _split(t)
        if op in ("s_cbranch_scc0", "s_cbranch_scc1"):
            c = f.trace(i, {("scc", 0)})
            key = {"uniform": "scc_uniform", "lane-data": "scc_lane_data", "lane-index": "lane_index"}.get(c, "unknown")
        elif op in ("s_cbranch_vccz", "s_cbranch_vccnz"):
            c = f.trace(i, {("vcc", 0)})
            key = {"uniform": "vcc_uniform", "lane-data": "vcc_lane_data", "lane-index": "lane_index"}.get(c, "unknown")
        elif op in ("s_cbranch_execz", "s_cbranch_execnz"):
            # the mask tested: traced through s_and_saveexec / s_and / s_or / s_mov on EXEC to the v_cmp results that formed it
            c = f.trace(i, {("exec", 0)})
            key = "exec"
            if c == "lane-data":
                out["exec_lane_data"] += 1
            elif c == "unknown":
                out["unknown"] += 1
            c = "exec mask from " + c
        elif op.startswith(("s_setpc", "s_swappc")):
            # calls of out-of-line functions and their returns: the target is s_getpc + constant or the saved return address
            srcs = set()
            for o in ops[-1:]:
                srcs |= _regs(o)
            c = "call/return" if op.startswith("s_setpc") and f.trace(i, srcs) != "lane-data" or op.startswith("s_swappc") and f.trace(i, srcs) != "lane-data" else "lane-data"
            key = "calls" if c == "call/return" else "unknown"
        elif op.startswith(("s_cbranch_i_fork", "s_cbranch_g_fork", "s_cbranch_join")):
            c, key = "fork", "unknown"
        else:
            continue
        out[key] += 1
        out["detail"].append("%5d %s -> %s" % (i, t, c))
    return out


def demangle(names):
    if not names:
        return []
    return subprocess.run(["c++filt"] + list(names), capture_output=True, text=True).stdout.splitlines()


def run(verbose=False):
    bdir = os.path.join(ROOT, "modarith_amd", "build")
    objs = sorted(os.listdir(bdir)) if os.path.isdir(bdir) else []
    allow = json.load(open(ALLOW))["kernels"] if os.path.exists(ALLOW) else []
    rows, problems = [], []
    cache = {}
    todo = []
    for og, pat in AUDITED:
        for o in fnmatch.filter(objs, og):
            if o not in cache:
                funcs = disassemble(os.path.join(bdir, o))
                syms = list(funcs)
                cache[o] = (funcs, dict(zip(syms, demangle(syms))))
            funcs, names = cache[o]
            for sym, ins in funcs.items():
                name = re.sub(r"^void ", "", names.get(sym, sym))
                if pat not in name or not ins:
                    continue
                todo.append((o, name, ins))
    # the register tracing is pure Python and takes a second or two per kernel: one worker per core
    import multiprocessing
    with multiprocessing.Pool(min(8, os.cpu_count() or 1)) as pool:
        audits = pool.map(audit_function, [t[2] for t in todo], chunksize=1)
    for (o, name, ins), a in zip(todo, audits):
        short = re.sub(r"\(.*", "", name)
        entry = next((e for e in allow if e["match"] in short and short.endswith(e.get("ends", ""))), None)
        row = {"object": o, "kernel": short, **{k: a[k] for k in a if k != "detail"}}
        rows.append(row)
        if verbose:
            print(short)
            for d in a["detail"]:
                print("    " + d)
        if entry is None:
            problems.append("%s: no allow-list entry" % short)
            continue
        lane = a["scc_lane_data"] + a["vcc_lane_data"] + a["exec_lane_data"]
        if lane > entry.get("lane_data_branches", 0):
            problems.append("%s: %d data-dependent branch(es), %d allowed" % (short, lane, entry.get("lane_data_branches", 0)))
        if a["exec"] + a["lane_index"] > entry.get("exec_branches", 0):
            problems.append("%s: %d exec-mask / lane-index branch(es), %d allowed" % (short, a["exec"] + a["lane_index"], entry.get("exec_branches", 0)))
        if a["unknown"] > entry.get("unknown", 0):
            problems.append("%s: %d unclassified branch(es)" % (short, a["unknown"]))
    return rows, problems


def survey():
    """--all: every kernel of every built object that holds a branch on lane data (report only).  Outside the audited set these are the
    documented contract votes of the `Auto` policies (kernels.h: "is any limb of this wave beyond the contract?" -- false for every
    output of a field function) and the zero test of the simultaneous inversion; the `_ct` entry points do not take them."""
    bdir = os.path.join(ROOT, "modarith_amd", "build")
    total, flagged = 0, {}
    for o in sorted(os.listdir(bdir)):
        if not o.endswith(".o"):
            continue
        funcs = disassemble(os.path.join(bdir, o))
        syms = list(funcs)
        names = dict(zip(syms, demangle(syms)))
        for sym, ins in funcs.items():
            if not ins:
                continue
            total += 1
            a = audit_function(ins)
            lane = a["scc_lane_data"] + a["vcc_lane_data"] + a["exec_lane_data"]
            if lane or a["unknown"]:
                short = re.sub(r"\(.*", "", re.sub(r"^void ", "", names.get(sym, sym)))
                family = re.sub(r"P_\w+|C_\w+", "*", short)
                f = flagged.setdefault(family, {"kernels": 0, "lane_data_branches": 0, "unknown": 0, "example": short + " [" + o + "]"})
                f["kernels"] += 1
                f["lane_data_branches"] = max(f["lane_data_branches"], lane)
                f["unknown"] = max(f["unknown"], a["unknown"])
    print("%d kernels in %s; families with a branch on lane data:" % (total, bdir))
    for fam, f in sorted(flagged.items()):
        print("  %4d kernels, up to %d branch(es)%s  %s   e.g. %s" % (f["kernels"], f["lane_data_branches"], (", %d unclassified" % f["unknown"]) if f["unknown"] else "", fam, f["example"]))
    return 0


def main(argv):
    if "--all" in argv:
        return survey()
    rows, problems = run(verbose="--verbose" in argv)
    print("%-4s %-4s %-4s %-4s %-4s %-4s %-4s %-4s %-4s  %s" % ("sccU", "sccD", "vccU", "vccD", "idx", "exec", "excD", "call", "unk", "kernel [object]"))
    for r in rows:
        print("%4d %4d %4d %4d %4d %4d %4d %4d %4d  %s [%s]" % (r["scc_uniform"], r["scc_lane_data"], r["vcc_uniform"], r["vcc_lane_data"], r["lane_index"], r["exec"], r["exec_lane_data"], r["calls"], r["unknown"],
                                                         r["kernel"][:110], r["object"]))
    if "--json" in argv:
        with open(argv[argv.index("--json") + 1], "w") as f:
            json.dump({"kernels": rows, "problems": problems}, f, indent=1)
    for p in problems:
        print("PROBLEM:", p)
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
