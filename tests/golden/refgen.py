"""Drive the reference generators (pseudo.py / monty.py) in-process, in THIS container only.

The reference has no importable API: all work happens at module scope and it shells out to the
external `addchain` tool (absent from this image) at pseudo.py:1582-1587 / monty.py:2166-2171.
We therefore execute the reference's top-level statements one by one from its own source, skip
exactly the statements that invoke `addchain`, stop before the ctypes self-test block
(pseudo.py:1693, monty.py:2268), and then call the reference's own emitter functions
(prop, flat, modfsb, modadd, ... modimp) to obtain the C text it would have written to test.c.
`modpro` (the only consumer of addchain's output, pseudo.py:758-785) and its callers
modinv/modqr/modsqrt are NOT emitted; they are pinned mathematically after redc instead.

Nothing here ships: outputs go to a scratch dir; only vectors (tests/golden/*.json) are committed.
This module is test tooling and must never be imported by the product or on the GPU box.
"""
import ast, io, os, sys, subprocess, tempfile, contextlib, inspect, ctypes

REF = "/root/reference"

# emitters that need ac.txt (addchain output) directly or through modpro
_NEEDS_CHAIN = {"modpro", "modinv", "modqr", "modsqrt"}

_ORDER = ["prop", "flat", "modfsb", "modadd", "modsub", "modneg", "modmli", "modmul", "modsqr",
          "modcpy", "modnsqr", "nres", "redc", "modis1", "modis0", "modzer", "modone", "modint",
          "modcmv", "modcsw", "modshl", "modshr", "modhaf", "mod2r", "modexp", "modimp",
          "modsign", "modcmp"]


def load(script, wl, prime, overrides=None):
    """Run `script` (pseudo.py|monty.py) up to the end of its parameter block; return namespace."""
    overrides = overrides or {}
    path = os.path.join(REF, script)
    src = open(path).read()
    tree = ast.parse(src)
    ns = {"__name__": "__refgen__", "__file__": path}
    old_argv, old_cwd = sys.argv, os.getcwd()
    sys.argv = [script, str(wl), prime]
    scratch = tempfile.mkdtemp(prefix="refgen_")
    os.chdir(scratch)
    log = io.StringIO()
    try:
        for node in tree.body:
            seg = ast.get_source_segment(src, node) or ""
            if isinstance(node, ast.With) and "test.c" in seg:
                break  # self-test / file emission block: not executed
            if not isinstance(node, ast.FunctionDef) and (
                    "addchain" in seg or "inv.acc" in seg or "remove_unused" in seg
                    or "cline" in seg):
                continue  # external tool absent; only modpro consumes its output
            with contextlib.redirect_stdout(log):
                exec(compile(ast.Module([node], []), path, "exec"), ns)
            if isinstance(node, ast.Assign):
                for t in node.targets:
                    if isinstance(t, ast.Name) and t.id in overrides:
                        ns[t.id] = overrides[t.id]
    finally:
        sys.argv = old_argv
        os.chdir(old_cwd)
    ns["_log"] = log.getvalue()
    ns["_scratch"] = scratch
    ns["_argv"] = [script, str(wl), prime]
    return ns


def emit_c(ns, makestatic=False):
    """C text of header + every emitter that does not need the addition chain."""
    ns["makestatic"] = makestatic
    ns["DECOR"] = ""
    out = io.StringIO()
    old_argv, sys.argv = sys.argv, ns["_argv"]  # header() prints the command line
    with contextlib.redirect_stdout(out):
        ns["header"]()
        for name in _ORDER:
            fn = ns[name]
            ar = len(inspect.signature(fn).parameters)
            args = [ns["n"], ns.get("m", 0)][:ar]
            print(fn(*args))
    sys.argv = old_argv
    return out.getvalue()


def build(ns, extra_c="", tag="ref"):
    """Compile the emitted C (+ optional harness code appended) to a shared object; return CDLL."""
    csrc = emit_c(ns) + "\n" + extra_c
    d = ns["_scratch"]
    cpath = os.path.join(d, tag + ".c")
    so = os.path.join(d, tag + ".so")
    open(cpath, "w").write(csrc)
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-w", "-o", so, cpath])
    return ctypes.CDLL(so), csrc
