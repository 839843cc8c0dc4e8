"""Build libmodarith_amd.so (HIP kernels + C-ABI shim) for gfx950, in-tree.

  python -m modarith_amd.build [--force]

Steps: (1) the parameter driver emits csrc/generated/params_<PRIME>.h (modarith_amd.emit),
(2) hipcc compiles one translation unit per prime plus the common one, in parallel,
(3) hipcc links modarith_amd/libmodarith_amd.so.  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import subprocess
import time
import sys

from . import emit

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libmodarith_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
# -amdgpu-codegenprepare-mul24=0: the IR-level 24-bit-multiply formation of this compiler miscompiles the fused split-product
# chains of C2065 (4 x 52-bit limbs; wrong for every lane, right at -O0, right with this switch; found in round 3 by comparing the two product policies lane by lane);
# the DAG-level mul24 selection stays on.  Measured cost on the VALU-bound kernels: see DESIGN.md.
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-mllvm", "-amdgpu-codegenprepare-mul24=0"]
# (source unit, object name, extra flags).  Every curve unit is compiled three times (MA_CURVE_PART, capi_curve.inc):
# its two scalar-multiplication kernels take minutes each for the 7- and 9-limb fields and go to separate jobs.
_CURVE_UNITS = ["capi_%s" % c for c in emit.BUILT_CURVES] + ["capi_%sW" % c for c in emit.BUILT_WCURVES]
UNITS = ([(u, u, []) for u in ["capi_common", "capi_ED25519F", "capi_ED25519F2", "capi_ED448F", "capi_ED448F2", "capi_NIST256F", "capi_NIST256F2", "capi_SECP256K1F", "capi_SECP256K1F2", "capi_NIST256G", "capi_SECP256K1G", "capi_ED25519G", "capi_ED448G"] + ["capi_%s" % p for p in emit.CORE_PRIMES]]
         + [(u, "%s_part%d" % (u, part), ["-DMA_CURVE_PART=%d" % part]) for u in _CURVE_UNITS for part in (1, 2, 3)]
         + [("generated/capi_%s" % p, "capi_%s" % p, []) for p in emit.EXTRA_PRIMES])
# longest first, so the pool does not finish on a long tail: measured compile seconds of the slow units (8 jobs on 8 cores; the
# rest take 10-30 s each)
_COST = {"capi_SIDH751": 180, "capi_NIST521W_part2": 170, "capi_ED500_part2": 143, "capi_SIDH610": 105, "capi_NIST521W_part1": 91, "capi_CSIDH512": 90,
         "capi_ED448G": 80, "capi_NIST384": 61, "capi_X448": 47, "capi_ED448F": 43, "capi_SIDH503": 37, "capi_ED448F2": 36,
         "capi_ED500_part3": 37, "capi_NIST384W_part2": 35, "capi_NIST521W_part3": 32, "capi_ED500_part1": 27, "capi_NIST256G": 24}
UNITS.sort(key=lambda t: -_COST.get(t[1], 25 if ("-DMA_CURVE_PART=1" in t[2] or "-DMA_CURVE_PART=2" in t[2]) else 15))


def _stamp() -> str:
    h = hashlib.sha256()
    for root, _, files in sorted(os.walk(CSRC)):
        for f in sorted(files):
            h.update(f.encode())
            h.update(open(os.path.join(root, f), "rb").read())
    h.update(open(os.path.join(os.path.dirname(HERE), "include", "modarith_amd.h"), "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


_TIMES = {}
_ROOT = os.path.dirname(HERE)


def _unit_deps(obj: str):
    """in-tree files the unit's last compile read (from its -MD depfile)"""
    dep = obj[:-2] + ".d"
    if not os.path.exists(dep):
        return []
    text = open(dep).read().replace("\\\n", " ")
    return [f for f in sorted(set(text.split()[1:])) if os.path.abspath(f).startswith(_ROOT + os.sep)]


def _unit_hash(obj: str, cmd) -> str | None:
    """hash of the command line and of every in-tree file the unit's last compile read (from its -MD depfile);
    None when the depfile is missing or names a file that no longer exists"""
    dep = obj[:-2] + ".d"
    if not (os.path.exists(dep) and os.path.exists(obj)):
        return None
    # paths enter the hash RELATIVE to the repository root: the same sources built in another directory (a clone, the driver's scratch
    # copy) get the same unit hashes, which profiles/<tag>_valu_pmc.json records to say which build its counters belong to
    rel = lambda f: os.path.relpath(os.path.abspath(f), _ROOT)
    h = hashlib.sha256(" ".join(rel(c) if os.path.isabs(c) and os.path.abspath(c).startswith(_ROOT + os.sep) else c for c in cmd).encode())
    text = open(dep).read().replace("\\\n", " ")
    for f in sorted(set(text.split()[1:]), key=rel):
        if not os.path.abspath(f).startswith(_ROOT + os.sep):
            continue                      # toolchain headers: covered by the hipcc path in the command line
        if not os.path.exists(f):
            return None
        h.update(rel(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def _compile(unit) -> str:
    """compile one unit unless its object is up to date (same command, same contents of every included in-tree file)"""
    src_unit, obj_name, extra = unit
    src = os.path.join(CSRC, src_unit + ".hip")
    obj = os.path.join(OBJ, obj_name + ".o")
    cmd = [HIPCC] + FLAGS + extra + ["-c", src]
    hfile = obj[:-2] + ".hash"
    if os.path.exists(hfile) and open(hfile).read() == (_unit_hash(obj, cmd) or "-"):
        return obj
    t0 = time.time()
    subprocess.run(cmd + ["-MD", "-MF", obj[:-2] + ".d", "-o", obj], check=True, timeout=int(os.environ.get("MA_BUILD_TIMEOUT", "1500")))
    _TIMES[obj_name] = time.time() - t0
    # a source edited WHILE the unit compiled would be hashed in its new state against an object made from the old one: such a
    # unit gets no hash and is compiled again by the next build
    stale = any(os.path.exists(f) and os.path.getmtime(f) > t0 for f in _unit_deps(obj))
    with open(hfile, "w") as f:
        f.write("-" if stale else (_unit_hash(obj, cmd) or "-"))
    return obj


UNIT_HASHES = os.path.join(HERE, "unit_hashes.json")


def _write_unit_hashes() -> None:
    """modarith_amd/unit_hashes.json: per unit, 16 hex digits of the hash of its command line and of every in-tree file it was compiled
    from.  It travels with the library (the objects do not), so that a counter summary under profiles/ can say which build it was taken
    on and bench.py can tell when the kernels of a leg have changed since (valu_roofline: source_stale)."""
    import json
    out = {}
    for _, obj_name, _ in UNITS:
        hfile = os.path.join(OBJ, obj_name + ".hash")
        if os.path.exists(hfile):
            out[obj_name] = open(hfile).read()[:16]
    with open(UNIT_HASHES, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)


def unit_hashes() -> dict:
    import json
    return json.load(open(UNIT_HASHES)) if os.path.exists(UNIT_HASHES) else {}


def build(force: bool = False, verbose: bool = True) -> str:
    emit.emit_all()
    os.makedirs(OBJ, exist_ok=True)
    stamp_file = os.path.join(OBJ, "stamp")
    stamp = _stamp()
    if not force and os.path.exists(LIB) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        if not os.path.exists(UNIT_HASHES):
            _write_unit_hashes()
        return LIB
    if force:
        for f in os.listdir(OBJ):
            if f.endswith(".hash"):
                os.remove(os.path.join(OBJ, f))
    if verbose:
        print("[modarith_amd] building %d HIP units for %s (unchanged units are reused) ..." % (len(UNITS), ARCH), flush=True)
    with cf.ThreadPoolExecutor(max_workers=min(int(os.environ.get('MA_BUILD_JOBS', '8')), len(UNITS))) as ex:
        objs = list(ex.map(_compile, UNITS))
    subprocess.check_call([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs)
    with open(stamp_file, "w") as f:
        f.write(stamp)
    _write_unit_hashes()
    if verbose:
        slow = sorted(_TIMES.items(), key=lambda kv: -kv[1])[:6]
        print("[modarith_amd] compiled %d units; slowest: " % len(_TIMES) + ", ".join("%s %.0f s" % kv for kv in slow), flush=True)
        print("[modarith_amd] built", LIB, flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
