"""Generator mode: a field for ANY prime the reference generators accept, built at run time.

  python -m modarith_amd.generate 64 2**251-9               # the reference's own command-line shape
  python -m modarith_amd.generate 64 BP256=0xa9fb57db...5377 --monty
  python -m modarith_amd.generate 64 2**251-9 --time        # ... and run the time.c protocol on the GPU, as the generators do last
  python -m modarith_amd.generate curve NIST224 weierstrass NIST224 -3 0xb405...ffb4 0xffff...2a3d 0xb70e...1d21 0xbd37...7e34
  python -m modarith_amd.generate --list

This is the counterpart of `python pseudo.py 64 <prime>` / `python monty.py 64 <prime>` (pseudo.py:1461-1473,
1552-1566; monty.py:2111-2135): where the reference writes a specialised field.c for the prime, this driver derives
the same constants (modarith_amd.params), emits them as a `struct P_<TAG>` (modarith_amd.emit.header_text), and has
hipcc instantiate the hand-written kernels of csrc/field.h + csrc/kernels.h for it -- one translation unit, about
ten seconds -- into a plug-in `modarith_amd/plugins/libmodarith_amd_<TAG>.so` that exports the same C-ABI as a
built-in prime: `<fn>_<TAG>_ct` (host pointers, the reference's signatures) and `<fn>_<TAG>_batch` (device
pointers), declared by `MODARITH_AMD_DECLARE(<TAG>)` of include/modarith_amd.h.  The plug-in links against
libmodarith_amd.so (launch geometry, error text, staging buffers); `Field("<TAG>")` loads it.

Naming follows the generators' decoration rule (pseudo.py:1940-1944, monty.py:2510-2520): a named prime keeps its
name; an unnamed pseudo-Mersenne 2^n - m is tagged `<n><m>` ("25519"); any other unnamed modulus must be given a
name (`NAME=<expression>` or `name=`), as monty.py insists ("Modulus must have a name").  Only 64-bit words are
built: the MI355X kernels hold u64 limbs (SURVEY 8 sizes); 16 / 32 are refused with the reason.

There is no CPU path here either: the plug-in contains GPU kernels only, and a missing hipcc is an error.
"""
from __future__ import annotations

import hashlib
import json
import os
import re
import subprocess
import sys
from dataclasses import dataclass
from typing import List, Optional

from . import emit
from .params import NAMED, FieldParams, derive_monty, derive_pseudo

HERE = os.path.dirname(os.path.abspath(__file__))
PLUGIN_DIR = os.environ.get("MA_PLUGIN_DIR", os.path.join(HERE, "plugins"))
_TAG_RE = re.compile(r"^[A-Za-z0-9][A-Za-z0-9_]*$")


# unnamed moduli the test-suite generates (tests/golden/field_<TAG>.json hold what the reference generators emit for them):
# (argument as on the reference's command line, family) -- an unnamed pseudo-Mersenne, a three-limb one, a general 256-bit
# prime (brainpoolP256r1) in full Montgomery form, and monty.py's PM shortcut for an unnamed 2^n - m
EXAMPLES = (("2**251-9", "pseudo"), ("2**130-5", "pseudo"),
            ("BP256=0xa9fb57dba1eea9bc3e660a909d838d726e3bf623d52620282013481d1f6e5377", "monty"), ("M2519=2**251-9", "monty"))


class GenerateError(ValueError):
    pass


@dataclass
class Generated:
    tag: str                    # the <TAG> of the exported symbols
    lib: str                    # path of the plug-in shared object
    params: FieldParams
    built: bool                 # False when an up-to-date plug-in was reused


def _evaluate(expr: str) -> int:
    """the modulus of a command-line argument: an expression that starts with a digit (pseudo.py:1554-1556), or
    "00" + decimal for a group order (monty.py:2116-2118)"""
    if not expr or not expr[0].isdigit():
        raise GenerateError("%r: an unnamed modulus is a python expression that starts with a digit, e.g. 2**255-19" % (expr,))
    if expr.startswith("00"):
        return int(expr)
    if not re.fullmatch(r"[0-9a-fA-FxX*+\-() ]+", expr):
        raise GenerateError("%r: only integers, + - * ** and parentheses are evaluated" % (expr,))
    return int(eval(expr, {"__builtins__": {}}))


def resolve(prime: str, family: Optional[str] = None, name: Optional[str] = None, radix: Optional[int] = None) -> FieldParams:
    """prime (a name of modarith_amd.params.NAMED, an expression, or NAME=expression) -> FieldParams with .name = TAG"""
    if "=" in prime and name is None:
        name, prime = prime.split("=", 1)
    if prime in NAMED:
        p, fam = NAMED[prime]
        name = name or prime
        family = family or fam
    else:
        p = _evaluate(prime)
    n = p.bit_length()
    fp = None
    if family in (None, "pseudo"):
        try:
            fp = derive_pseudo(name or "_", p, radix)
        except ValueError as e:
            if family == "pseudo":
                raise GenerateError("%s (pseudo.py:1563-1592)" % e) from None
    if fp is None:
        try:
            fp = derive_monty(name or "_", p, radix)
        except ValueError as e:
            raise GenerateError("%s (monty.py:2131-2230)" % e) from None
    if name is None:
        if fp.family == "pseudo" or fp.pm:
            name = "%d%d" % (n, (1 << n) - p)                   # the generators' own tag for an unnamed 2^n - m
        else:
            raise GenerateError("Modulus must have a name - unable to make one for you (monty.py:2517-2519): pass NAME=<expression>")
    if not _TAG_RE.match(name):
        raise GenerateError("%r cannot be part of a C identifier" % (name,))
    fp.name = name
    return fp


def _flags() -> List[str]:
    from .build import FLAGS
    return list(FLAGS) + ["-I", os.path.join(HERE, "csrc", "generated"), "-I", os.path.join(HERE, "csrc")]


def _key(fp: FieldParams, tag: str) -> str:
    """what a plug-in was made from: the constants, the flags and every kernel source (path-independent, so a plug-in built in one
    checkout is recognised as current in a copy of it)"""
    from .build import FLAGS, _stamp
    h = hashlib.sha256((" ".join(FLAGS) + "\n" + emit.header_text(fp) + "\n" + emit.capi_unit_text(tag) + "\n" + _stamp()).encode())
    return h.hexdigest()


def plugin_path(tag: str, plugin_dir: Optional[str] = None) -> str:
    return os.path.join(plugin_dir or PLUGIN_DIR, "libmodarith_amd_%s.so" % tag)


def generate(prime: str, wl: int = 64, family: Optional[str] = None, name: Optional[str] = None, radix: Optional[int] = None,
             plugin_dir: Optional[str] = None, force: bool = False, verbose: bool = False) -> Generated:
    """derive the constants of `prime`, emit them, compile the kernels for it; returns the plug-in to load.
    An existing plug-in is reused when neither the constants nor any kernel source it was compiled from have changed."""
    if wl != 64:
        raise GenerateError("only 64-bit words are built for the GPU (u64 limbs, 128-bit column sums); the reference's 16- and "
                            "32-bit forms have no counterpart here")
    fp = resolve(prime, family, name, radix)
    tag = fp.name
    from . import _lib
    if tag in _lib.PRIMES and NAMED.get(tag, (None,))[0] == fp.p and (family is None or NAMED[tag][1] == fp.family) and radix is None:
        return Generated(tag, _lib.LIB_PATH, fp, False)         # a built-in prime: nothing to generate
    if tag in _lib.PRIMES:
        raise GenerateError("%s names a built-in field with other constants; choose another name" % tag)
    if fp.nlimbs > emit.MAX_GENERATED_LIMBS:
        raise GenerateError("%d limbs: the kernels keep every operand in registers and are built for at most %d limbs" % (fp.nlimbs, emit.MAX_GENERATED_LIMBS))
    d = plugin_dir or PLUGIN_DIR
    os.makedirs(d, exist_ok=True)
    hdr, unit = os.path.join(d, "params_%s.h" % tag), os.path.join(d, "capi_%s.hip" % tag)
    obj, lib, meta = os.path.join(d, "capi_%s.o" % tag), plugin_path(tag, d), os.path.join(d, "%s.json" % tag)
    key = _key(fp, tag)
    emit._write(hdr, emit.header_text(fp))
    # the paste-marker shim of this field, next to its plug-in: what a consumer includes where the reference says "paste field.c here"
    emit._write(os.path.join(d, "field_%s.h" % tag), emit.field_shim_text(fp, tag))
    emit._write(unit, emit.capi_unit_text(tag).replace('"../capi_prime.inc"', '"capi_prime.inc"'))
    from .build import ARCH, HIPCC
    cmd = [HIPCC] + _flags() + ["-c", unit]
    if not force and os.path.exists(lib) and os.path.exists(meta):
        try:
            if json.load(open(meta)).get("hash") == key:
                return Generated(tag, lib, fp, False)
        except (ValueError, OSError):
            pass
    if not os.path.exists(HIPCC):
        raise GenerateError("%s not found: generating a field needs the ROCm compiler (there is no CPU path)" % HIPCC)
    main = os.path.join(HERE, "libmodarith_amd.so")
    if not os.path.exists(main):
        raise GenerateError("%s is missing: build it first (python -m modarith_amd.build); plug-ins link against it" % main)
    if verbose:
        print("[modarith_amd] hipcc %s -> %s" % (os.path.basename(unit), os.path.basename(lib)), flush=True)
    # object, library and metadata are written under process-private names and moved into place: another process (a second rank,
    # a parallel test worker) generating or loading the same plug-in never sees a half-written file
    tmp = ".%d.tmp" % os.getpid()
    subprocess.run(cmd + ["-o", obj + tmp], check=True, timeout=int(os.environ.get("MA_BUILD_TIMEOUT", "1500")))
    rel = os.path.relpath(HERE, d)
    subprocess.check_call([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib + tmp, obj + tmp, "-L", HERE, "-l:libmodarith_amd.so",
                           "-Wl,-rpath,$ORIGIN/" + rel, "-Wl,-rpath," + HERE])
    with open(meta + tmp, "w") as f:
        json.dump({"tag": tag, "prime": prime, "p": hex(fp.p), "family": fp.family, "radix": fp.radix, "nlimbs": fp.nlimbs, "hash": key}, f, indent=1)
    os.replace(obj + tmp, obj)
    os.replace(lib + tmp, lib)
    os.replace(meta + tmp, meta)
    return Generated(tag, lib, fp, True)


# ---------------------------------------------------------------------------------------------------------------------
# curves of one's own: the counterpart of curve.py's table ("More curves can be added here", curve.py:73-203)
@dataclass
class GeneratedCurve:
    name: str                   # upper-case name; symbols ecn_<lower>_*
    kind: str                   # "edwards" | "weierstrass"
    field: str                  # tag of the field (built-in or generated)
    lib: str
    nlimbs: int
    nbytes: int
    built: bool


def curve_plugin_path(name: str, plugin_dir: Optional[str] = None) -> str:
    return os.path.join(plugin_dir or PLUGIN_DIR, "libmodarith_amd_curve_%s.so" % name.lower())


def generate_curve(name: str, kind: str, field: str, a: int, b: int, order: int, gx: int, gy: int, cof: int = 0,
                   plugin_dir: Optional[str] = None, force: bool = False, verbose: bool = False) -> GeneratedCurve:
    """The curve layer (curve.h: ecn_<name>_mul / mul2 / add / dbl / set / get ..., scalar and batched) for a curve that is not in
    curve.py's table.  kind "edwards": a x^2 + y^2 = 1 + b x^2 y^2 with a = +-1 (edwards.c; cof = log2 of the cofactor);
    kind "weierstrass": y^2 = x^3 + a x + b with a = -3 or 0 and prime order (weierstrass.c).  `field`: a built-in prime name or the
    tag of a generated field (generate() first).  One hipcc unit (the scalar-multiplication kernels take about a minute to
    compile); the plug-in exports what MODARITH_AMD_DECLARE_EDWARDS(<lower-case name>, Nlimbs) declares."""
    from . import _lib, curves
    from .params import derive
    if kind not in ("edwards", "weierstrass"):
        raise GenerateError("kind must be 'edwards' or 'weierstrass'")
    if not _TAG_RE.match(name) or not name[0].isalpha():
        raise GenerateError("%r cannot be part of a C identifier" % (name,))
    up, low = name.upper(), name.lower()
    if low in _lib.CURVES:
        raise GenerateError("%s is a built-in curve" % up)
    d = plugin_dir or PLUGIN_DIR
    if field in _lib.PRIMES:
        fp = derive(field)
    elif os.path.exists(os.path.join(d, "%s.json" % field)):
        fp = params_of_plugin(field, d)
    elif os.path.exists(os.path.join(PLUGIN_DIR, "%s.json" % field)):
        fp = params_of_plugin(field)
    else:
        raise GenerateError("field %r is neither built in nor generated: run generate() for it first" % (field,))
    p = fp.p
    # the checks curve.py leaves to its user: the generator is on the curve, the order annihilates it
    if kind == "edwards":
        if a not in (1, -1):
            raise GenerateError("edwards.c handles a = 1 and a = -1")
        if gy and (a * gx * gx + gy * gy - 1 - b * gx * gx * gy * gy) % p:
            raise GenerateError("the generator is not on the curve")
        c = curves.EdwardsCurve(up, field, a, b, cof, order, gx, gy, fp)
        hdr_text = emit.curve_header_text_of(c)
        cls, inc = "ma::Edwards<ma::C_%s>" % up, "edwards.h"
    else:
        if a not in (-3, 0):
            raise GenerateError("weierstrass.c handles a = -3 and a = 0")
        if gy and (gy * gy - gx ** 3 - a * gx - b) % p:
            raise GenerateError("the generator is not on the curve")
        c = curves.WeierstrassCurve(up, field, a, b, order, gx, gy, fp)
        hdr_text = emit.wcurve_header_text_of(c)
        cls, inc = "ma::Weierstrass<ma::C_%s>" % up, "weierstrass.h"
    unit_text = ("// GENERATED by modarith_amd/generate.py -- do not edit.  C-ABI of the curve layer for %s (%s over %s); body: csrc/capi_curve.inc\n"
                 '#include "modarith_amd.h"\nextern "C" {\nMODARITH_AMD_DECLARE_EDWARDS(%s, %d)\n}\n#include "curve_%s.h"\n#include "%s"\n'
                 "#define MA_CURVE_CLASS %s\n#define MA_CNAME %s\n#include \"capi_curve.inc\"\n" % (up, kind, field, low, fp.nlimbs, up, inc, cls, low))
    os.makedirs(d, exist_ok=True)
    hdr, unit = os.path.join(d, "curve_%s.h" % up), os.path.join(d, "capi_curve_%s.hip" % up)
    obj, lib, meta = os.path.join(d, "capi_curve_%s.o" % up), curve_plugin_path(up, d), os.path.join(d, "curve_%s.json" % up)
    from .build import ARCH, FLAGS, HIPCC, _stamp
    key = hashlib.sha256((" ".join(FLAGS) + "\n" + hdr_text + "\n" + unit_text + "\n" + emit.header_text(fp) + "\n" + _stamp()).encode()).hexdigest()
    out = GeneratedCurve(up, kind, field, lib, fp.nlimbs, fp.nbytes, False)
    if not force and os.path.exists(lib) and os.path.exists(meta):
        try:
            if json.load(open(meta)).get("hash") == key:
                return out
        except (ValueError, OSError):
            pass
    if not os.path.exists(HIPCC):
        raise GenerateError("%s not found: generating a curve needs the ROCm compiler (there is no CPU path)" % HIPCC)
    emit._write(hdr, hdr_text)
    emit._write(unit, unit_text)
    if verbose:
        print("[modarith_amd] hipcc %s -> %s" % (os.path.basename(unit), os.path.basename(lib)), flush=True)
    tmp = ".%d.tmp" % os.getpid()
    inc_dirs = ["-I", os.path.join(HERE, "csrc", "generated"), "-I", os.path.join(HERE, "csrc"), "-I", os.path.join(os.path.dirname(HERE), "include"), "-I", PLUGIN_DIR, "-I", d]
    subprocess.run([HIPCC] + list(FLAGS) + inc_dirs + ["-c", unit, "-o", obj + tmp], check=True, timeout=int(os.environ.get("MA_BUILD_TIMEOUT", "1500")))
    subprocess.check_call([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib + tmp, obj + tmp, "-L", HERE, "-l:libmodarith_amd.so",
                           "-Wl,-rpath,$ORIGIN/" + os.path.relpath(HERE, d), "-Wl,-rpath," + HERE])
    with open(meta + tmp, "w") as f:
        json.dump({"curve": up, "kind": kind, "field": field, "a": a, "b": hex(b) if b >= 0 else "-" + hex(-b), "order": hex(order), "cof": cof,
                   "gx": hex(gx), "gy": hex(gy), "nlimbs": fp.nlimbs, "nbytes": fp.nbytes, "hash": key}, f, indent=1)
    os.replace(obj + tmp, obj)
    os.replace(lib + tmp, lib)
    os.replace(meta + tmp, meta)
    out.built = True
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Montgomery ladders of one's own: rfc7748.c describes its curve in an #ifdef block (A24, COF, GENERATOR, TWIST_SECURE; rfc7748.c:118-132)
# and is otherwise generic over the pasted field; this is that block for a curve other than X25519 / X448
def ladder_plugin_path(name: str, plugin_dir: Optional[str] = None) -> str:
    return os.path.join(plugin_dir or PLUGIN_DIR, "libmodarith_amd_ladder_%s.so" % name)


def generate_ladder(name: str, field: str, a24: int, cof: int, twist_secure: bool = True, plugin_dir: Optional[str] = None,
                    force: bool = False, verbose: bool = False) -> str:
    """`rfc7748_<name>(bk, bu, bv)` and `rfc7748_<name>_batch(bk, bu, bv, n, stream)` for the Montgomery curve
    v^2 = u^3 + A u^2 + u with (A - 2) / 4 = a24 over a built-in or generated field, cofactor 2^cof (2 or 3): the reference's
    rfc7748() call for call on the bit-exact field (csrc/ladder.h k_rfc7748: clamp, 5 M + 4 S + a24 per bit, generic=False sums,
    modpro + modinv, little-endian Nbytes records).  Only the TWIST_SECURE branch of rfc7748.c:224-227 is built; records move as
    64-bit words, so Nbytes must be a multiple of 8.  Returns the plug-in's path."""
    from . import _lib
    from .params import derive
    if not _TAG_RE.match(name) or name in _lib.LADDERS:
        raise GenerateError("%r: not a usable name (X25519 and X448 are built in)" % (name,))
    if not twist_secure:
        raise GenerateError("only the TWIST_SECURE form of rfc7748() is built (rfc7748.c:224-227); the point-validation branch is not")
    if cof not in (2, 3):
        raise GenerateError("COF is 2 or 3 (rfc7748.c:122)")
    if not 0 < a24 < (1 << 28):
        raise GenerateError("A24 must be a small positive integer: it is the `int` of modmli (rfc7748.c:209)")
    d = plugin_dir or PLUGIN_DIR
    if field in _lib.PRIMES:
        fp = derive(field)
    elif os.path.exists(os.path.join(d, "%s.json" % field)):
        fp = params_of_plugin(field, d)
    elif os.path.exists(os.path.join(PLUGIN_DIR, "%s.json" % field)):
        fp = params_of_plugin(field)
    else:
        raise GenerateError("field %r is neither built in nor generated: run generate() for it first" % (field,))
    if fp.nbytes % 8:
        raise GenerateError("%d-byte records: the ladder kernel moves records as 64-bit words" % fp.nbytes)
    sym = "rfc7748_%s" % name
    unit_text = "\n".join([
        "// GENERATED by modarith_amd/generate.py -- do not edit.  rfc7748() for the Montgomery curve %s: A24 = %d, COF = %d, over %s" % (name, a24, cof, field),
        '#include "params_%s.h"' % field, '#include "modarith_amd.h"', '#include "capi_common.h"', '#include "kernels.h"', '#include "ladder.h"', "",
        "namespace {", "using namespace ma;", "using P = ma::P_%s;" % field, "constexpr int NB = P::NBYTES;", "}", "",
        'extern "C" int %s_batch(const char* bk, const char* bu, char* bv, size_t n, void* st) {' % sym,
        "    if (n == 0) return 0;",
        "    if ((reinterpret_cast<uintptr_t>(bk) | reinterpret_cast<uintptr_t>(bu) | reinterpret_cast<uintptr_t>(bv)) & 7u) {",
        '        set_error("%s: byte records must be 8-byte aligned");' % sym,
        "        return (int)hipErrorInvalidValue;",
        "    }",
        "    const int block = ladder_block();",
        "    k_rfc7748<P, %d, %d><<<grid_for(n, block), block, 0, (hipStream_t)st>>>(" % (a24, cof),
        "        reinterpret_cast<const spint*>(bk), reinterpret_cast<const spint*>(bu), reinterpret_cast<spint*>(bv), n);",
        '    return check_launch("%s");' % sym,
        "}",
        "// the reference's own signature (rfc7748.c:156), host pointers: one record through the staging buffer",
        'extern "C" void %s(const char* bk, const char* bu, char* bv) {' % sym,
        "    StageBase s;                   // failures are recorded (modarith_amd_status()), never fatal; bv is zero-filled then",
        "    char *dk = (char*)s.take(NB), *du = (char*)s.take(NB), *dv = (char*)s.take(NB);",
        "    s.h2d(dk, bk, NB);",
        "    s.h2d(du, bu, NB);",
        '    if (!s.bad) s.check(%s_batch(dk, du, dv, 1, nullptr), "%s");' % (sym, sym),
        "    s.d2h(bv, dv, NB);",
        "}", ""])
    os.makedirs(d, exist_ok=True)
    unit, obj = os.path.join(d, "capi_ladder_%s.hip" % name), os.path.join(d, "capi_ladder_%s.o" % name)
    lib, meta = ladder_plugin_path(name, d), os.path.join(d, "ladder_%s.json" % name)
    from .build import ARCH, FLAGS, HIPCC, _stamp
    key = hashlib.sha256((" ".join(FLAGS) + "\n" + unit_text + "\n" + emit.header_text(fp) + "\n" + _stamp()).encode()).hexdigest()
    if not force and os.path.exists(lib) and os.path.exists(meta):
        try:
            if json.load(open(meta)).get("hash") == key:
                return lib
        except (ValueError, OSError):
            pass
    if not os.path.exists(HIPCC):
        raise GenerateError("%s not found: generating a ladder needs the ROCm compiler (there is no CPU path)" % HIPCC)
    emit._write(unit, unit_text)
    if verbose:
        print("[modarith_amd] hipcc %s -> %s" % (os.path.basename(unit), os.path.basename(lib)), flush=True)
    tmp = ".%d.tmp" % os.getpid()
    inc_dirs = ["-I", os.path.join(HERE, "csrc", "generated"), "-I", os.path.join(HERE, "csrc"), "-I", os.path.join(os.path.dirname(HERE), "include"), "-I", PLUGIN_DIR, "-I", d]
    subprocess.run([HIPCC] + list(FLAGS) + inc_dirs + ["-c", unit, "-o", obj + tmp], check=True, timeout=int(os.environ.get("MA_BUILD_TIMEOUT", "1500")))
    subprocess.check_call([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib + tmp, obj + tmp, "-L", HERE, "-l:libmodarith_amd.so",
                           "-Wl,-rpath,$ORIGIN/" + os.path.relpath(HERE, d), "-Wl,-rpath," + HERE])
    with open(meta + tmp, "w") as f:
        json.dump({"ladder": name, "field": field, "a24": a24, "cof": cof, "nbytes": fp.nbytes, "nbits": fp.n, "hash": key}, f, indent=1)
    os.replace(obj + tmp, obj)
    os.replace(lib + tmp, lib)
    os.replace(meta + tmp, meta)
    return lib


# ladders the test-suite generates: M-383 (Aranha-Barreto-Pereira-Ricardini: v^2 = u^3 + 2065150 u^2 + u over 2^383 - 187, the
# built-in PM383 field; base point u = 12) and a ladder over the GENERATED field 2^251 - 9 (A = 49382: a test curve -- the ladder
# is algebra on (A - 2) / 4 and is checked against plain integer arithmetic, whatever the curve's group looks like)
EXAMPLE_LADDERS = (dict(name="M383", field="PM383", a24=516287, cof=3), dict(name="T2519", field="2519", a24=12345, cof=3))


def installed_curves(plugin_dir: Optional[str] = None) -> List[dict]:
    d = plugin_dir or PLUGIN_DIR
    out = []
    if os.path.isdir(d):
        for f in sorted(os.listdir(d)):
            if f.startswith("curve_") and f.endswith(".json") and os.path.exists(curve_plugin_path(f[6:-5], d)):
                try:
                    out.append(json.load(open(os.path.join(d, f))))
                except ValueError:
                    pass
    return out


# curves the test-suite generates: Curve1174 (Bernstein-Hamburg-Krasnova-Lange: x^2 + y^2 = 1 - 1174 x^2 y^2 over 2^251 - 9, the generated
# field 2519) and NIST P-224 (a = -3 over the built-in NIST224 field) -- neither is in curve.py's table; the reference's own
# edwards.c / weierstrass.c, given the same definitions the way curve.py asks its user to insert them, produced
# tests/golden/curveref_CURVE1174.json / curveref_NIST224.json
EXAMPLE_CURVES = (
    dict(name="CURVE1174", kind="edwards", field="2519", a=1, b=-1174, cof=2,
         order=2**249 - 11332719920821432534773113288178349711,
         gx=1582619097725911541954547006453739763381091388846394833492296309729998839514,
         gy=3037538013604154504764115728651437646519513534305223422754827055689195992590),
    dict(name="NIST224", kind="weierstrass", field="NIST224", a=-3, b=0xb4050a850c04b3abf54132565044b0b7d7bfd8ba270b39432355ffb4,
         order=0xffffffffffffffffffffffffffff16a2e0b8f03e13dd29455c5c2a3d,
         gx=0xb70e0cbd6bb4bf7f321390b94a03c1d356c21122343280d6115c1d21, gy=0xbd376388b5f723fb4c22dfe6cd4375a05a07476444d5819985007e34),
)


def installed(plugin_dir: Optional[str] = None) -> List[dict]:
    """metadata of every plug-in whose shared object is present"""
    d = plugin_dir or PLUGIN_DIR
    out = []
    if os.path.isdir(d):
        for f in sorted(os.listdir(d)):
            if f.endswith(".json") and os.path.exists(plugin_path(f[:-5], d)):
                try:
                    m = json.load(open(os.path.join(d, f)))
                except ValueError:
                    continue
                if "tag" in m:                         # (the directory also holds the plug-ins of fused chains, modarith_amd/fuse.py)
                    out.append(m)
    return out


def params_of_plugin(tag: str, plugin_dir: Optional[str] = None) -> FieldParams:
    """FieldParams of an installed plug-in, re-derived from its recorded modulus / family / radix"""
    d = plugin_dir or PLUGIN_DIR
    meta = json.load(open(os.path.join(d, "%s.json" % tag)))
    p = int(meta["p"], 16)
    fp = (derive_pseudo if meta["family"] == "pseudo" else derive_monty)(tag, p, meta["radix"])
    return fp


def report(fp: FieldParams) -> str:
    """the lines the reference generators print about their choice (pseudo.py:1600-1612, monty.py:2200-2230), for the CLI"""
    L = ["Chosen radix is %d bits, using %d limbs with excess of %d bits" % (fp.radix, fp.nlimbs, fp.xcess)]
    if fp.family == "pseudo":
        L.append("pseudo-Mersenne 2^%d - %d: fold constant mm = %#x%s%s%s%s" % (fp.n, fp.m, fp.mm, ", overflow form" if fp.overflow else "",
                                                                             ", tighter reduction" if fp.fred else "", ", EPM" if fp.epm else "",
                                                                             ", carry_on" if fp.carry_on else ""))
    else:
        L.append("Montgomery form, R = 2^%d%s, ndash = %#x%s%s" % (fp.radix * (fp.nlimbs + (1 if fp.E else 0)), " (virtual limb)" if fp.E else "", fp.ndash,
                                                                  ", trinomial" if fp.trin else "", ", exploitable pseudo-Mersenne (PM)" if fp.pm else ""))
        L.append("prime limbs: " + " ".join(("%d" % v) if abs(v) < 10 else ("%#x" % v) for v in fp.ppw))
    L.append("split products: %s; inversion chain: 2-adicity %d" % ("cut at bit %d" % emit.split_point(fp) if emit.split_point(fp) else "exact 128-bit products only", fp.pm1d2))
    return "\n".join(L)


def time_report(tag: str, outer: int = 100000, lanes: int = 1 << 16) -> List[str]:
    """What the generators do last: build time.c and run it (pseudo.py:1861-1925, monty.py:2440-2500) -- seed-42 operands,
    `outer` x 1000 dependent modmul, the same number of modsqr, `outer` / 2 x 2... modinv, the 24-bit check word of each.  Here the
    chains run on the GPU, one per lane (csrc/kernels.h k_time): the words are the reference's for outer = 100000 (its own depth);
    the times are per dependent operation in one wave (latency-bound) and the rate with `lanes` chains in flight."""
    import random
    import time

    import torch
    from .field import Field
    F = Field(tag)
    fp = F.params
    random.seed(42)                                              # pseudo.py:1862-1866
    ra, rb, rs, ri = (random.randint(0, fp.p - 1) for _ in range(4))
    mk = lambda v: [(v >> (fp.radix * i)) & ((1 << fp.radix) - 1) for i in range(fp.nlimbs)]      # makebig: every limb masked
    out = []
    for leg, x, y, nops in (("modmul", ra, rb, outer * 1000), ("modsqr", rs, None, outer * 1000), ("modinv", ri, None, max(1, outer // 2) * 2)):
        o = outer if leg != "modinv" else max(1, outer // 2)
        res = {}
        for what, L in (("wave", 64), ("chip", lanes)):
            xa = F.from_limbs([mk(x)]).expand(-1, L).contiguous()
            ya = F.from_limbs([mk(y)]).expand(-1, L).contiguous() if y is not None else None
            F.time_protocol(leg, xa, ya, 1)
            torch.cuda.synchronize(F.device)
            t0 = time.perf_counter()
            z = F.time_protocol(leg, xa, ya, o)
            torch.cuda.synchronize(F.device)
            res[what] = time.perf_counter() - t0
            word = int(z[0, 0].item()) & 0xFFFFFF
        out.append("%s check 0x%06x Nanosecs= %d (one wave, per dependent call; %d calls)   %.3g %s/s with %d chains in flight"
                   % (leg, word, round(res["wave"] / nops * 1e9), nops, lanes * nops / res["chip"], leg, lanes))
    return out


def main(argv: List[str]) -> int:
    args = [a for a in argv if not a.startswith("--")]
    if "--list" in argv:
        for m in installed():
            print("%-12s %-6s %2d x %2d bits  %s" % (m["tag"], m["family"], m["nlimbs"], m["radix"], m["prime"]))
        for m in installed_curves():
            print("%-12s %-11s over %-8s a = %d, b = %s" % (m["curve"], m["kind"], m["field"], m["a"], m["b"]))
        d = PLUGIN_DIR
        for f in sorted(os.listdir(d)) if os.path.isdir(d) else []:
            if f.startswith("ladder_") and f.endswith(".json") and os.path.exists(ladder_plugin_path(f[7:-5])):
                m = json.load(open(os.path.join(d, f)))
                print("%-12s ladder      over %-8s A24 = %d, COF = %d" % (m["ladder"], m["field"], m["a24"], m["cof"]))
        return 0
    if args and args[0] == "ladder":
        # python -m modarith_amd.generate ladder <name> <field> <A24> <COF>
        if len(args) != 5:
            print("Valid syntax - python -m modarith_amd.generate ladder <name> <field> <A24> <COF>")
            return 2
        try:
            lib = generate_ladder(args[1], args[2], int(args[3], 0), int(args[4], 0), force="--force" in argv, verbose=True)
        except (GenerateError, ValueError) as e:
            print(e)
            return 2
        print("%s: void rfc7748_%s(const char *bk, const char *bu, char *bv); int rfc7748_%s_batch(bk, bu, bv, n, stream); rfc7748(%r, ...)" % (lib, args[1], args[1], args[1]))
        return 0
    if args and args[0] == "curve":
        # python -m modarith_amd.generate curve <NAME> edwards|weierstrass <field> <a> <b> <order> <gx> <gy> [cof]   (integers: any python literal)
        if len(args) not in (9, 10):
            print("Valid syntax - python -m modarith_amd.generate curve <name> edwards|weierstrass <field> <a> <b> <order> <gx> <gy> [log2 cofactor]")
            return 2
        try:
            nums = [int(v, 0) for v in args[4:]]
            g = generate_curve(args[1], args[2], args[3], nums[0], nums[1], nums[2], nums[3], nums[4], nums[5] if len(nums) > 5 else 0,
                               force="--force" in argv, verbose=True)
        except (GenerateError, ValueError) as e:
            print(e)
            return 2
        print("%s %s: C-ABI ecn_%s_* (MODARITH_AMD_DECLARE_EDWARDS(%s, %d)); Curve(%r)" % ("built" if g.built else "up to date:", g.lib, g.name.lower(), g.name.lower(), g.nlimbs, g.name))
        return 0
    if len(args) != 2:
        print("Syntax error")
        print("Valid syntax - python -m modarith_amd.generate <word length> <prime> OR <prime name> OR <name>=<prime> [--pseudo|--monty] [--force] [--time[=outer]]")
        print("For example - python -m modarith_amd.generate 64 2**255-19")
        return 2
    fam = "pseudo" if "--pseudo" in argv else "monty" if "--monty" in argv else None
    try:
        g = generate(args[1], int(args[0]), family=fam, force="--force" in argv, verbose=True)
    except GenerateError as e:
        print(e)
        return 2
    print(report(g.params))
    for a in argv:
        if a == "--time" or a.startswith("--time="):
            for line in time_report(g.tag, int(a.split("=", 1)[1]) if "=" in a else 100000):
                print(line)
    print("%s %s: C-ABI <fn>_%s_ct / <fn>_%s_batch (MODARITH_AMD_DECLARE(%s)); Field(%r)" % ("built" if g.built else "up to date:", g.lib, g.tag, g.tag, g.tag, g.tag))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
