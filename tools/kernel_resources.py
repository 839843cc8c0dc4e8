#!/usr/bin/env python3
"""Register / scratch budget of the gfx950 kernels in the built objects (modarith_amd/build/*.o, plug-ins).

  python tools/kernel_resources.py [substring ...] [--spills] [--json out.json]

Reads each object's .hip_fatbin section, unbundles the gfx950 code object and prints, from its metadata notes,
.vgpr_count / .agpr_count / .vgpr_spill_count / .sgpr_count / .private_segment_fixed_size per kernel (demangled).
No GPU needed.  --spills lists only kernels with spilled VGPRs or a scratch frame.
"""
import json
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("vgpr_count", "agpr_count", "vgpr_spill_count", "sgpr_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size")


def code_object(obj, tmp):
    """path of the gfx950 code object inside a host object (None if it has none)"""
    fb = os.path.join(tmp, "fb.bin")
    co = os.path.join(tmp, "k.co")
    r = subprocess.run([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fb], capture_output=True)
    if r.returncode or not os.path.exists(fb) or os.path.getsize(fb) == 0:
        return None
    r = subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fb,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], capture_output=True)
    return co if r.returncode == 0 else None


def kernels_of(obj):
    with tempfile.TemporaryDirectory() as tmp:
        co = code_object(obj, tmp)
        if co is None:
            return []
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    out = []
    for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
        blk = ".agpr_count:" + blk
        rec = {}
        for k in KEYS:
            m = re.search(r"\." + k + r":\s+(\d+)", blk)
            rec[k] = int(m.group(1)) if m else 0
        m = re.search(r"\.name:\s+(\S+)", blk)
        if not m:
            continue
        rec["symbol"] = m.group(1)
        out.append(rec)
    if out:
        names = subprocess.run(["c++filt"] + [r["symbol"] for r in out], capture_output=True, text=True).stdout.splitlines()
        for r, nm in zip(out, names):
            r["name"] = re.sub(r"^void ", "", nm)
    return out


def main(argv):
    pats = [a for a in argv if not a.startswith("--")]
    only_spills = "--spills" in argv
    jout = None
    if "--json" in argv:
        jout = argv[argv.index("--json") + 1]
        pats = [p for p in pats if p != jout]
    objs = []
    for d in ("modarith_amd/build", "modarith_amd/plugins"):
        p = os.path.join(ROOT, d)
        if os.path.isdir(p):
            objs += sorted(os.path.join(p, f) for f in os.listdir(p) if f.endswith(".o"))
    rows = []
    for o in objs:
        for k in kernels_of(o):
            if pats and not any(p in k["name"] for p in pats):
                continue
            if only_spills and not (k["vgpr_spill_count"] or k["private_segment_fixed_size"]):
                continue
            k["object"] = os.path.relpath(o, ROOT)
            rows.append(k)
    print("%5s %5s %6s %5s %8s  %s" % ("vgpr", "agpr", "vspill", "sgpr", "scratchB", "kernel  [object]"))
    for k in rows:
        print("%5d %5d %6d %5d %8d  %s  [%s]" % (k["vgpr_count"], k["agpr_count"], k["vgpr_spill_count"], k["sgpr_count"],
                                                 k["private_segment_fixed_size"], k["name"][:150], os.path.basename(k["object"])))
    if jout:
        with open(jout, "w") as f:
            json.dump(rows, f, indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
