"""Register-budget guard (no GPU): every kernel of the built code objects stays inside the register file -- no spilled VGPRs, no
accumulation registers used as overflow (agpr_count: on gfx950 a kernel that asks for more than 256 VGPRs gets the upper half of the
unified file and runs at one wave per SIMD) -- unless tools/spill_allowlist.json names it with the reason and the measured cost.
tools/kernel_resources.py reads the figures from the metadata notes of the code objects (modarith_amd/build/*.o and the plug-ins).
The constant-time guard (tests/test_ct_audit.py) keeps `no branch on lane data` true; this one keeps `zero spills` true."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources  # noqa: E402


def _rows():
    rows = []
    for d in ("modarith_amd/build", "modarith_amd/plugins"):
        p = os.path.join(ROOT, d)
        if os.path.isdir(p):
            for f in sorted(os.listdir(p)):
                if f.endswith(".o"):
                    for k in kernel_resources.kernels_of(os.path.join(p, f)):
                        k["object"] = f
                        rows.append(k)
    return rows


def test_no_kernel_spills_outside_the_allow_list():
    if not os.path.exists(os.path.join(ROOT, "modarith_amd", "build", "capi_X25519.o")):
        pytest.skip("no built objects (run __graft_entry__.build())")
    allow = json.load(open(os.path.join(ROOT, "tools", "spill_allowlist.json")))["kernels"]
    rows = _rows()
    assert len(rows) > 3000                                   # the whole build was read
    bad, used = [], set()
    for k in rows:
        if not (k["vgpr_spill_count"] or k["agpr_count"]):
            continue
        ent = next((a for a in allow if a["match"] in k["name"] and (not a.get("object") or a["object"] in k["object"])), None)
        if ent is None:
            bad.append("%s [%s]: %d spilled, %d accumulation registers, no allow-list entry" % (k["name"][:120], k["object"], k["vgpr_spill_count"], k["agpr_count"]))
            continue
        used.add(ent["match"])
        if k["vgpr_spill_count"] > ent.get("vgpr_spill_count", 0) or k["agpr_count"] > ent.get("agpr_count", 0):
            bad.append("%s [%s]: %d spilled / %d accumulation registers, the allow-list permits %d / %d" % (
                k["name"][:120], k["object"], k["vgpr_spill_count"], k["agpr_count"], ent.get("vgpr_spill_count", 0), ent.get("agpr_count", 0)))
    assert not bad, "\n".join(bad)
    stale = [a["match"] for a in allow if a["match"] not in used]
    assert not stale, "allow-list entries that no kernel needs any more (remove them): %s" % stale
    for a in allow:
        assert a.get("why"), a


def test_headline_kernels_fit_their_occupancy():
    """the kernels of the BASELINE configs and of the fused ED25519 pipeline: registers low enough for the occupancy their launch code
    assumes, no scratch at all"""
    if not os.path.exists(os.path.join(ROOT, "modarith_amd", "build", "capi_X25519.o")):
        pytest.skip("no built objects")
    want = {  # substring -> (object, max VGPRs)
        "k_binary<ma::P_X25519, ma::OpMulAuto<ma::P_X25519>, 2>": ("capi_X25519.o", 96),
        "k_x25519_fe26_xz": ("capi_X25519.o", 168),
        "k_ed25519_lad(": ("capi_ED25519F.o", 168),
        "k_ed25519_lad_gen2": ("capi_ED25519G.o", 168),
        "k_cond<ma::P_X25519, true>": ("capi_X25519.o", 64),
    }
    for sub, (obj, vmax) in want.items():
        ks = [k for k in kernel_resources.kernels_of(os.path.join(ROOT, "modarith_amd", "build", obj)) if sub in k["name"]]
        assert ks, sub
        for k in ks:
            assert k["vgpr_count"] <= vmax and k["vgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0 and k["agpr_count"] == 0, (sub, k)
