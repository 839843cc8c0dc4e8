"""Reading and writing the golden fixtures.  A fixture is data (inputs and the reference's outputs as hex strings / byte lists);
the known-answer sets a reader wants to look at stay plain JSON (READABLE), everything else is the same JSON xz-compressed
(<name>.json.xz, about 4 x smaller: `xzcat tests/golden/field_GM384.json.xz | head`).  load() finds either form."""
import json
import lzma
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# kept as plain JSON: the headline field of BASELINE.json, the ladders, the curve of the reference's own test vector
READABLE = ("field_X25519.json", "ladder_X25519.json", "edwards_ED25519.json")


def path_of(name):
    p = os.path.join(HERE, name)
    return p if os.path.exists(p) or name in READABLE else p + ".xz"


def load(name):
    p = os.path.join(HERE, name)
    if os.path.exists(p):
        with open(p) as f:
            return json.load(f)
    with lzma.open(p + ".xz", "rt") as f:
        return json.load(f)


def dump(obj, name):
    p = os.path.join(HERE, name)
    if name in READABLE:
        with open(p, "w") as f:
            json.dump(obj, f, indent=0, separators=(",", ":"))
        return p
    if os.path.exists(p):
        os.remove(p)
    with open(p + ".xz", "wb") as g:                            # (xz streams carry no timestamp: identical data -> identical bytes)
        g.write(lzma.compress(json.dumps(obj, separators=(",", ":")).encode(), preset=9 | lzma.PRESET_EXTREME))
    return p + ".xz"
