#!/usr/bin/env python3
"""Does the best tile size of the tiled SoA layout depend on the SHAPE of the kernel?  (round 6: the 2-stream 8-limb kernels read 0.744
(r05 builder line) to 0.810 (r04 driver run) with their streaming control moving the same way, and modarith_amd_recommended_ld had one
answer, 4096, for every shape.)

  python tools/tile_shape_sweep.py [log2 n = 24] [placements = 8] [tiles = 1024,2048,4096,8192,flat]   ->  stdout (profiles/r06_tile_shape_sweep.log)

tile in {1024, 2048, 4096, 8192, flat} x {2 streams: modsqr, 3 streams: modmul} x {5 limbs: X25519, 8 limbs: X448}, each over
`placements` operand sets that are allocated one after the other and held together (so that they are different physical placements),
every set probed with 3 warm + 10 timed launches.  Reports worst / median / best fraction of the 8 TB/s HBM peak per configuration,
and the same for the streaming controls (modcpy: 2 streams, modadd: 3) on the tile size with the best worst case."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modarith_amd.field import Field

LG = int(sys.argv[1]) if len(sys.argv) > 1 else 24
NP = int(sys.argv[2]) if len(sys.argv) > 2 else 8
TILES = [int(t) if t != "flat" else None for t in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1024, 2048, 4096, 8192, None]
n = 1 << LG
PEAK = 8000.0


def probe(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10


def sweep(P, op, tile):
    F = Field(P, tile=tile)
    streams = 3 if op in ("modmul", "modadd") else 2
    sets = []
    for i in range(NP):
        a = F.uniform(n, array=2 * i)
        if F.params.montgomery:
            F.nres(a, out=a)
        b = None
        if streams == 3:
            b = F.uniform(n, array=2 * i + 1)
            if F.params.montgomery:
                F.nres(b, out=b)
        sets.append((a, b, torch.empty_like(a)))
    fr = []
    for a, b, c in sets:
        fn = {"modmul": lambda: F.modmul(a, b, out=c), "modadd": lambda: F.modadd(a, b, out=c), "modsqr": lambda: F.modsqr(a, out=c), "modcpy": lambda: F.modcpy(a, out=c)}[op]
        ms = probe(fn)
        fr.append(streams * 8 * F.N * n / (ms * 1e-3) / 1e9 / PEAK)
    del sets
    torch.cuda.empty_cache()
    return fr


def main():
    print("n = 2^%d elements, %d placements per configuration, fraction of %d GB/s (worst / median / best)" % (LG, NP, PEAK), flush=True)
    best = {}
    for P in ("X25519", "X448"):
        for op in ("modsqr", "modmul"):
            rows = []
            for tile in TILES:
                fr = sweep(P, op, tile)
                rows.append((min(fr), tile))
                print("%-7s %-7s %d limbs %d streams  tile %-5s  %.3f / %.3f / %.3f    %s" % (
                    P, op, Field(P).N, 3 if op == "modmul" else 2, tile or "flat", min(fr), statistics.median(fr), max(fr), " ".join("%.3f" % v for v in fr)), flush=True)
            best[(P, op)] = max(r for r in rows if r[1])[1]
    print("tile with the best WORST placement:", {"%s %s" % k: v for k, v in best.items()}, flush=True)
    for P in ("X25519", "X448"):
        for op, of in (("modcpy", "modsqr"), ("modadd", "modmul")):
            for tile in sorted({4096, best[(P, of)]}):
                fr = sweep(P, op, tile)
                print("control %-7s %-7s tile %-5s  %.3f / %.3f / %.3f" % (P, op, tile, min(fr), statistics.median(fr), max(fr)), flush=True)


if __name__ == "__main__":
    main()
