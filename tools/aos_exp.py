import os, sys
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd())
from modarith_amd.fuse import Chain
def mk(P, name, multi):
    ch = Chain(P, name); ch.aos_multi = multi
    a, b = ch.inputs(2)
    ch.output(ch.modsqr(ch.modmul(ch.modadd(a, b), ch.modsub(a, b))))
    return ch.build()
if __name__ == "__main__":
    chains = {(P, m): mk(P, "aosx%d" % m, bool(m)) for P in ("X25519", "X448") for m in (0, 1)}
    if "--build" in sys.argv: sys.exit(0)
    import torch
    from modarith_amd.field import Field
    n = 1 << 24
    for P in ("X25519", "X448"):
        F = Field(P)
        x, y = F.to_aos(F.nres(F.uniform(n, array=0))), F.to_aos(F.nres(F.uniform(n, array=1)))
        z = [torch.empty_like(x), torch.empty_like(x)]
        for m in (0, 1):
            f = chains[(P, m)]
            f.aos(x, y, out=[z[m]]); torch.cuda.synchronize()
            ts = []
            for _ in range(9):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); f.aos(x, y, out=[z[m]]); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
            print(P, "multi" if m else "single", "%.3f ms" % sorted(ts)[4])
        print("equal:", bool(torch.equal(z[0], z[1])))
