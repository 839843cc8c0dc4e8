#!/usr/bin/env python3
"""fused chains (modarith_amd/fuse.py) against the call-by-call sequence of the batched API: 2^24 elements on tiles of 4096,
HIP-event times, median of 9 (GPU box)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modarith_amd.field import Field
from modarith_amd.fuse import Chain

n = 1 << int(os.environ.get("LOG2N", "24"))


def timed(fn, reps=9):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


res = {}
for P in ("X25519", "NIST256", "X448"):
    F = Field(P, tile=4096)
    x, y = F.nres(F.uniform(n, array=0)), F.nres(F.uniform(n, array=1))
    o1, o2 = F.empty(n), F.empty(n)
    N = F.N
    # 1: products of sums: z = ((x + y)(x - y))^2  (4 calls, 440 / 704 B per element call by call; 120 / 192 fused)
    ch = Chain(P, "t_prod"); a, b = ch.inputs(2)
    ch.output(ch.modsqr(ch.modmul(ch.modadd(a, b), ch.modsub(a, b))))
    f1 = ch.build(ept=int(os.environ["EPT"]) if "EPT" in os.environ else None, policy=os.environ.get("POLICY", "vote"), waves=int(os.environ.get("WAVES", "0")))
    t1, t2 = F.empty(n), F.empty(n)
    def calls1():
        F.modadd(x, y, out=t1); F.modsub(x, y, out=t2); F.modmul(t1, t2, out=t1); F.modsqr(t1, out=o1)
    ms_f, ms_c = timed(lambda: f1(x, y, out=[o2])), timed(calls1)
    assert torch.equal(o1, o2)
    res["%s ((x+y)(x-y))^2" % P] = {"fused_ms": ms_f, "calls_ms": ms_c, "speedup": ms_c / ms_f, "fused_GBps": ch.traffic_bytes() * n / ms_f / 1e6,
                                    "fused_bytes_per_element": ch.traffic_bytes(), "calls_bytes_per_element": ch.unfused_traffic_bytes()}
    # 2: the doubling half of a ladder step (rfc7748.c:194-209), two results
    ch = Chain(P, "t_step"); a, b = ch.inputs(2)
    A, B = ch.modadd(a, b), ch.modsub(a, b)
    AA, BB = ch.modsqr(A), ch.modsqr(B)
    E = ch.modsub(AA, BB)
    ch.output(ch.modmul(AA, BB)); ch.output(ch.modmul(E, ch.modadd(AA, ch.modmli(E, 121665))))
    f2 = ch.build(ept=int(os.environ["EPT"]) if "EPT" in os.environ else None, policy=os.environ.get("POLICY", "vote"), waves=int(os.environ.get("WAVES", "0")))
    t3, t4, p1, p2 = F.empty(n), F.empty(n), F.empty(n), F.empty(n)
    def calls2():
        F.modadd(x, y, out=t1); F.modsub(x, y, out=t2); F.modsqr(t1, out=t1); F.modsqr(t2, out=t2)
        F.modsub(t1, t2, out=t3); F.modmul(t1, t2, out=o1); F.modmli(t3, 121665, out=t4); F.modadd(t1, t4, out=t4); F.modmul(t3, t4, out=o2)
    ms_f, ms_c = timed(lambda: f2(x, y, out=[p1, p2])), timed(calls2)
    assert torch.equal(o1, p1) and torch.equal(o2, p2)
    res["%s ladder-step doubling (9 calls, 2 results)" % P] = {"fused_ms": ms_f, "calls_ms": ms_c, "speedup": ms_c / ms_f, "fused_GBps": ch.traffic_bytes() * n / ms_f / 1e6,
                                                             "fused_bytes_per_element": ch.traffic_bytes(), "calls_bytes_per_element": ch.unfused_traffic_bytes()}
    # 3: the same two chains for a consumer that holds ELEMENT-MAJOR arrays x[n][N] (field.c's scalar callers): FusedChain.aos
    # against aos_to_soa x 2, the fused chain, soa_to_aos -- and against the converters around the call-by-call sequence
    xa, ya = F.to_aos(x), F.to_aos(y)
    za = torch.empty_like(xa)
    def conv_fused():
        u, v = F.from_aos(xa), F.from_aos(ya)
        r, = f1(u, v)
        return F.to_aos(r)
    def conv_calls():
        u, v = F.from_aos(xa), F.from_aos(ya)
        F.modadd(u, v, out=t1); F.modsub(u, v, out=t2); F.modmul(t1, t2, out=t1); F.modsqr(t1, out=o1)
        return F.to_aos(o1)
    ms_a, ms_cf, ms_cc = timed(lambda: f1.aos(xa, ya, out=[za])), timed(conv_fused), timed(conv_calls)
    assert torch.equal(za, conv_calls())
    res["%s ((x+y)(x-y))^2, element-major in and out" % P] = {"aos_chain_ms": ms_a, "converters_plus_fused_ms": ms_cf, "converters_plus_calls_ms": ms_cc,
                                                             "speedup_over_converters_plus_calls": ms_cc / ms_a, "GBps": 3 * 8 * N * n / ms_a / 1e6}
    del x, y, o1, o2, t1, t2, t3, t4, p1, p2, xa, ya, za
    torch.cuda.empty_cache()
print(json.dumps({"n": n, "layout": "tiles of 4096", "device": torch.cuda.get_device_name(0), "chains": res}, indent=1))
