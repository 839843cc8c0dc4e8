import os, sys
sys.path.insert(0, "/root/repo")
import torch, numpy as np
from modarith_amd.field import Field
from modarith_amd import emit
from modarith_amd.params import derive
P = sys.argv[1] if len(sys.argv) > 1 else "C2065"
fp = derive(P); F = Field(P)
n = 64
torch.manual_seed(1)
x = torch.randint(0, 1 << fp.radix, (fp.nlimbs, n), dtype=torch.int64, device="cuda")
x[fp.nlimbs - 1] &= (1 << (fp.n - fp.radix * (fp.nlimbs - 1))) - 1
def nsqr(a, k):
    for _ in range(k): a = F.modsqr(a)
    return a
prog = emit.addition_chain(fp.pe)
reg = {"x": x}; acc = None
for st in prog:
    if st[0] == "dbl": reg[st[1]] = F.modmul(nsqr(reg[st[2]], st[3]), reg[st[2]])
    elif st[0] == "inc": reg[st[1]] = F.modmul(F.modsqr(reg[st[2]]), x)
    elif st[0] == "start": acc = reg[st[1]]
    elif st[0] == "run": acc = F.modmul(nsqr(acc, st[1]), reg[st[2]])
    elif st[0] == "sqr": acc = nsqr(acc, st[1])
manual = acc
pro = F.modpro(x)
print("manual chain (batch modsqr/modmul) == modpro kernel:", torch.equal(manual, pro))
inv_h = F.modinv(x, pro)
inv = F.modinv(x)
print("modinv(x) == modinv(x, h):", torch.equal(inv, inv_h))
one = F.redc(F.modmul(inv, x))
print("x * modinv(x) == 1:", bool((one[0] == 1).all() and (one[1:] == 0).all()))
one = F.redc(F.modmul(inv_h, x))
print("x * modinv(x,h) == 1:", bool((one[0] == 1).all() and (one[1:] == 0).all()))
