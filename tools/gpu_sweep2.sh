#!/bin/bash
# grid-cap sweep of the element-wise kernels with non-temporal streaming + X448 ladder timing (GPU box)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
for mb in 256 512 1024 2048 4096 8192 16384 65536; do
  MA_MAX_BLOCKS=$mb python3 tools/sweep.py 24 20 > $OUT/sweep2_mb$mb.json 2>/dev/null
done
python3 - <<'PY' > $OUT/x448_ladder.txt 2>&1
import time, torch, sys, os
sys.path.insert(0, os.getcwd())
from modarith_amd.field import rfc7748
for C, nb, m in (("X448", 56, 1 << 20), ("X25519", 32, 1 << 22)):
    k = torch.randint(0, 256, (m, nb), dtype=torch.uint8, device="cuda")
    u = torch.randint(0, 256, (m, nb), dtype=torch.uint8, device="cuda")
    rfc7748(C, k[:4096], u[:4096]); torch.cuda.synchronize()
    t0 = time.perf_counter(); rfc7748(C, k, u); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(C, m, "scalars", dt * 1e3, "ms", m / dt, "per s")
os.environ["MA_LADDER_IMPL"] = "field"
PY
MA_LADDER_IMPL=field python3 - <<'PY' >> $OUT/x448_ladder.txt 2>&1
import time, torch, sys, os
sys.path.insert(0, os.getcwd())
from modarith_amd.field import rfc7748
m = 1 << 22
k = torch.randint(0, 256, (m, 32), dtype=torch.uint8, device="cuda")
u = torch.randint(0, 256, (m, 32), dtype=torch.uint8, device="cuda")
rfc7748("X25519", k[:4096], u[:4096]); torch.cuda.synchronize()
t0 = time.perf_counter(); rfc7748("X25519", k, u); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("X25519 field-form ladder", m, "scalars", dt * 1e3, "ms", m / dt, "per s")
PY
cat $OUT/x448_ladder.txt
