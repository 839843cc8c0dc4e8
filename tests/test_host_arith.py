"""The device arithmetic of the fused kernels, compiled for the HOST and run against the CPU oracle (no GPU needed):
tools/fe_host_check.hip instantiates the very functions the kernels wrap -- x25519_fe26_one, x448_fe28_one (ladders),
ed25519_mul_get_one, ed25519_mul2_get_one, ed448_mul_get_one (fused scalar / double multiplication + affine export) and the half-limb column products of
Field<P_X25519,true>, Field<P_NIST256,true> and Field<P_X448,true> -- with MA_DEV = __host__ __device__, and compares every output with the oracle
(rfc7748, ecn mul + ecn get, modmul / modsqr), including special points, corner scalars and the limb contract's edge
classes.  This checks the limb arithmetic and the group-law logic; code generation for gfx950 is checked on the GPU box."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC) and shutil.which("hipcc") is None, reason="needs hipcc (host compile of the HIP headers)")
def test_device_arithmetic_on_host_against_oracle(oracle, tmp_path):
    exe = str(tmp_path / "fe_host_check")
    odir = os.path.join(ROOT, "oracle")
    cc = HIPCC if os.path.exists(HIPCC) else "hipcc"
    subprocess.run([cc, "-O2", "-std=c++17", "-w", os.path.join(ROOT, "tools", "fe_host_check.hip"), "-o", exe, "--offload-arch=gfx950",
                    "-L" + odir, "-l:liboracle.so", "-Wl,-rpath," + odir], check=True, timeout=900)
    p = subprocess.run([exe, "400"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:]
    lines = [l for l in p.stdout.splitlines() if "records" in l]
    assert len(lines) == 13 and all(" 0 differ" in l for l in lines), p.stdout
