#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X batched field engine.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): batched modmul modulo 2^255-19, 5 x 51-bit limbs, 2^24 field
elements per GPU in limb-interleaved SoA, inputs resident in HBM before the timed region.  One step =
one modmul pass over the batch (c[j] = a[j]*b[j], 120 algorithmic bytes per element).  With N GPUs
every rank owns its own 2^24-element batch (independent units, no data-path collective): weak
scaling; value = all ranks' modmuls / max-over-ranks time.

Output: the LAST stdout line is the contract line the driver parses (contract_line(): at most 4 KB); everything measured goes to
bench_detail.json beside this file (MA_BENCH_DETAIL: another path) and to stdout lines starting with "# detail".  In the line:
  roofline      HBM roofline of the modmul kernel: algorithmic bytes per launch / mean launch time, measured with HIP events
                on the launch stream over the timed region; traffic from rocprofv3 --pmc child passes of this script (N = 1).
  cpu_baseline  (rank 0, N=1 only) the CPU oracle -- a port of the reference's generated field.c -- on the host cores:
                all-core modmul throughput on a bounded sample, plus the reference's time.c protocol on one core.
  x25519        BASELINE.json configs[4] shape: batched RFC 7748 X25519 ladder, 2^23 scalars per GPU (--scaling strong: 2^26 in
                total over the ranks), fraction of the integer multiply-add issue ceiling, the RCCL gather timed separately.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
BYTES_PER_MODMUL = 120         # 3 arrays x 5 limbs x 8 B
LOG2_ELEMS = int(os.environ.get("MA_BENCH_LOG2_ELEMS", "24"))
LOG2_LADDER = int(os.environ.get("MA_BENCH_LOG2_LADDER", "23"))
# HBM layout of the field batches: limb-interleaved SoA in tiles of TILE elements (include/modarith_amd.h "TILED");
# MA_BENCH_TILE=0 benchmarks the flat layout buf[limb * n + j] instead.  The line reports the other layout as a data set.
TILE = int(os.environ.get("MA_BENCH_TILE", "4096"))


def _native_baseline_lib():
    """the cpu_baseline leg is timed on code built ON this machine with `gcc -O3 -march=native`:
    oracle/Makefile target `native` (the three BASELINE fields, the ladder, the pthread driver; ~3 s).  The portable
    liboracle.so that travels from the build container is the fallback, and the JSON says which one ran."""
    import subprocess
    odir = os.path.join(ROOT, "oracle")
    flags = "-O3 -march=native -fPIC -fno-semantic-interposition -funroll-loops"
    try:
        subprocess.run(["make", "-s", "-B", "-C", odir, "native"], check=True, capture_output=True, timeout=300)
        lib = ctypes.CDLL(os.path.join(odir, "libbaseline_native.so"))
        built = "on this host"
    except Exception as e:                                   # no compiler on the box: time the portable build
        lib = ctypes.CDLL(os.path.join(odir, "liboracle.so"))
        flags = "-O3 -march=x86-64-v3 -mtune=generic -fPIC -fno-semantic-interposition -funroll-loops"
        built = "in the build container (native build failed: %s)" % type(e).__name__
    try:
        cc = subprocess.run(["gcc", "--version"], capture_output=True, text=True).stdout.splitlines()[0]
    except Exception:
        cc = "gcc (version unknown)"
    lib.oracle_parallel.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int]
    lib.oracle_parallel.restype = ctypes.c_int
    lib.time_modmul_X25519.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), ctypes.c_long]
    lib.time_modmul_X25519.restype = ctypes.c_uint
    return lib, cc, flags, built


# the reference's time.c operands (seed 42) and check words per field: SURVEY 8(c), tests/golden/field_<P>.json "time"
TIME_C = {
    "X25519": (51, 5, ("11dc60f4392456de3eb13b9046685257bdd640fb06671ad11c80317fa3b1799d", "4b95423416419f828b9d2434e465e150bd9c66b3ad3c2d6d1a3d1fa7bc8960a9",
                       "4d0ef322815ef6d13b8faa1837f8a88b17fc695a07a0ca6e0822e8f36c031199", "35b2d3528b8148f6b38a088ca65ed389b74d0fb132e706298fadc1a606cb0fb3"),
               (0x116640, 0x675a88, 0xe70a06)),
    "NIST256": (52, 5, ("23b8c1e9392456de3eb13b9046685257bdd640fb06671ad11c80317fa3b1799d", "972a846916419f828b9d2434e465e150bd9c66b3ad3c2d6d1a3d1fa7bc8960a9",
                        "9a1de644815ef6d13b8faa1837f8a88b17fc695a07a0ca6e0822e8f36c031199", "6b65a6a48b8148f6b38a088ca65ed389b74d0fb132e706298fadc1a606cb0fb3"),
                (0xa47501, 0x717a99, 0xe1e067)),
    "X448": (56, 8, ("8b9d2434e465e150bd9c66b3ad3c2d6d1a3d1fa7bc8960a923b8c1e9392456de3eb13b9046685257bdd640fb06671ad11c80317fa3b1799d",
                     "b74d0fb132e706298fadc1a606cb0fb39a1de644815ef6d13b8faa1837f8a88b17fc695a07a0ca6e0822e8f36c031199972a846916419f82",
                     "28df6ec4ce4a2bbdc241330b01a9e71fde8a774bcf36d58b4737819096da1dac72ff5d2a386ecbe06b65a6a48b8148f6b38a088ca65ed389",
                     "5be6128e18c267976142ea7d17be31111a2a73ed562b0f79c37459eef50bea63371ecd7b27cd813047229389571aa8766c307511b2b9437a"),
             (0xbcdde4, 0xa8450d, 0x189f52)),
}


def time_c_protocol(lib, P):
    """the three legs of the reference's time.c for field P on ONE host core, full depth; every check word is asserted"""
    radix, nl, ops, want = TIME_C[P]
    U = ctypes.c_uint64 * nl
    mk = lambda h: U(*[(int(h, 16) >> (radix * i)) & ((1 << radix) - 1) for i in range(nl)])      # makebig, pseudo.py:190-199
    fm, fs, fi = getattr(lib, "time_modmul_" + P), getattr(lib, "time_modsqr_" + P), getattr(lib, "time_modinv_" + P)
    fm.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), ctypes.c_long]
    fs.argtypes = fi.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_long]
    fm.restype = fs.restype = fi.restype = ctypes.c_uint
    out = {}
    for leg, call, nops, ref in (("modmul", lambda: fm(mk(ops[0]), mk(ops[1]), 100000), 10**8, want[0]),
                                 ("modsqr", lambda: fs(mk(ops[2]), 100000), 10**8, want[1]),
                                 ("modinv", lambda: fi(mk(ops[3]), 50000), 10**5, want[2])):
        t0 = time.perf_counter()
        w = call()
        dt = time.perf_counter() - t0
        assert w == ref, "time.c %s %s check word %#x, the reference's is %#x" % (P, leg, w, ref)
        out[leg] = {"ns_per_op": dt / nops * 1e9, "ops": nops, "check_word": "0x%06x" % w, "reference_check_word": "0x%06x" % ref}
    return out


def cpu_quota_cores():
    """the CPU time this process may use per second of wall clock, in cores: cgroup v2 cpu.max (or v1 cfs quota / period) of the
    process's own cgroup and its ancestors, capped by the affinity mask; None when no quota is set.  The visible core count
    (sched_getaffinity) overstates a container whose quota is smaller -- round 4's cpu_baseline.cores said 256 on a box whose
    quota was about ten."""
    quota = None
    try:
        rel = ""
        for line in open("/proc/self/cgroup"):
            parts = line.strip().split(":", 2)
            if len(parts) == 3 and parts[0] == "0":
                rel = parts[2]
        path = os.path.normpath("/sys/fs/cgroup/" + rel.lstrip("/"))
        while path.startswith("/sys/fs/cgroup"):
            f = os.path.join(path, "cpu.max")
            if os.path.exists(f):
                q, per = open(f).read().split()[:2]
                if q != "max":
                    v = float(q) / float(per)
                    quota = v if quota is None else min(quota, v)
            if path == "/sys/fs/cgroup":
                break
            path = os.path.dirname(path)
    except Exception:
        pass
    if quota is None:
        try:                                                     # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    return quota


def cpu_baseline(a_host, b_host, min_seconds=6.0):
    """oracle (kind "port": CPU restatement of the reference's generated field.c, limb-exact against the
    reference's golden vectors) timed on the host cores: all-core modmul throughput over the same
    2^24-element workload (the very arrays the GPU multiplied), the reference's full time.c protocol on one core,
    and the ladder."""
    import numpy as np
    from tests.util import vp
    lib, cc, flags, built = _native_baseline_lib()
    cores = len(os.sched_getaffinity(0))          # threads started (what the affinity mask shows)
    quota = cpu_quota_cores()                     # what the container may actually burn: `cores_usable` is the number to divide by
    n = a_host.shape[1]
    a, b = a_host, b_host
    c = np.empty_like(a)
    lib.oracle_parallel(0, vp(a), vp(b), vp(c), n, n, cores)  # warm (page faults, thread start)
    passes, t0 = 0, time.perf_counter()
    while True:
        lib.oracle_parallel(0, vp(a), vp(b), vp(c), n, n, cores)
        passes += 1
        dt = time.perf_counter() - t0
        if dt >= min_seconds or passes >= 1024:
            break
    thr = passes * n / dt
    del c
    # reference-faithful latency: the full time.c protocol on one core -- all three legs, the reference's loop counts, its
    # seed-42 operands (random.seed(42), four randint(0,p-1): pseudo.py:1862-1866) and its 24-bit check words:
    # 10^8 dependent modmul (pseudo.py:1177-1253), 10^8 modsqr (1256-1322), 10^5 modinv (1324-1386)
    time_c = time_c_protocol(lib, "X25519")
    lat = time_c["modmul"]["ns_per_op"] * 1e-9
    chk = int(time_c["modmul"]["check_word"], 16)
    # the Montgomery fields of configs[2], configs[3] the same way, their nine legs spread over host threads (ctypes
    # releases the GIL); X25519 above runs alone so that its figures are undisturbed
    import concurrent.futures as cf
    with cf.ThreadPoolExecutor(max_workers=2) as ex:
        time_c_other = dict(zip(("NIST256", "X448"), ex.map(lambda P: time_c_protocol(lib, P), ("NIST256", "X448"))))
    # ladder on all cores, bounded sample
    rng = np.random.default_rng(7)
    m = 512 * cores
    k = rng.integers(0, 256, size=(m, 32), dtype=np.uint8)
    u = rng.integers(0, 256, size=(m, 32), dtype=np.uint8)
    o = np.empty_like(u)
    t0 = time.perf_counter()
    lib.oracle_parallel(3, vp(k), vp(u), vp(o), m, 0, cores)
    ldt = time.perf_counter() - t0
    t0 = time.perf_counter()
    lib.oracle_parallel(3, vp(k), vp(u), vp(o), 256, 0, 1)       # one thread, for the effective parallelism
    l1 = 256 / (time.perf_counter() - t0)
    return {
        "value": thr, "unit": "modmul/s", "cores": cores, "cpu_quota_cores": quota, "cores_usable": min(cores, quota) if quota else cores, "kind": "port",
        "sample": "oracle modmul_X25519 over the 2^%d-element workload (the GPU's own input arrays) x %d passes, %d threads, %.1f s wall" % (n.bit_length() - 1, passes, cores, dt),
        "compiler": cc, "flags": flags, "built": built,
        "time_c_protocol": dict(time_c, ns_per_modmul=lat * 1e9, modmul_per_s=1.0 / lat, cores=1, dependent_modmuls=10**8,
                                check_word=hex(chk), reference_check_word="0x116640"),
        "time_c_protocol_other_fields": time_c_other,
        "x25519_scalar_mults_per_s": m / ldt, "x25519_sample": "%d ladders, %d threads, %.1f s wall" % (m, cores, ldt),
        "x25519_one_thread_per_s": l1,
        "effective_parallelism": {"x25519": (m / ldt) / l1, "modmul": thr * lat,
                                  "note": "all-thread rate / one-thread rate; tracks cpu_quota_cores, not `cores`, when the container's CPU quota is smaller than the visible core count (modmul over 2 GB of SoA arrays is also DRAM-bound)"},
    }


_VALU_DOC = {}
MAD_ISSUE_CYCLES = 4.62        # v_mad_u64_u32, cycles per wave-instruction per SIMD at 8 waves/SIMD: measured, isolated stream of independent chains
#                                (profiles/r05_valubench.log:15; 4.63 in profiles/r06_valubench.log).  Architectural: 16 lanes per clock = 4.0.
SIMDS = 1024                   # 256 CUs x 4 SIMDs


def _valu_doc():
    """the counter summary (instructions and multiply-adds per record of every VALU-bound leg, tools/collect_valu_legs_pmc.py), the
    algorithmic floors (tools/mad_floor.py) and the unit hashes of the library in this tree (modarith_amd/unit_hashes.json)"""
    if "doc" not in _VALU_DOC:
        doc, floor, now = None, {}, {}
        for name in ("r06_valu_pmc.json", "r05_valu_pmc.json"):
            path = os.path.join(ROOT, "profiles", name)
            if os.path.exists(path):
                doc = json.load(open(path))
                doc["_source"] = "profiles/" + name
                break
        fpath = os.path.join(ROOT, "profiles", "mad_floor.json")
        if os.path.exists(fpath):
            floor = json.load(open(fpath)).get("legs", {})
        hpath = os.path.join(ROOT, "modarith_amd", "unit_hashes.json")
        if os.path.exists(hpath):
            now = json.load(open(hpath))
        _VALU_DOC.update(doc=doc, floor=floor, now=now)
    return _VALU_DOC["doc"], _VALU_DOC["floor"], _VALU_DOC["now"]


def valu_roofline(leg, per_s_per_gpu, sclk_GHz=None):
    """Integer multiply-add roofline of a VALU-bound leg (both ladders, ecn mul / mul2, the fused curve kernels): SURVEY 8(d) "the
    meaningful ceiling is integer-MAD issue" (rfc7748.c:186-221 is nothing but field products).
      peak  = SIMDS x clock x 64 lanes / (MAD_ISSUE_CYCLES x mad_per_scalar)      [records/s]
      frac  = achieved records/s / peak, at the shader clock measured beside the leg (modarith_amd/clock.py)
    mad_per_scalar: v_mad_u64_u32 per record of THIS tree's kernels (disassembly weighted by trip counts, tools/isa_mix.py, in the
    counter summary).  Every instruction that is not a multiply-add lowers `frac`; so does a multiply-add the algorithm does not need:
    mad_floor_per_scalar is the schoolbook count of the leg's algorithm on its limb form (profiles/mad_floor.json, tools/mad_floor.py)
    and frac_of_floor_ceiling prices the leg against THAT count.  frac_of_mix_ceiling is the figure rounds 2-5 led with: the ceiling of
    the kernel's own instruction mix (5.0 / 2.5 cycles per multiply-add / other instruction at 2.4 GHz) -- a scheduler-efficiency
    figure that cannot fall when instructions are wasted; kept for comparison only."""
    doc, floors, now = _valu_doc()
    if not doc or leg not in doc.get("legs", {}):
        return None
    L = doc["legs"][leg]
    # were the kernels of this leg rebuilt from other sources since the counters were taken?  (None: the summary does not say)
    was = doc.get("unit_hashes")
    stale = any(now.get(u) != was.get(u) for u in L["units"]) if (was and now and L.get("units")) else None
    instr, mad = L["instr_per_scalar"], L["mad_per_scalar"]
    if not instr or not mad:
        return None
    clk = sclk_GHz or 2.4
    peak = SIMDS * clk * 1e9 * 64 / (MAD_ISSUE_CYCLES * mad)
    floor = floors.get(leg, {}).get("mad_floor_per_scalar")
    mix_cost = (5.0 * mad + 2.5 * (instr - mad)) / instr
    mix_peak = SIMDS * 2.4e9 * 64 / (mix_cost * instr)
    return {"bound": "valu-mad", "achieved": per_s_per_gpu, "peak": peak, "unit": "records/s", "frac": per_s_per_gpu / peak,
            "frac_at_2.4GHz": per_s_per_gpu / (peak * 2.4 / clk), "frac_at_architectural_4_cycles": per_s_per_gpu / (peak * MAD_ISSUE_CYCLES / 4.0),
            "sclk_GHz": sclk_GHz, "mad_issue_cycles": MAD_ISSUE_CYCLES,
            "mad_per_scalar": mad, "mad_floor_per_scalar": floor, "mad_over_floor": (mad / floor) if floor else None,
            "frac_of_floor_ceiling": (per_s_per_gpu / (peak * mad / floor)) if floor else None,
            "frac_of_mix_ceiling": per_s_per_gpu / mix_peak, "instr_per_scalar": instr, "non_mad_per_mad": (instr - mad) / mad,
            "kernels": sorted(L["kernels"]), "source": doc["_source"], "source_stale": stale}


def measure_traffic(timeout_s=240):
    """HBM bytes per launch of the headline kernel, measured in THIS run: two child processes -- started after this process's own
    timed regions -- run this same script for a few launches under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes
    with --kernel-trace only, as MI355X_MICROARCH.md's HBM section prescribes; the program itself after `--`), and their
    counter_collection.csv is read back.  gfx950 correction of that section: FETCH_SIZE reports half the bytes of a 16-B-per-lane
    coalesced streaming read -> doubled; WRITE_SIZE is exact; both in KiB.  Any failure (no rocprofv3, a refused counter) returns
    a dict with only "failed": <reason>; the caller then quotes the committed summary of the same passes
    (profiles/traffic_modmul_X25519.json) and says in traffic_source that the live measurement failed and why.
    The program after `--` is the REAL interpreter binary of this process (os.path.realpath(sys.executable)): a `python3`
    resolved through PATH may be a shim or wrapper script -- an exec hop after the profiler's preload has initialised the GPU,
    which this pool forbids -- or another interpreter without torch."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"failed": "rocprofv3 not found"}
    py = os.path.realpath(sys.executable)
    kern = "k_binary<ma::P_X25519, ma::OpMulAuto<ma::P_X25519"
    out = {}
    env = dict(os.environ, TMPDIR="/tmp", MA_BENCH_PLACEMENTS="1", MA_BENCH_CHILD="1")
    base = tempfile.mkdtemp(prefix="ma_traffic_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(base, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", py, os.path.abspath(__file__),
                   "--steps", "5", "--warmup", "2", "--no-cpu", "--no-ladder", "--no-others", "--no-verify", "--no-traffic"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
            files = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))
            if r.returncode != 0 or not files:
                return {"failed": "%s pass: exit code %d, %d counter file(s)" % (counter, r.returncode, len(files))}
            vals = [float(row["Counter_Value"]) for row in csv.DictReader(open(files[0])) if kern in row["Kernel_Name"] and row.get("Counter_Name", counter) == counter]
            if not vals:
                return {"failed": "%s pass: no row of the headline kernel in the counter file" % counter}
            out[counter] = (sum(vals) / len(vals), len(vals))
    except Exception as ex:
        return {"failed": "%s: %s" % (type(ex).__name__, str(ex)[:120])}
    finally:
        shutil.rmtree(base, ignore_errors=True)
    fk, wk = out["FETCH_SIZE"][0], out["WRITE_SIZE"][0]
    return {"hbm_bytes_per_launch": 2 * fk * 1024 + wk * 1024, "hbm_read_bytes_per_launch": 2 * fk * 1024, "hbm_write_bytes_per_launch": wk * 1024,
            "launches_sampled": min(out["FETCH_SIZE"][1], out["WRITE_SIZE"][1]),
            "source": "measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child passes of this script (FETCH_SIZE doubled per the gfx950 correction)"}


DETAIL_NAME = "bench_detail.json"
LINE_LIMIT = 4096              # bytes: the driver's parser refused the 35 KB line of round 5 (BENCH_r05.parsed = null); r04's 15 KB passed


def _sig(x, digits=5):
    """floats to `digits` significant figures (the detail file keeps full precision)"""
    if isinstance(x, float):
        return float("%.*g" % (digits, x))
    if isinstance(x, list):
        return [_sig(v, digits) for v in x]
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    return x


def detail_lines(detail):
    """the detail dict as stdout lines that do not start with "{": one per top-level key, one per leg of the two big blocks"""
    out = []
    for k, v in detail.items():
        if k in ("other_configs", "data_sets") and isinstance(v, dict):
            out += ["# detail %s.%s: %s" % (k, kk, json.dumps(vv)) for kk, vv in v.items()]
        else:
            out.append("# detail %s: %s" % (k, json.dumps(v)))
    return out


def contract_line(d, detail_note=DETAIL_NAME):
    """The one JSON line the driver parses, built from the detail dict: the contract keys, the HBM roofline of the headline kernel with
    its placement fractions, the CPU baseline summary, the X25519 ladder summary, the verdict of the oracle check and where the rest
    is.  Pure function of `d` (tests/test_bench_line.py feeds it recorded detail files); never longer than LINE_LIMIT bytes."""
    cfg, rl, cpu, lad, ver = d["config"], d["roofline"], d.get("cpu_baseline"), d.get("x25519"), d.get("verified_against_oracle")
    out = {k: d[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    out["config"] = {"workload": cfg["workload"], "elements_per_gpu": cfg["elements_per_gpu"], "placement_probe_GBps": cfg["placement_probe_GBps"],
                     "placement_policy": cfg["placement_policy"][:120]}
    out["roofline"] = {k: rl.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "algorithmic_bytes_per_launch",
                                               "frac_first_placement", "frac_median_placement")}
    out["roofline"]["traffic_source"] = ("live rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE child passes" if "measured in this run" in (rl.get("traffic_source") or "")
                                         else (rl.get("traffic_source") or "")[:160] or None)
    if cpu:
        tc = cpu.get("time_c_protocol") or {}
        words = [leg for leg in (tc.get("modmul"), tc.get("modsqr"), tc.get("modinv")) if isinstance(leg, dict)]
        for other in (cpu.get("time_c_protocol_other_fields") or {}).values():
            words += [leg for leg in other.values() if isinstance(leg, dict)]
        out["cpu_baseline"] = {"value": cpu["value"], "unit": cpu["unit"], "cores": cpu["cores"], "cpu_quota_cores": cpu.get("cpu_quota_cores"), "kind": cpu["kind"],
                               "compiler": (cpu.get("compiler") or "")[:60], "flags": (cpu.get("flags") or "")[:90], "sample": (cpu.get("sample") or "")[:160],
                               "ns_per_modmul_one_core": tc.get("ns_per_modmul"),
                               "check_words_ok": bool(words) and all(w.get("check_word") == w.get("reference_check_word") for w in words), "check_words": len(words),
                               "x25519_scalar_mults_per_s": cpu.get("x25519_scalar_mults_per_s")}
    else:
        out["cpu_baseline"] = None
    if lad:
        lr = lad.get("roofline") or {}
        out["x25519"] = {"value": lad["value"], "unit": lad["unit"], "scalars_per_gpu": lad["scalars_per_gpu"], "scalars_total": lad.get("scalars_total"), "scaling": lad.get("scaling"),
                         "sclk_GHz": lad.get("sclk_GHz"), "frac_of_mad_only_ceiling": lr.get("frac"), "frac_of_floor_ceiling": lr.get("frac_of_floor_ceiling"),
                         "frac_of_mix_ceiling": lr.get("frac_of_mix_ceiling"), "mad_per_scalar": lr.get("mad_per_scalar"), "mad_floor_per_scalar": lr.get("mad_floor_per_scalar"),
                         "value_wall_clock_3_passes": lad.get("value_wall_clock_3_passes"), "gather_ms": lad.get("gather_ms"), "gather_GBps": lad.get("gather_GBps"),
                         "records_sha256": lad.get("records_sha256"),
                         "host_resident_pipelined_per_s": (lad.get("host_resident") or {}).get("end_to_end_pipelined_per_s")}
    else:
        out["x25519"] = None
    out["verified_against_oracle"] = ({"all_ranks_equal_oracle": ver.get("all_ranks_equal_oracle"), "ranks_checked": ver.get("ranks_checked"),
                                       "modmul_elements_per_rank": ver.get("modmul_elements_per_rank"), "x25519_records_per_rank": ver.get("x25519_records_per_rank")}
                                      if ver else None)
    sp = (d.get("rank_spread") or {}).get("modmul_per_s")
    if d["n_gpus"] > 1 and sp:
        out["rank_spread_modmul_per_s"] = [sp["min"], sp["mean"], sp["max"]]
    if d.get("dist"):
        out["dist"] = d["dist"]
    out["detail"] = detail_note
    out = {k: (_sig(v) if isinstance(v, (dict, list)) else v) for k, v in out.items()}      # the contract's own scalars keep full precision
    line = json.dumps(out, separators=(",", ":"))
    if len(line) > LINE_LIMIT:                               # cannot happen with the bounded strings above; the contract keys survive whatever does
        for k in ("dist", "rank_spread_modmul_per_s", "x25519", "verified_against_oracle"):
            out.pop(k, None)
            line = json.dumps(out, separators=(",", ":"))
            if len(line) <= LINE_LIMIT:
                break
    return line


def launch_ranks(n_ranks, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, one process per GPU, through
    torch.distributed.run (the command the driver uses for N>1), relay their output and return their exit code.
    This parent never imports torch.cuda nor touches HIP, and it does not replace itself: the ranks are children."""
    import socket
    import subprocess
    with socket.socket() as sk:             # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-ladder", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-others", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="do not measure the headline kernel's HBM traffic with rocprofv3 child passes (N=1 only; the committed summary is quoted instead)")
    ap.add_argument("--no-verify", action="store_true", help="skip the per-rank spot check of the timed outputs against the CPU oracle")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="x25519 block: weak = 2^MA_BENCH_LOG2_LADDER records per GPU (default); strong = BASELINE.json configs[4] literally, "
                         "2^MA_BENCH_LOG2_LADDER_TOTAL (26) records divided over the N ranks.  The headline modmul line is weak either way.")
    ap.add_argument("--launch-check", action="store_true",
                    help="only check the rank launch: gloo group over the N ranks, no GPU work (tests/test_bench_launch.py)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.launch_check:
        dist.init_process_group("gloo")
        tt = torch.tensor([rank + 1], dtype=torch.int64)
        dist.all_reduce(tt)
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "rank_sum": int(tt[0])}))
        dist.destroy_process_group()
        return
    # MA_BENCH_BACKEND=gloo lets the N>1 control flow be exercised on a box with fewer GPUs than ranks
    backend = os.environ.get("MA_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and world > ndev:
        raise SystemExit("bench.py: %d ranks but only %d GPU(s) visible" % (world, ndev))
    dev = torch.device("cuda", local % max(ndev, 1))
    torch.cuda.set_device(dev)
    # MA_BENCH_FORCE_DIST=1 initialises the process group even for one rank, so that the RCCL code path (init, max
    # over ranks, barrier, result gather) can be exercised on a single-GPU box
    use_dist = world > 1 or os.environ.get("MA_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)      # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend)
    if args.gpus != world and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d; reporting n_gpus = WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    cdev = dev if backend == "nccl" else torch.device("cpu")   # where collective payloads live

    def max_over_ranks(vals):
        tt = torch.tensor(vals, dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return [float(v) for v in tt]

    single = world == 1            # side figures (other data sets, other fields, the curve layer) are N=1 material only

    from modarith_amd.field import Field, rfc7748
    F = Field("X25519", dev, tile=TILE or None)
    n = 1 << LOG2_ELEMS
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    # SURVEY 8(d) C2 input recipe: a[j], b[j] uniform in [0,p), element j of the splitmix64 stream keyed by
    # (seed 42, array id) reduced mod p -- generated on the device (csrc/kernels.h k_uniform), regenerable on the host
    # from (seed, array, j) alone (tests/util.py uniform_model).  Every rank draws its own arrays (ids 16*rank + ...).
    SEED, AID = 42, 16 * rank
    a = F.uniform(n, seed=SEED, array=AID + 0)
    b = F.uniform(n, seed=SEED, array=AID + 1)
    c = torch.empty_like(a)
    # Placement probe: the streaming rate of one and the same kernel over one and the same data differs
    # reproducibly with WHERE the driver put the pages of the three arrays -- by up to 12 % on flat rows, by 1-4 % on tiles.
    # MA_BENCH_PLACEMENTS (default 4) operand triples with identical contents are allocated one after the other and each is
    # probed with 10 launches; all probe rates are reported.  The timed region runs on the FIRST-allocated triple -- what a
    # caller that allocates once gets -- unless MA_BENCH_KEEP_BEST=1 asks for the fastest of the probed ones (rounds 1-2).
    placements = max(1, int(os.environ.get("MA_BENCH_PLACEMENTS", "4")))
    keep_best = os.environ.get("MA_BENCH_KEEP_BEST") == "1"
    probe_rates = []
    if placements > 1:
        def probe(x, y, z):
            for _ in range(3):
                F.modmul(x, y, out=z)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10):
                F.modmul(x, y, out=z)
            e1.record()
            torch.cuda.synchronize()
            return BYTES_PER_MODMUL * n * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        cands = [(a, b, c)]
        for _ in range(placements - 1):
            cands.append((F.uniform(n, seed=SEED, array=AID + 0), F.uniform(n, seed=SEED, array=AID + 1), torch.empty_like(a)))
        probe_rates = [probe(*t) for t in cands]
        best = max(range(len(cands)), key=lambda i: probe_rates[i]) if keep_best else 0
        a, b, c = cands[best]
        del cands
        torch.cuda.empty_cache()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        F.modmul(a, b, out=c)
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        F.modmul(a, b, out=c)
    ev1.record()
    barrier()
    dt = time.perf_counter() - t0
    kern_ms = ev0.elapsed_time(ev1) / args.steps      # mean launch duration on the launch stream
    my_dt, my_kern_ms = dt, kern_ms                   # this rank's own figures (reported per rank below)
    if use_dist:
        dt, kern_ms = max_over_ranks([dt, kern_ms])
    value = world * n * args.steps / dt
    achieved = BYTES_PER_MODMUL * n / (kern_ms * 1e-3) / 1e9

    # SURVEY 8(d) "report median": per-launch durations of a SEPARATE pass of the same launches on the same buffers (one event
    # between consecutive launches; the timed region above keeps the contract's single bracket and is not touched)
    nl = max(20, min(args.steps, 100))
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(nl + 1)]
    torch.cuda.synchronize()
    for i in range(nl):
        evs[i].record()
        F.modmul(a, b, out=c)
    evs[nl].record()
    torch.cuda.synchronize()
    per_launch = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(nl))
    launch_stats = {"launches": nl, "median_ms": per_launch[nl // 2] if nl % 2 else 0.5 * (per_launch[nl // 2 - 1] + per_launch[nl // 2]),
                    "min_ms": per_launch[0], "max_ms": per_launch[-1], "p10_ms": per_launch[nl // 10], "p90_ms": per_launch[(9 * nl) // 10]}
    launch_stats["median_GBps"] = BYTES_PER_MODMUL * n / (launch_stats["median_ms"] * 1e-3) / 1e9
    launch_stats["frac_of_hbm_peak_median"] = launch_stats["median_GBps"] / HBM_PEAK_GBS

    # size-independent correctness guard inside the bench: a*b == b*a and (a*b) canonical form is stable
    chk = F.modmul(b, a)
    assert torch.equal(chk, c), "modmul is not commutative bit-for-bit: kernel bug"
    del chk

    def rate(fn, reps=20, warm=3, warm_ms=60.0):
        # >= 3 warm-up launches AND >= 60 ms of them: the side legs are separated by host work (allocations, input
        # generation, a plug-in load), during which the part drops its clocks; the first 20-40 streaming launches after such a gap
        # run 2-5 % slower while they ramp back (profiles/history/r04_chain_dvfs.log: every kernel, not one in
        # particular), and three launches of 0.3 ms do not cover that.  Kernels of tens of ms per launch get their three.
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(warm):
            fn()
        e1.record()
        torch.cuda.synchronize()
        spent = e0.elapsed_time(e1)
        for _ in range(min(2000, int(max(0.0, warm_ms - spent) / max(spent / warm, 1e-3)))):
            fn()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    # SURVEY 8(d) C2's second and third data sets (non-canonical limbs), short runs of the same kernel:
    # values in [p,2p) with the top limb unmasked (what the reference's self-test feeds, pseudo.py:1763-1775), and
    # the previous pass's outputs fed back in (what time.c and the ladders do, pseudo.py:1235-1242)
    data_sets = {"uniform_mod_p": {"modmul_per_s_per_gpu": n / (kern_ms * 1e-3), "GBps": achieved, "kernel_ms": kern_ms,
                                   "recipe": "splitmix64(seed 42, array id, j), %d words, mod p; canonical limbs" % 5}}
    if not args.no_others and single:
        a2 = F.uniform(n, seed=SEED, array=AID + 0, plus_p=True)
        b2 = F.uniform(n, seed=SEED, array=AID + 1, plus_p=True)
        c2 = torch.empty_like(a2)
        ms = rate(lambda: F.modmul(a2, b2, out=c2))
        data_sets["p_to_2p_top_limb_unmasked"] = {"modmul_per_s_per_gpu": n / (ms * 1e-3), "GBps": BYTES_PER_MODMUL * n / (ms * 1e-3) / 1e9, "kernel_ms": ms}
        # same residues, other representative: the canonical results must agree
        r1, r2 = F.redc(c), F.redc(c2)
        assert torch.equal(r1, r2), "modmul of [p,2p) representatives disagrees with the canonical ones"
        del a2, b2, r1, r2
        F.modmul(c, a, out=c2)                       # fed back: operands are outputs of previous passes
        fb = torch.empty_like(c2)
        ms = rate(lambda: F.modmul(c2, c, out=fb))
        data_sets["fed_back_outputs"] = {"modmul_per_s_per_gpu": n / (ms * 1e-3), "GBps": BYTES_PER_MODMUL * n / (ms * 1e-3) / 1e9, "kernel_ms": ms}
        # shared multiplicand c[j] = a[j]*b0 (80 B per element)
        b0 = [int(v) for v in F.to_limbs(F.uniform(1, seed=SEED, array=AID + 1))[0]]      # element 0 of b
        ms = rate(lambda: F.modmuls(a, b0, out=fb))
        data_sets["shared_multiplicand"] = {"modmul_per_s_per_gpu": n / (ms * 1e-3), "GBps": 80 * n / (ms * 1e-3) / 1e9, "kernel_ms": ms, "bytes_per_element": 80,
                                            "frac_of_hbm_peak": 80 * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel": "k_mul_shared<P_X25519,2,true>"}
        # streaming controls on the same buffers: the same three (two) streams with no multiplication
        ms = rate(lambda: F.modadd(a, b, out=fb))
        data_sets["control_modadd_3_streams"] = {"GBps": BYTES_PER_MODMUL * n / (ms * 1e-3) / 1e9, "kernel_ms": ms, "frac_of_hbm_peak": BYTES_PER_MODMUL * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        ms = rate(lambda: F.modcpy(a, out=fb))
        data_sets["control_modcpy_2_streams"] = {"GBps": 80 * n / (ms * 1e-3) / 1e9, "kernel_ms": ms, "frac_of_hbm_peak": 80 * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        del c2, fb
        # the same batch through the OTHER layout (flat rows n * 8 bytes apart when the headline runs on tiles, and vice versa)
        Fo = Field("X25519", dev, tile=None if TILE else 4096)
        ao, bo = (F.to_flat(a), F.to_flat(b)) if TILE else (Fo.to_tiled(a), Fo.to_tiled(b))
        co = torch.empty_like(ao)
        ms = rate(lambda: Fo.modmul(ao, bo, out=co))
        assert torch.equal(F.to_flat(c), Fo.to_flat(co)), "the two layouts disagree"
        data_sets["other_layout_%s" % ("flat" if TILE else "tiled_4096")] = {"modmul_per_s_per_gpu": n / (ms * 1e-3), "GBps": BYTES_PER_MODMUL * n / (ms * 1e-3) / 1e9,
                                                                              "kernel_ms": ms, "frac_of_hbm_peak": BYTES_PER_MODMUL * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                                              "note": "same operand values, same kernel, the other HBM layout; one placement (no probe)"}
        del ao, bo, co
        # what a caller gets who writes Field("X25519") and nothing else (no tile argument): the object's default layout
        Fd = Field("X25519", dev)
        ad, bd = Fd.uniform(n, seed=SEED, array=AID + 0), Fd.uniform(n, seed=SEED, array=AID + 1)
        cd = torch.empty_like(ad)
        ms = rate(lambda: Fd.modmul(ad, bd, out=cd))
        assert torch.equal(Fd.to_flat(cd), F.to_flat(c)), "the default-layout batch disagrees"
        data_sets["default_caller"] = {"default_caller_GBps": BYTES_PER_MODMUL * n / (ms * 1e-3) / 1e9, "kernel_ms": ms, "frac_of_hbm_peak": BYTES_PER_MODMUL * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                       "layout": "tiled, tile = %d" % ad.shape[2] if ad.dim() == 3 else "flat", "note": "Field('X25519').uniform(n) / .modmul(): no layout argument given"}
        del ad, bd, cd

    others = {}
    # (Measured right after the streaming data sets.  With ~75-80 % VALU issue occupancy the four-call chain has little headroom over
    # HBM and is the first kernel to show the clock ramp after a host-side pause -- rate() warms for 60 ms for that reason:
    # docs/fused_chains.md, profiles/history/r04_chain_dvfs.log.)
    # Fused chains (modarith_amd/fuse.py): a sequence of field.c calls per element as ONE streaming kernel on
    # registers, against the same calls through the batched API, on the timed region's own operands.  z = ((a + b)(a - b))^2:
    # four calls, 440 B per element call by call, 120 B fused.  The chain's plug-in is built by __graft_entry__.build() and
    # travels with the tree (rebuilt here in seconds if it is not current); equal limbs are asserted.
    if not args.no_others and single:
        chain_equal = True
        try:
            from modarith_amd.fuse import bench_chain
            ch = bench_chain("X25519")
            fz = ch.build()
            t1, t2, zc = torch.empty_like(a), torch.empty_like(a), torch.empty_like(a)
            def calls():
                F.modadd(a, b, out=t1); F.modsub(a, b, out=t2); F.modmul(t1, t2, out=t1); F.modsqr(t1, out=zc)
            # The fused kernel is timed on the headline's own operand triple (a, b -> c): the same three streams over the same
            # placement as the timed region, so the two rates compare like with like (a separately allocated output lands on another
            # placement and reads 5-10 % lower, which says nothing about the kernel: profiles/history/r04_chain_pmc.json).  c is restored
            # in the `finally` whatever happens, because the verifier below checks c = a * b.
            try:
                ms_f = rate(lambda: fz(a, b, out=[c]))
                chain_equal = None                                   # (decided below, once zc exists)
                ms_c = rate(calls)
                chain_equal = bool(torch.equal(c, zc))
            finally:
                F.modmul(a, b, out=c)
            others["fused_chain_X25519"] = {"chain": "modsqr(modmul(modadd(a,b), modsub(a,b)))", "elements": n, "fused_ms": ms_f, "calls_ms": ms_c,
                                            "speedup": ms_c / ms_f, "fused_bytes_per_element": ch.traffic_bytes(), "calls_bytes_per_element": ch.unfused_traffic_bytes(),
                                            "fused_GBps": ch.traffic_bytes() * n / (ms_f * 1e-3) / 1e9, "field_ops_per_s_per_gpu": 4 * n / (ms_f * 1e-3),
                                            "frac_of_hbm_peak": ch.traffic_bytes() * n / (ms_f * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                            "frac_of_headline_kernel": kern_ms / ms_f,
                                            "buffers": "the timed region's operand triple (a, b -> c)",
                                            "limbs_equal_to_call_sequence": chain_equal}
            del t1, t2, zc
        except (RuntimeError, OSError, subprocess.CalledProcessError, ValueError) as ex:   # a missing compiler on the box must not cost the headline line
            others["fused_chain_X25519"] = {"skipped": repr(ex)[:200]}
            chain_equal = True
        assert chain_equal, "fused chain differs from the call-by-call sequence"

    # the other single-GPU configs of BASELINE.json (configs[2], configs[3]) with the same protocol, short runs:
    # parity for them is in tests/; these are side figures, not the headline
    if not args.no_others and single:
        for P, ops in (("NIST256", ("modmul",)), ("X448", ("modmul", "modsqr"))):
            # the tile size the library recommends for the field's shape (round 6: 8192 for the 8-limb X448, profiles/r06_tile_shape_sweep.log)
            Fp = Field(P, dev, tile=(Field(P, dev).recommended_tile(n) if "MA_BENCH_TILE" not in os.environ else (TILE or None)))
            # uniform in [0,p) by the same recipe, then nres (Montgomery form)
            xa = Fp.uniform(n, seed=SEED, array=AID + 2)
            xb = Fp.uniform(n, seed=SEED, array=AID + 3)
            Fp.nres(xa, out=xa)
            Fp.nres(xb, out=xb)
            xc = torch.empty_like(xa)
            for op in ops:
                fn = (lambda: Fp.modmul(xa, xb, out=xc)) if op == "modmul" else (lambda: Fp.modsqr(xa, out=xc))
                # streaming control on the SAME three (two) buffers: the same streams without the multiplication
                ctl = (lambda: Fp.modadd(xa, xb, out=xc)) if op == "modmul" else (lambda: Fp.modcpy(xa, out=xc))
                ms, cms = rate(fn), rate(ctl)                # interleaved, best of two each: the rate of a buffer set drifts by a per cent or two
                ms, cms = min(ms, rate(fn)), min(cms, rate(ctl))
                nbytes = (3 if op == "modmul" else 2) * 8 * Fp.N * n
                others["%s_%s" % (P, op)] = {"ops_per_s_per_gpu": n / (ms * 1e-3), "GBps": nbytes / (ms * 1e-3) / 1e9,
                                             "frac_of_hbm_peak": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel_ms": ms,
                                             "control": "modadd" if op == "modmul" else "modcpy", "control_GBps": nbytes / (cms * 1e-3) / 1e9,
                                             "frac_of_control": cms / ms,
                                             "tile": Fp.tile, "inputs": "uniform mod p (splitmix64 recipe), nres'd"}
            del xa, xb, xc
        # the curve layer built on the path, one pass each: side figures (VALU-bound kernels)
        from modarith_amd.edwards import Curve
        from modarith_amd.clock import timed_with_clock
        last_clock = [None]

        def vleg(key, rate):                                     # the VALU roofline of a curve leg, at the clock just measured
            return valu_roofline(key, rate, last_clock[0]) or {"bound": "valu-mad", "note": "no counter summary for this leg under profiles/"}
        for cname, m in (("ED25519", 1 << 20), ("ED448", 1 << 19), ("SECP256K1", 1 << 19), ("NIST256", 1 << 19)):
            Cv = Curve(cname, dev)
            e = torch.randint(0, 256, (m, Cv.nbytes), dtype=torch.uint8, device=dev, generator=gen)
            f = torch.randint(0, 256, (m, Cv.nbytes), dtype=torch.uint8, device=dev, generator=gen)
            G = Cv.gen(m)
            # warm-up of EVERY function timed below on a small slice (loads the code objects, allocates the workspaces), so that
            # the multi-call reference legs are not charged first-call costs the fused legs have already paid
            Pp = Cv.mul(e[:4096].contiguous(), Cv.gen(4096))
            Cv.get(Pp)
            Cv.mul2(e[:4096].contiguous(), Cv.gen(4096), f[:4096].contiguous(), Pp)
            del Pp
            # ... and one call at FULL size: the window-table workspace is sized to the resident grid (up to 566 MB for ED25519 at four
            # waves per SIMD) and is allocated on the first call that needs it -- device memory management is not what the leg measures
            Cv.mul(e, G.clone())

            def timed_leg(fn, reps=3, warm=2):
                # median of `reps` calls, each between HIP events, after `warm` full-size calls: the first launches after a host-side
                # pause run 3-13 % slower while the part brings its clocks back (profiles/history/r04_ecn_sustained.log: 2.67, 2.93, 3.08, 3.17 ...
                # e7/s for P-256), and a single wall-clock call -- what this block timed until the end of round 4 -- reads exactly that.
                # One further call runs with the shader-clock probe beside it (modarith_amd/clock.py): last_clock[0] = GHz during the leg.
                t, ghz, out = timed_with_clock(fn, reps=reps, warm=warm)
                last_clock[0] = ghz
                return t, out

            t_mul, Q = timed_leg(lambda: Cv.mul(e, G.clone()))                  # (the clone -- 3 x N x 8 bytes per point -- rides in the leg: < 0.1 %)
            others["%s_ecn_mul" % cname] = {"scalar_mults_per_s_per_gpu": m / t_mul, "points": m, "bound": "VALU", "sclk_GHz": last_clock[0],
                                            "roofline": vleg("%s_ecn_mul" % cname, m / t_mul)}
            t_mul2, R = timed_leg(lambda: Cv.mul2(e, G, f, Q))
            others["%s_ecn_mul2" % cname] = {"double_mults_per_s_per_gpu": m / t_mul2, "pairs": m, "bound": "VALU", "sclk_GHz": last_clock[0],
                                             "roofline": vleg("%s_ecn_mul2" % cname, m / t_mul2)}
            if cname in Cv.FUSED:
                # the reference's call pattern ecnXXXmul + ecnXXXget (ed448.c:182-184): two-call form against the fused kernel
                tf, (fx_, fy_, _) = timed_leg(lambda: Cv.mul_get(e, Q))
                rl = vleg("%s_ecn_mul_get_fused" % cname, m / tf)
                tw, (wx_, wy_, _) = timed_leg(lambda: Cv.get(Cv.mul(e, Q.clone())))
                assert torch.equal(fx_, wx_) and torch.equal(fy_, wy_), "fused mul_get differs from mul + get"
                others["%s_ecn_mul_get_fused" % cname] = {"scalar_mults_per_s_per_gpu": m / tf, "points": m, "bound": "VALU", "sclk_GHz": rl.get("sclk_GHz"), "roofline": rl,
                                                          "two_call_form_per_s": m / tw, "speedup": tw / tf,
                                                          "bytes_equal_to_two_call_form": True}
                del fx_, fy_, wx_, wy_
            if cname in getattr(Cv, "FUSED2", ()):
                # verification pattern ecnXXXmul2 + ecnXXXget (ed448.c:305): fused against the two calls
                mq = m // 2
                e2, f2, G2, Q2 = e[:mq].contiguous(), f[:mq].contiguous(), G[:, :, :mq].contiguous(), Q[:, :, :mq].contiguous()
                tf, (fx_, fy_, _) = timed_leg(lambda: Cv.mul2_get(e2, G2, f2, Q2))
                rl = vleg("%s_ecn_mul2_get_fused" % cname, mq / tf)
                tw, (wx_, wy_, _) = timed_leg(lambda: Cv.get(Cv.mul2(e2, G2, f2, Q2)))
                assert torch.equal(fx_, wx_) and torch.equal(fy_, wy_), "fused mul2_get differs from mul2 + get"
                others["%s_ecn_mul2_get_fused" % cname] = {"double_mults_per_s_per_gpu": mq / tf, "pairs": mq, "bound": "VALU", "sclk_GHz": rl.get("sclk_GHz"), "roofline": rl,
                                                           "two_call_form_per_s": mq / tw, "speedup": tw / tf,
                                                           "bytes_equal_to_two_call_form": True}
                del fx_, fy_, wx_, wy_, e2, f2, G2, Q2
            if cname in getattr(Cv, "FUSEDG", ()):
                # key generation / signing opening ecnXXXgen + ecnXXXmul + ecnXXXget (nist256.c:150-161, ed448.c:167-184): fixed-base kernel
                tf, (gx_, gy_, _) = timed_leg(lambda: Cv.mulgen_get(e))
                rl = vleg("%s_ecn_mulgen_get_fused" % cname, m / tf)
                tw, (wx_, wy_, _) = timed_leg(lambda: Cv.get(Cv.mul(e, Cv.gen(m))))
                assert torch.equal(gx_, wx_) and torch.equal(gy_, wy_), "fused mulgen_get differs from gen + mul + get"
                others["%s_ecn_mulgen_get_fused" % cname] = {"scalar_mults_per_s_per_gpu": m / tf, "scalars": m, "bound": "VALU", "sclk_GHz": rl.get("sclk_GHz"), "roofline": rl,
                                                             "three_call_form_per_s": m / tw, "speedup": tw / tf,
                                                             "bytes_equal_to_three_call_form": True}
                del gx_, gy_, wx_, wy_
            if cname in getattr(Cv, "FUSEDG2", ()):
                # verification ecnXXXgen + ecnXXXmul2(e, G, f, Q) + ecnXXXget (nist256.c:251-256, ed448.c:305): generator part on the fixed-base table
                mq = m // 2
                e2, f2, Q2 = e[:mq].contiguous(), f[:mq].contiguous(), Q[:, :, :mq].contiguous()
                tf, (vx_, vy_, _) = timed_leg(lambda: Cv.mulgen2_get(e2, f2, Q2))
                rl = vleg("%s_ecn_mulgen2_get_fused" % cname, mq / tf)
                tw, (wx_, wy_, _) = timed_leg(lambda: Cv.get(Cv.mul2(e2, Cv.gen(mq), f2, Q2)))
                assert torch.equal(vx_, wx_) and torch.equal(vy_, wy_), "fused mulgen2_get differs from gen + mul2 + get"
                others["%s_ecn_mulgen2_get_fused" % cname] = {"double_mults_per_s_per_gpu": mq / tf, "pairs": mq, "bound": "VALU", "sclk_GHz": rl.get("sclk_GHz"), "roofline": rl,
                                                              "three_call_form_per_s": mq / tw, "speedup": tw / tf,
                                                              "bytes_equal_to_three_call_form": True}
                del vx_, vy_, wx_, wy_, e2, f2, Q2
            del e, f, G, Q, R

    # The reference's time.c protocol ON THE GPU (the shape of simd/pseudo_cuda.py:1163-1231: every lane runs the serially
    # dependent chains on the seed-42 operands, in registers): one wave for the latency per operation, 2^18 lanes for the
    # in-register (VALU-bound) aggregate rate.  Depth 10^5 (the reference's loop counts / 1000: its `scale`; a single wave needs
    # 0.7 us per dependent modmul, the full 10^8 would take a minute); check words of that depth from the reference-generated
    # fixture tests/golden/field_X25519.json "time" (the modinv chain has period 2, so its word does not depend on the depth).
    if not args.no_others and single:
        radix, nl, ops, _ = TIME_C["X25519"]
        mkl = lambda h: [(int(h, 16) >> (radix * i)) & ((1 << radix) - 1) for i in range(nl)]
        tc = {}
        for leg, opx, opy, outer, nops, ref in (("modmul", ops[0], ops[1], 100, 10**5, 0x570963), ("modsqr", ops[2], None, 100, 10**5, 0x9d7cea),
                                                ("modinv", ops[3], None, 500, 10**3, 0xe70a06)):
            res = {}
            for tag, lanes in (("one_wave", 64), ("all_lanes", 1 << 18)):
                xa = F.from_limbs([mkl(opx)]).expand(-1, lanes).contiguous()
                ya = F.from_limbs([mkl(opy)]).expand(-1, lanes).contiguous() if opy else None
                F.time_protocol(leg, xa, ya, 1)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                z = F.time_protocol(leg, xa, ya, outer)
                torch.cuda.synchronize(); t1 = time.perf_counter() - t0
                w = int(z[0, 0].item()) & 0xFFFFFF
                assert w == ref and bool((z == z[:, :1]).all()), "time.c %s on the GPU: check word %#x, the reference's is %#x" % (leg, w, ref)
                res[tag] = t1
                del xa, ya, z
            tc[leg] = {"one_wave_ns_per_op": res["one_wave"] / nops * 1e9, "ops": nops, "check_word": "0x%06x" % ref, "reference_check_word": "0x%06x" % ref,
                       "all_lanes_ops_per_s": (1 << 18) * nops / res["all_lanes"], "lanes": 1 << 18}
        others["time_c_protocol_gpu_X25519"] = tc

    ladder = None
    my_lt = None
    x448 = None
    if not args.no_ladder:
        if args.scaling == "strong":
            # BASELINE.json configs[4] literally: 2^26 records in total, contiguous shards of 2^26 / N per rank
            from modarith_amd.dist import shard_range
            total = 1 << int(os.environ.get("MA_BENCH_LOG2_LADDER_TOTAL", "26"))
            lo, hi = shard_range(total, rank, world)
            m = hi - lo
        else:
            m = 1 << LOG2_LADDER
        if args.scaling == "strong":
            # the records of the whole job are a function of their GLOBAL index (splitmix64 of (stream, index, word)), so that N ranks
            # working on [lo, hi) and one rank working on [0, total) multiply the same records: the gathered bytes of an N-rank run
            # must equal those of a one-rank run (x25519.records_sha256; tests/test_gpu_bench.py runs 8 ranks against 1)
            def records(stream, lo_, hi_):
                M = (1 << 64) - 1
                def s64(v):                                    # two's-complement int64 of a 64-bit constant
                    v &= M
                    return v - (1 << 64) if v >= (1 << 63) else v
                idx = torch.arange(lo_, hi_, dtype=torch.int64, device=dev).repeat_interleave(4) * 4 + torch.arange(4, dtype=torch.int64, device=dev).repeat(hi_ - lo_)
                z = (idx + 1 + (stream << 40)) * s64(0x9E3779B97F4A7C15)
                z = (z ^ ((z >> 30) & ((1 << 34) - 1))) * s64(0xBF58476D1CE4E5B9)
                z = (z ^ ((z >> 27) & ((1 << 37) - 1))) * s64(0x94D049BB133111EB)
                z = z ^ ((z >> 31) & ((1 << 33) - 1))
                return z.view(torch.uint8).reshape(hi_ - lo_, 32).contiguous()
            k, u = records(1, lo, hi), records(2, lo, hi)
        else:
            k = torch.randint(0, 256, (m, 32), dtype=torch.uint8, device=dev, generator=gen)
            u = torch.randint(0, 256, (m, 32), dtype=torch.uint8, device=dev, generator=gen)
        o = torch.empty_like(u)
        # Timed like the curve legs since round 5 (two full-size warm passes, then the MEDIAN of five passes, each between HIP events
        # on the launch stream): up to round 4 this leg was one warm pass and the wall clock over three, right behind the one-wave
        # time.c legs that leave the part nearly idle for 0.3 s -- it read the clock ramp (driver run of round 4: 1.06e8/s where the
        # same binary gives 1.17e8/s once warm).  The contract's barrier brackets the whole block; the shader clock during one further
        # pass is measured by the probe of modarith_amd/clock.py and reported beside the rate.
        from modarith_amd.clock import clock_during
        rfc7748("X25519", k, u, out=o)                       # (code objects, the split form's workspace)
        rfc7748("X25519", k, u, out=o)
        barrier()
        reps, lts = 5, []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rfc7748("X25519", k, u, out=o)
            e1.record()
            torch.cuda.synchronize()
            lts.append(e0.elapsed_time(e1) * 1e-3)
        lt = sorted(lts)[reps // 2]
        lt_all = sorted(lts)
        ladder_clock, _ = clock_during(lambda: rfc7748("X25519", k, u, out=o), lt, dev)
        barrier()
        # the method of rounds 2-4 beside the median, so that the rounds stay comparable: wall clock over three passes between barriers
        tw0 = time.perf_counter()
        for _ in range(3):
            rfc7748("X25519", k, u, out=o)
        barrier()
        lt_wall = (time.perf_counter() - tw0) / 3
        my_lt = lt
        gather_ms = None
        m_all = m
        records_sha = None
        if not use_dist and args.scaling == "strong":
            import hashlib
            records_sha = hashlib.sha256(o.cpu().numpy().tobytes()).hexdigest()
        if use_dist:
            from modarith_amd.dist import gather_records
            payload = o if backend == "nccl" else o.cpu()
            tt = torch.tensor([m], dtype=torch.int64, device=cdev)
            dist.all_reduce(tt)
            m_all = int(tt[0])
            barrier()
            t0 = time.perf_counter()
            allv = gather_records(payload, m_all, dst=0)       # the only collective: final result gather (RCCL over xGMI)
            barrier()
            gather_ms = (time.perf_counter() - t0) * 1e3
            if rank == 0:
                assert allv.shape[0] == m_all and torch.equal(allv[:m].to(o.device), o)
                import hashlib
                records_sha = hashlib.sha256(allv.cpu().numpy().tobytes()).hexdigest()
            del allv
            lt, gather_ms, lt_wall = max_over_ranks([lt, gather_ms, lt_wall])
        ladder = {"value": m_all / lt, "unit": "X25519 scalar-mults/s", "scalars_per_gpu": m, "scalars_total": m_all, "ms_per_pass": lt * 1e3,
                  "scaling": args.scaling, "shard": [lo, hi] if args.scaling == "strong" else None,
                  "records_sha256": records_sha,      # strong scaling: all results in global index order (gathered to rank 0): the same for every N
                  "gather_ms": gather_ms, "io_bytes_per_scalar": 96,
                  "gather_GBps": (m_all * 32 / (gather_ms * 1e-3) / 1e9) if gather_ms else None,
                  "gather_payload_bytes": m_all * 32 if gather_ms else None,
                  "bound": "VALU 32-bit integer multiply-add issue (not HBM)",
                  "value_wall_clock_3_passes": m_all / lt_wall, "timing": "value: median of %d event-timed passes after 2 full-size warm passes (max over ranks); value_wall_clock_3_passes: wall clock over 3 passes between barriers (the method of rounds 2-4)" % reps, "ms_per_pass_min": lt_all[0] * 1e3, "ms_per_pass_max": lt_all[-1] * 1e3,
                  "sclk_GHz": ladder_clock, "roofline": valu_roofline("x25519", m / my_lt, ladder_clock)}
        if single:
            # public-key generation: the same function on the base point u = 9 (rfc7748.c:297-333), fixed-base kernel
            from modarith_amd.field import rfc7748_base
            rfc7748_base("X25519", k[:4096].contiguous())
            torch.cuda.synchronize(); tb0 = time.perf_counter()
            pk = rfc7748_base("X25519", k)
            torch.cuda.synchronize(); tb = time.perf_counter() - tb0
            ub = torch.zeros((4096, 32), dtype=torch.uint8, device=dev); ub[:, 0] = 9
            assert torch.equal(pk[:4096], rfc7748("X25519", k[:4096].contiguous(), ub)), "fixed-base public keys differ from the ladder on u = 9"
            del pk, ub
            ladder["base_point_public_keys_per_s_per_gpu"] = m / tb
            # SURVEY 8(d) "include H2D/D2H separately for C5": the reference's own GPU harness holds its records on the HOST
            # (simd/rfc7748_simt.cu:257-270 cudaMemcpy in, 302-312 cudaMemcpy out).  Pinned host buffers, C ABI only
            # (modarith_amd/hostio.py): the two transfer legs alone, the three legs one after the other, and the pipelined form
            # (three streams, two device slots, chunks of 2^20 records) whose bytes are compared with the device-resident result.
            from modarith_amd import _lib as mlib
            from modarith_amd.hostio import PinnedBytes, ladder_host
            L = mlib.load()
            hk, hu, hv = PinnedBytes(m, 32), PinnedBytes(m, 32), PinnedBytes(m, 32)
            hk.array[:] = k.cpu().numpy(); hu.array[:] = u.cpu().numpy()
            def leg(fn, reps=3):
                fn(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / reps
            def h2d():
                mlib.check(L.modarith_amd_memcpy_h2d(k.data_ptr(), hk.ptr, m * 32, None), "h2d"); mlib.check(L.modarith_amd_memcpy_h2d(u.data_ptr(), hu.ptr, m * 32, None), "h2d")
                mlib.check(L.modarith_amd_sync(None), "sync")
            def d2h():
                mlib.check(L.modarith_amd_memcpy_d2h(hv.ptr, o.data_ptr(), m * 32, None), "d2h"); mlib.check(L.modarith_amd_sync(None), "sync")
            t_in, t_out = leg(h2d), leg(d2h)
            t_pipe = leg(lambda: ladder_host("X25519", hk, hu, hv, chunk=1 << 20), reps=2)
            import numpy as np
            assert np.array_equal(hv.array, o.cpu().numpy()), "host-resident (pipelined) ladder differs from the device-resident one"
            ladder["host_resident"] = {"h2d_ms": t_in * 1e3, "d2h_ms": t_out * 1e3, "h2d_bytes": m * 64, "d2h_bytes": m * 32,
                                       "h2d_GBps": m * 64 / t_in / 1e9, "d2h_GBps": m * 32 / t_out / 1e9,
                                       "end_to_end_serial_per_s": m / (t_in + lt + t_out), "end_to_end_pipelined_per_s": m / t_pipe,
                                       "pipelined_ms": t_pipe * 1e3, "chunk_records": 1 << 20, "device_resident_per_s": m / lt,
                                       "pipelined_over_device_resident": lt / t_pipe,
                                       "note": "pinned host records in, pinned host results out (simd/rfc7748_simt.cu:257-312 shape); pipelined bytes equal the device-resident result"}
            hk.close(); hu.close(); hv.close()
            # the other curve of rfc7748.c (120-132): X448, 56-byte records, 2^MA_BENCH_LOG2_X448 (21) of them
            m4 = 1 << int(os.environ.get("MA_BENCH_LOG2_X448", "21"))
            k4 = torch.randint(0, 256, (m4, 56), dtype=torch.uint8, device=dev, generator=gen)
            u4 = torch.randint(0, 256, (m4, 56), dtype=torch.uint8, device=dev, generator=gen)
            o4 = torch.empty_like(u4)
            from modarith_amd.clock import timed_with_clock
            t4, clk4, _ = timed_with_clock(lambda: rfc7748("X448", k4, u4, out=o4), reps=3, warm=2)
            x448 = {"value": m4 / t4, "unit": "X448 scalar-mults/s", "scalars_per_gpu": m4, "ms_per_pass": t4 * 1e3, "io_bytes_per_scalar": 168,
                    "timing": "median of 3 event-timed passes after 2 full-size warm passes", "sclk_GHz": clk4,
                    "bound": "VALU 32-bit integer multiply-add issue (not HBM)", "roofline": valu_roofline("x448", m4 / t4, clk4)}
    # EVERY rank spot-checks its own timed outputs against the CPU oracle (checker only, outside every timed
    # region): first / last 4096 and a strided sample of the modmul batch and of the ladder records.  The verdicts are
    # AND-reduced over the ranks, so that one N-GPU line says whether all N devices computed the reference's results.
    verified, my_ok = None, None
    if not args.no_verify:
        import numpy as np
        from tests.oracle_binding import load_oracle
        from tests.util import vp
        if rank != 0 and use_dist:
            dist.barrier()                                   # rank 0 builds the checker first if it did not travel
        oracle = load_oracle(build=(rank == 0 and not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so"))))
        if rank == 0 and use_dist:
            dist.barrier()
        idx = torch.cat([torch.arange(0, 4096), torch.arange(n - 4096, n), torch.arange(0, n, max(n // 4096, 1))]).unique().to(dev)
        af, bf, cf = F.to_flat(a), F.to_flat(b), F.to_flat(c)           # element order j, whatever the HBM layout
        ha = np.ascontiguousarray(af[:, idx].cpu().numpy().view(np.uint64))
        hb = np.ascontiguousarray(bf[:, idx].cpu().numpy().view(np.uint64))
        hc = np.empty_like(ha)
        oracle.fn("batch_modmul", "X25519")(vp(ha), vp(hb), vp(hc), ha.shape[1], ha.shape[1])
        ok_mul = bool(np.array_equal(cf[:, idx].cpu().numpy().view(np.uint64), hc))
        del af, bf, cf
        verified = {"modmul_elements_per_rank": int(idx.numel())}
        ok_lad = True
        if ladder is not None:
            m = k.shape[0]
            lidx = torch.cat([torch.arange(0, 4096), torch.arange(m - 4096, m), torch.arange(0, m, max(m // 2048, 1))]).unique().to(dev)
            hk = np.ascontiguousarray(k[lidx].cpu().numpy()); hu = np.ascontiguousarray(u[lidx].cpu().numpy())
            ho = np.empty_like(hu)
            oracle.lib.oracle_parallel(3, vp(hk), vp(hu), vp(ho), hk.shape[0], 0, max(1, len(os.sched_getaffinity(0)) // world))
            ok_lad = bool(np.array_equal(o[lidx].cpu().numpy(), ho))
            verified["x25519_records_per_rank"] = int(lidx.numel())
        my_ok = ok_mul and ok_lad
        all_ok = my_ok
        if use_dist:
            tt = torch.tensor([1 if my_ok else 0], dtype=torch.int32, device=cdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MIN)
            all_ok = bool(int(tt[0]))
        verified.update(all_ranks_equal_oracle=all_ok, ranks_checked=world)
        if single:
            assert ok_mul, "modmul differs from the oracle"
            assert ok_lad, "rfc7748 differs from the oracle"

    # per-rank inventory: which device each rank ran on and what it measured itself
    props = torch.cuda.get_device_properties(dev)
    bus = getattr(props, "pci_bus_id", None)
    mine = {"rank": rank, "local_rank": local, "device_index": dev.index, "device_name": torch.cuda.get_device_name(dev),
            "pci_bus_id": ("%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), bus, getattr(props, "pci_device_id", 0))) if bus is not None else None,
            "uuid": str(getattr(props, "uuid", "")) or None,
            "modmul_per_s": n * args.steps / my_dt, "kernel_ms": my_kern_ms, "hbm_GBps": BYTES_PER_MODMUL * n / (my_kern_ms * 1e-3) / 1e9,
            "x25519_per_s": (k.shape[0] / my_lt) if my_lt else None, "x25519_records": k.shape[0] if my_lt else None,
            "x25519_shard": ([lo, hi] if (my_lt and args.scaling == "strong") else None), "verified_against_oracle": my_ok}
    ranks = [mine]
    dist_info = None
    if use_dist:
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)
        dist_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                     "distinct_devices": len({(r["pci_bus_id"], r["uuid"], r["device_index"]) for r in ranks})}

    # N = 1: the PMC passes for roofline.traffic, as child processes AFTER every timed region of this process (they are ordinary
    # subprocesses -- nothing that has touched HIP is replaced or re-executed -- and cannot disturb the placement or the timing of
    # the launches measured above)
    live_traffic = None
    if (world == 1 and args.gpus == 1 and not args.no_traffic and os.environ.get("MA_BENCH_CHILD") != "1"
            and os.environ.get("MA_BENCH_TRAFFIC", "1") != "0"):
        live_traffic = measure_traffic()

    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu:
            import numpy as np
            cpu = cpu_baseline(np.ascontiguousarray(F.to_flat(a).cpu().numpy().view(np.uint64)), np.ascontiguousarray(F.to_flat(b).cpu().numpy().view(np.uint64)))
        # HBM bytes per launch from the PMC passes (FETCH_SIZE, WRITE_SIZE; tools/gpu_profile.sh) of the same command on
        # the same build: a property of the kernel and the batch size, so it is only quoted for the profiled size
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic_modmul_X25519.json")
        live_failed = live_traffic.get("failed") if live_traffic else None
        if live_traffic is not None and not live_failed:   # (the children inherit this process's environment: same batch size and layout)
            traffic, traffic_source = live_traffic["hbm_bytes_per_launch"], live_traffic["source"] + " (%d launches)" % live_traffic["launches_sampled"]
        elif os.path.exists(tpath):
            tdoc = json.load(open(tpath))
            if tdoc.get("algorithmic_bytes_per_launch") == BYTES_PER_MODMUL * n:
                traffic = tdoc.get("hbm_bytes_per_launch")
                traffic_source = "profiles/traffic_modmul_X25519.json (rocprofv3 --pmc passes, tag %s)" % tdoc.get("tag")
                if live_failed:
                    traffic_source += "; the live measurement of this run FAILED (%s)" % live_failed
        per_rank = [r["modmul_per_s"] for r in ranks]
        kms = [r["kernel_ms"] for r in ranks]
        med = sorted(probe_rates)[len(probe_rates) // 2] if probe_rates else None      # upper median of the probe rates
        if probe_rates and len(probe_rates) % 2 == 0:
            med = 0.5 * (sorted(probe_rates)[len(probe_rates) // 2 - 1] + sorted(probe_rates)[len(probe_rates) // 2])
        detail = {
            "metric": "256-bit modmul/s (2^255-19)", "value": value, "unit": "modmul/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": "batched modmul 2^255-19, 5x51-bit limbs, 2^%d elements per GPU, limb-interleaved SoA, %s" % (LOG2_ELEMS, ("tiles of %d elements [n/%d][5][%d]" % (TILE, TILE, TILE)) if TILE else "flat rows [5][n]"),
                       "layout": {"tile": TILE, "note": "limb i of element j at buf[((j / tile) * 5 + i) * tile + j % tile]; tile = 0: flat buf[i * n + j]; data_sets.other_layout_* times the other form"},
                       "elements_per_gpu": n, "placement_probe_GBps": [round(r, 1) for r in probe_rates], "inputs": "uniform mod p: splitmix64 stream (seed 42, array id, j) reduced mod p, generated on the device",
                       "placement_policy": ("timed on the %s of %d probed operand placements (identical contents), first-allocated first" % ("fastest" if keep_best else "FIRST-allocated", placements)) if placements > 1 else "single placement, no probe",
                       "parallelism": "independent batches, %d rank(s), no data-path collective" % world},
            # the same quantity from the placement probe (10 launches each, rank 0): median and first-allocated placement
            "value_median_placement": (med * 1e9 / BYTES_PER_MODMUL * world) if med else None,
            "value_first_placement": (probe_rates[0] * 1e9 / BYTES_PER_MODMUL * world) if probe_rates else None,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_source, "kernel": "k_binary<P_X25519,OpMulAuto,2>", "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_launch": BYTES_PER_MODMUL * n,
                         "frac_median_placement": (med / HBM_PEAK_GBS) if med else None,
                         "frac_first_placement": (probe_rates[0] / HBM_PEAK_GBS) if probe_rates else None},
            "cpu_baseline": cpu,
            "x25519": ladder,
            "x448": x448,
            # per-launch durations of a separate pass of the same launches (SURVEY 8(d) "report median"); ms_per_step above is the contract's mean
            "ms_per_step_median": launch_stats["median_ms"], "ms_per_step_min": launch_stats["min_ms"], "ms_per_step_max": launch_stats["max_ms"],
            "launch_stats": launch_stats,
            "data_sets": data_sets,
            "other_configs": others,
            "verified_against_oracle": verified,
            "ranks": ranks,
            "rank_spread": {"modmul_per_s": {"min": min(per_rank), "mean": sum(per_rank) / len(per_rank), "max": max(per_rank)},
                            "kernel_ms": {"min": min(kms), "mean": sum(kms) / len(kms), "max": max(kms)}},
            "dist": dist_info,
        }
        # Everything measured goes to the detail file beside this script (MA_BENCH_DETAIL names another path) and to stdout lines that
        # start with "# "; the LAST stdout line is the compact contract line (<= LINE_LIMIT bytes), the only line that starts with "{".
        detail_path = os.environ.get("MA_BENCH_DETAIL") or os.path.join(ROOT, DETAIL_NAME)
        detail_note = os.path.basename(detail_path)
        try:
            with open(detail_path, "w") as f:
                json.dump(detail, f, indent=1)
        except OSError as ex:                                # a read-only tree must not cost the line
            detail_note = "not written (%s); see the '# detail' lines of stdout" % type(ex).__name__
        for dl in detail_lines(detail):
            print(dl)
        line = contract_line(detail, detail_note)
    # The JSON line must be the LAST thing on stdout.  RCCL writes a version banner ("RCCL version : ...", five lines) to the C
    # stdout of rank 0 when the communicator is made; redirected to a file or a pipe that buffer is flushed at exit -- after a line
    # printed from Python.  So: every rank flushes its C and Python streams, the ranks meet, the process group is torn down, the
    # streams are flushed again, and only then rank 0 prints the line.
    import ctypes
    libc = ctypes.CDLL(None)
    sys.stdout.flush()
    libc.fflush(None)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
        sys.stdout.flush()
        libc.fflush(None)
    if rank == 0:
        print(line, flush=True)


if __name__ == "__main__":
    main()
