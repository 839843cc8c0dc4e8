"""The shader clock a kernel sees WHILE a piece of work runs (include/modarith_amd.h modarith_amd_sclk_probe): a one-wave probe on a
second stream waits, then reads the shader-clock counter against the constant-rate wall clock over a window inside the work's run
time.  bench.py brackets every VALU-bound leg with it: a rate can then be told from the clock the part held while it was measured
(the MI355X moves between 1.9 and 2.4 GHz with the instruction mix, the box and the time since the last idle period)."""
from __future__ import annotations

import torch

from . import _lib

_state = {}


def _probe_state(dev):
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _state:
        L = _lib.load()
        _state[key] = {"lib": L, "khz": int(L.modarith_amd_wall_clock_khz()), "stream": torch.cuda.Stream(device=dev, priority=-1),
                       "buf": torch.zeros(2, dtype=torch.int64, device=dev)}
    return _state[key]


def clock_during(fn, expected_s: float, dev=None):
    """run fn() (which enqueues work on the current stream) with the probe beside it; the probe's window is the middle half of
    `expected_s`.  Returns (shader clock in GHz or None, fn's result).  Synchronises the device."""
    dev = dev or torch.device("cuda", torch.cuda.current_device())
    st = _probe_state(dev)
    if st["khz"] <= 0:
        out = fn()
        torch.cuda.synchronize()
        return None, out
    delay = max(50, int(expected_s * 0.25e6))
    window = max(100, int(expected_s * 0.5e6))
    st["buf"].zero_()
    torch.cuda.synchronize()
    with torch.cuda.stream(st["stream"]):
        _lib.check(st["lib"].modarith_amd_sclk_probe(st["buf"].data_ptr(), delay, window, st["stream"].cuda_stream), "sclk_probe")
    out = fn()
    torch.cuda.synchronize()
    cyc, ticks = (int(v) for v in st["buf"].tolist())
    ghz = (cyc / ticks * st["khz"] * 1e3 / 1e9) if ticks > 0 else None
    return ghz, out


def timed_with_clock(fn, reps: int = 3, warm: int = 2):
    """(median seconds of `reps` event-timed calls after `warm` calls, shader clock in GHz during one further call, last result)"""
    out = None
    for _ in range(warm):
        out = fn()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e-3)
    t = sorted(ts)[len(ts) // 2]
    ghz, out = clock_during(fn, t)
    return t, ghz, out
