#!/bin/bash
# round 5: A/B of the fused ED25519 kernels, window form against ladder form (profiles/r05_lad_ab.log also holds the occupancy variants of the first run: a 128-register build, an LDS claim for two waves per SIMD), on one box
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
{
for lg in 20 21; do
  LOG2N=$lg MA_ED25519_FUSED=window python tools/ed25519_lad_ab.py
  LOG2N=$lg python tools/ed25519_lad_ab.py
done
} 2>&1 | tee gpurun_out/r05_lad_ab.log
