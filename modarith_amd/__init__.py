"""modarith_amd -- MI355X-native batched finite-field engine behind modarith's field.c API.

Only what the hot path needs lives here: `params` / `emit` (the per-prime constant driver),
`csrc/` (HIP kernels + the C-ABI shim of include/modarith_amd.h), `field` (host-side mirror of the
reference interface, batched), `dist` (sharding of independent batches across ranks).
Importing this package does not touch the GPU; `Field(...)` loads the HIP library and fails loudly
if it has not been built.
"""
__all__ = ["params", "emit"]
__version__ = "0.1.0"
