"""Exhaustive check behind ln_stream.h ln_std10: for every var in [0, 2^24) the reference's ten float32 Newton steps
(ivit_modules.py:45-49) against floor(sqrt(var)).  Prints the smallest V0 such that t10 == isqrt(var) for all var >= V0 that are
not of the form (s + 1)^2 - 1, and what t10 is on those."""
import numpy as np
f32 = np.float32
v = np.arange(0, 1 << 24, dtype=np.int64)
vf = v.astype(f32)
t = np.full(v.shape, 65536.0, dtype=f32)
for it in range(10):
    t = np.floor((t + np.floor(vf / t)) * f32(0.5)).astype(f32)
t10 = t.astype(np.int64)
s = np.floor(np.sqrt(v.astype(np.float64))).astype(np.int64)
s = np.where(s * s > v, s - 1, s); s = np.where((s + 1) * (s + 1) <= v, s + 1, s)
special = (s + 1) ** 2 - 1 == v
bad = np.nonzero((t10 != s) & ~special)[0]
V0 = int(bad.max()) + 1
sp = np.nonzero(special & (v >= V0))[0]
print("V0 =", V0, "(ln_stream.h LN_NEWTON_CONVERGED must be >= this)")
print("special var >= V0:", len(sp), "t10 - isqrt in", np.unique(t10[sp] - s[sp], return_counts=True))
assert V0 <= 142883
