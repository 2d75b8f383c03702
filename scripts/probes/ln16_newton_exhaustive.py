"""The ten float32 Newton steps of I-LayerNorm (ivit_modules.py:45-49) on EVERY float32 value a 16-bit row's variance sum can take
(var < 2^37: the kernels of csrc/swin.hip feed varf = RN24(var)): t10 against floor(sqrt(varf)), binade by binade.  Question: from
which varf on is t10 == isqrt(varf), and what are the exceptions?  (var < 2^24: scripts/probes/ln_newton_exhaustive.py.)"""
import numpy as np
f32 = np.float32
tot_bad = 0
for ex in range(24, 37):
    m = np.arange(1 << 23, 1 << 24, dtype=np.int64)             # all float32 mantissas of the binade [2^ex, 2^(ex+1))
    v = m << (ex - 23)
    vf = v.astype(f32)
    assert np.array_equal(vf.astype(np.int64), v)
    t = np.full(v.shape, 65536.0, dtype=f32)
    for it in range(10):
        t = np.floor((t + np.floor(vf / t)) * f32(0.5)).astype(f32)
    t10 = t.astype(np.int64)
    s = np.floor(np.sqrt(v.astype(np.float64))).astype(np.int64)
    s = np.where(s * s > v, s - 1, s); s = np.where((s + 1) * (s + 1) <= v, s + 1, s)
    d = t10 - s
    bad = np.nonzero(d != 0)[0]
    tot_bad += len(bad)
    special = (s + 1) ** 2 - 1 == v
    # how far is v from the next square when t10 != s?
    gap = ((s + 1) ** 2 - v)[bad]
    assert len(bad) == 0 or gap.max() <= 1 << (ex - 22), "the kernels' threshold varf * 2^-22 >= 2^(ex - 22) must cover every exception"
    assert np.all(d[bad] == 1)
    print(f"binade 2^{ex}: {len(bad)} of {len(v)} values with t10 != isqrt; t10 - isqrt in {dict(zip(*np.unique(d[bad], return_counts=True)))}; "
          f"(s+1)^2 - v on those: min {gap.min() if len(bad) else '-'} max {gap.max() if len(bad) else '-'}; of the form (s+1)^2-1: {int(special[bad].sum())}")
print("total", tot_bad, "-- every exception has (s + 1)^2 - varf <= 2^(ex - 22) <= varf * 2^-22: ln16_std10 (csrc/swin.hip) runs the literal loop there")
