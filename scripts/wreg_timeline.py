"""Stamped timeline of the weights-in-registers GEMM: shader cycles per tile phase and per K step, wave 0 of every workgroup.
  32x32x32 form (lab flag 16; `one`: one workgroup per CU):   wreg_timeline.py [qkv|proj|fc1|fc2] [one]
  16x16x64 form, int8 or fused-residual epilogue, with the epilogue split into its phases (flags2 bit 8):
                                                              wreg_timeline.py [shape] --s16 [--resid]"""
import os; os.environ.setdefault("IVIT_USE_LAB_LIBRARY", "1")
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ivit_amd  # noqa: F401
from ivit_amd import _lib
DEV = "cuda:0"; M = 197 * 256
names = [a for a in sys.argv[1:] if a in ("qkv", "proj", "fc1", "fc2")] or ["fc1"]
one = "one" in sys.argv
S16 = "--s16" in sys.argv
RESID = "--resid" in sys.argv
rng = np.random.default_rng(0)
for name in names:
    N, K = {"qkv": (2304, 768), "proj": (768, 768), "fc1": (3072, 768), "fc2": (768, 3072)}[name]
    A = torch.from_numpy(rng.integers(-128, 128, size=(M, K)).astype(np.int8)).to(DEV)
    W = torch.from_numpy(rng.integers(-128, 128, size=(N, K)).astype(np.int8)).to(DEV)
    At, Wf = torch.empty_like(A), torch.empty_like(W)
    _lib.call("ivit_tile_operand_i8", _lib.ptr(A), K, M, K, _lib.ptr(At), _lib.stream_ptr())
    _lib.call("ivit_pack_weight_frags16_i8" if S16 else "ivit_pack_weight_frags_i8", _lib.ptr(W), K, N, K, _lib.ptr(Wf), _lib.stream_ptr())
    b = torch.zeros(N, dtype=torch.int32, device=DEV); m = torch.full((N,), (1 << 30) + 12345, dtype=torch.int32, device=DEV)
    e = torch.full((N,), 42, dtype=torch.int32, device=DEV); out = torch.empty(M, N, dtype=torch.int8, device=DEV)
    res = torch.from_numpy(rng.integers(-128, 128, size=(M, N)).astype(np.int8)).to(DEV)
    stamps = torch.zeros(512 * 4 * 32, dtype=torch.int64, device=DEV)
    _lib.call("ivit_debug_set_stamp_buffer", _lib.ptr(stamps))
    if S16:
        _lib.call("ivit_debug_set_gemm_flags2", 256)
    else:
        _lib.call("ivit_debug_set_gemm_flags", 16 | (4096 if one else 0))
    lay = 17 if S16 else 9
    for _ in range(3):
        if RESID:
            _lib.call("ivit_gemm_i8_requant_residual_ex", _lib.ptr(At), K, _lib.ptr(Wf), K, _lib.ptr(b), _lib.ptr(m), _lib.ptr(e),
                      _lib.ptr(res), N, 1610612736, 31, 1073741824, 32, _lib.ptr(out), N, M, N, K, lay, _lib.stream_ptr())
        else:
            _lib.call("ivit_gemm_i8_requant_ex", _lib.ptr(At), K, _lib.ptr(Wf), K, _lib.ptr(b), _lib.ptr(m), _lib.ptr(e), _lib.ptr(out), N, M, N, K, lay, _lib.stream_ptr())
    torch.cuda.synchronize()
    _lib.call("ivit_debug_set_gemm_flags", 0); _lib.call("ivit_debug_set_gemm_flags2", 0)
    s = stamps.cpu().numpy().reshape(512, 4, 32)
    nb = 256 if one else 512
    print(f"== {name} N={N} K={K} {'16x16x64' if S16 else '32x32x32'} {'residual epilogue' if RESID else 'int8 epilogue'}")
    for it in (0, 1, 2):
        t = s[:nb, it]
        ok = t[:, 3] > 0
        t = t[ok]
        d = np.stack([t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 3] - t[:, 0]], 1)
        print(f"tile {it}: median over {ok.sum()} workgroups  start->loop {np.median(d[:,0]):.0f}  loop {np.median(d[:,1]):.0f}  epilogue {np.median(d[:,2]):.0f}  total {np.median(d[:,3]):.0f}")
        if S16:
            print(f"   epilogue: phase 1 (requant -> LDS) {np.median(t[:, 20] - t[:, 2]):.0f}  barrier {np.median(t[:, 21] - t[:, 20]):.0f}  "
                  f"LDS reads + vmcnt(0) {np.median(t[:, 22] - t[:, 21]):.0f}  chunk arithmetic + stores {np.median(t[:, 3] - t[:, 22]):.0f}")
        if it + 1 < 4:
            nxt = s[:nb, it + 1, 0][ok]
            good = nxt > 0
            if good.any():
                print(f"   tile start to next tile start {np.median((nxt - t[:, 0])[good]):.0f}")
    # whole launch: first tile start to the last tile end, per workgroup, by the number of tiles it ran
    ntl = (s[:nb, :, 3] > 0).sum(axis=1)
    for k in (2, 3):
        sel = ntl == k
        if sel.any():
            span = s[:nb, :, 3].max(axis=1)[sel] - s[:nb, 0, 0][sel]
            print(f"workgroups with {k} tiles: {sel.sum()}, first start to last end: median {np.median(span):.0f} cycles")
    if not one and nb == 512:
        # phase of the two workgroups of a CU (b and b + 256: scripts/probes/xcc_map_probe.hip): which share of the time that at least
        # one of them spends in a main loop do BOTH spend in one (1 = in phase: the MFMA pipe idles during the epilogues; 0 = they alternate)
        both, any_ = [], []
        for bb in range(256):
            iv = []
            for w in (bb, bb + 256):
                for it in range(4):
                    if s[w, it, 3] > 0:
                        iv.append((w, int(s[w, it, 1]), int(s[w, it, 2])))
            if len({w for w, _, _ in iv}) < 2:
                continue
            t0 = max(min(a for w, a, _ in iv if w == bb), min(a for w, a, _ in iv if w == bb + 256))       # both have started
            t1 = min(max(e for w, _, e in iv if w == bb), max(e for w, _, e in iv if w == bb + 256))       # neither has run out of stamped tiles
            if t1 <= t0:
                continue
            ev = sorted([(max(a, t0), 1) for _, a, e in iv if e > t0 and a < t1] + [(min(e, t1), -1) for _, a, e in iv if e > t0 and a < t1])
            depth, last, b2, a1 = 0, t0, 0, 0
            for t, d in ev:
                if depth >= 1:
                    a1 += t - last
                if depth >= 2:
                    b2 += t - last
                depth += d
                last = t
            if a1:
                both.append(b2 / a1)
                any_.append(a1 / (t1 - t0))
        if both:
            print(f"CU pairs (b, b + 256), first four tiles: both workgroups in a main loop for {np.median(both):.2f} of the time at least one is "
                  f"(quartiles {np.percentile(both, 25):.2f} / {np.percentile(both, 75):.2f}); some main loop running {np.median(any_):.2f} of the time")
    ok = (s[:nb, 3, 0] > 0)
    dt = (s[:nb, 3, 0] - s[:nb, 0, 0]).astype(np.float64)[ok]
    dr = (s[:nb, 3, 16] - s[:nb, 0, 16]).astype(np.float64)[ok]
    if ok.any():
        print(f"s_memtime ticks per s_memrealtime tick (100 MHz): median {np.median(dt / dr):.2f}  -> counter rate {np.median(dt / dr) * 100:.0f} MHz")
    for blk in (0, 1, 100):
        t = s[blk, 1]
        steps = np.diff(t[4:16])
        print(f"workgroup {blk} tile 1: K-step durations {steps.tolist()}  loop {t[2] - t[1]}  epilogue {t[3] - t[2]}")
