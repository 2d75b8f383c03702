"""classify the bytes where the fused fc1 + ShiftGELU differs from GEMM + table pass (debugging aid)"""
import os, sys
os.environ.setdefault("IVIT_USE_LAB_LIBRARY", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import ivit_amd
from ivit_amd import _lib
from ivit_amd.prepare import dyadic
DEV = "cuda:0"
M, N, K = 256 * 197, 3072, 768
g = torch.Generator(device="cpu").manual_seed(5)
A = torch.randint(-128, 128, (M, K), dtype=torch.int8, generator=g).to(DEV)
W = torch.randint(-128, 128, (N, K), dtype=torch.int8, generator=g)
W[:, ::3] //= 8
W = W.to(DEV)
b = torch.randint(-50000, 50000, (N,), dtype=torch.int32, generator=g).to(DEV)
rng = np.random.default_rng(3)
m = torch.from_numpy(rng.integers(2 ** 30, 2 ** 31, N).astype(np.int64).astype(np.uint32).view(np.int32)).to(DEV)
e = torch.from_numpy((31 + rng.integers(13, 18, N)).astype(np.int32)).to(DEV)
s_g = np.float32(0.0517)
mg, eg = dyadic(np.float32(s_g * np.float32(1 / 128)), np.float32(0.011))
st = _lib.stream_ptr()
lut = torch.empty(65536, dtype=torch.int8, device=DEV)
_lib.call("ivit_shiftgelu_build_lut_ex", float(s_g), int(mg[0]), int(eg[0]), None, _lib.ptr(lut), st)
Wf = torch.zeros(N * K, dtype=torch.int8, device=DEV)
_lib.call("ivit_pack_weight_frags16_i8", _lib.ptr(W), K, N, K, _lib.ptr(Wf), st)
LAY = 16                      # row-major everything: easy indices
raw = torch.zeros(M, N, dtype=torch.int8, device=DEV)
_lib.call("ivit_gemm_i8_requant_ex", _lib.ptr(A), K, _lib.ptr(Wf), K, _lib.ptr(b), _lib.ptr(m), _lib.ptr(e), _lib.ptr(raw), N, M, N, K, LAY, st)
ref = torch.empty_like(raw)
_lib.call("ivit_shiftgelu_lut_i8_ex", _lib.ptr(raw), N, M, N, _lib.ptr(lut), _lib.ptr(ref), N, 0, st)
ws = torch.zeros((M + 127) // 128, dtype=torch.int32, device=DEV)
VARIANTS = [0x10000, 0x10000]
seen = torch.zeros(M, N, dtype=torch.int8, device=DEV)
_lib.call("ivit_debug_set_stamp_buffer", _lib.ptr(seen))
for it, fl2 in enumerate(VARIANTS):
    _lib.call("ivit_debug_set_gemm_flags2", fl2)
    print("flags2 = %#x" % fl2)
    out = torch.full((M, N), 3, dtype=torch.int8, device=DEV)
    _lib.call("ivit_gemm_i8_requant_gelu_ex", _lib.ptr(A), K, _lib.ptr(Wf), K, _lib.ptr(b), _lib.ptr(m), _lib.ptr(e), _lib.ptr(lut),
              _lib.ptr(ws), _lib.ptr(out), N, M, N, K, LAY, st)
    bad = (out != ref)
    nb = int(bad.sum())
    print(f"launch {it}: {nb} bytes differ; workspace max {int(ws.abs().max())}")
    if nb:
        idx = bad.nonzero()
        rows, cols = idx[:, 0].cpu().numpy(), idx[:, 1].cpu().numpy()
        o, r, k = out[bad].cpu().numpy(), ref[bad].cpu().numpy(), raw[bad].cpu().numpy()
        print("  equal to the raw k:", int((o == k).sum()), " equal to 3 (never written):", int((o == 3).sum()))
        print("  distinct rows", len(np.unique(rows)), " panels", len(np.unique(rows // 128)), " channel tiles", np.unique(cols // 256)[:12])
        pr = np.unique(rows // 128)
        print("  panels (first 10):", pr[:10], " rows mod 128 (first 10):", np.unique(rows % 128)[:10], " n rows-in-panel", len(np.unique(rows % 128)))
        # are whole 16-byte chunks wrong?
        ch = np.unique(rows.astype(np.int64) * (N // 16) + cols // 16)
        print("  16-byte chunks touched:", len(ch), " bytes per chunk: %.1f" % (nb / len(ch)))
        # would another row maximum explain it?
        rm = raw.max(dim=1).values.cpu().numpy()
        l2 = lut.cpu().numpy().reshape(256, 256)
        r0 = rows[0]
        cand = [mx for mx in range(-128, 128) if np.array_equal(l2[mx + 128, raw[r0].cpu().numpy().astype(np.int32) + 128][cols[rows == r0]], o[rows == r0])]
        print("  row", r0, "true max", rm[r0], "maxima that explain its wrong bytes:", cand[:8])
        sd = (seen != raw)
        print("  bytes the completing workgroups READ that differ from the raw GEMM output:", int(sd.sum()),
              " of which at positions whose output is wrong:", int((sd & bad).sum()), " wrong outputs at correctly read positions:", int((bad & ~sd).sum()))
        if int(sd.sum()):
            i2 = sd.nonzero()[:12].cpu().numpy()
            print("   first misread positions (row, col, read, raw):", [(int(a), int(b), int(seen[a, b]), int(raw[a, b])) for a, b in i2])
        rr = raw[r0].cpu().numpy().astype(np.int32)
        oo = out[r0].cpu().numpy().astype(np.int32)
        fx = ref[r0].cpu().numpy().astype(np.int32)
        wc = np.nonzero(oo != fx)[0]
        print("  wrong columns of that row:", len(wc), wc[:24])
        for c in wc[:10]:
            ms = [mx for mx in range(-128, 128) if l2[mx + 128, rr[c] + 128] == oo[c] and mx >= rr[c]]
            print(f"    col {c}: k {rr[c]} ref {fx[c]} out {oo[c]}; row maxima that would give it: {ms[:6]}{'...' if len(ms) > 6 else ''}")
        # per 256-column tile: is the output the table row of ONE other maximum?
        for tn in range(12):
            seg = slice(256 * tn, 256 * tn + 256)
            ms = [mx for mx in range(int(rr[seg].max()), 128) if np.array_equal(l2[mx + 128, rr[seg] + 128], oo[seg])]
            print(f"    channel tile {tn}: max of the tile {rr[seg].max()}, maxima consistent with the whole tile's output: {ms[:5]}")
        # the neighbouring rows
        for dr in (-16, -4, -1, 1, 4):
            r1 = r0 + dr
            print(f"    row {r1}: wrong bytes {int((out[r1] != ref[r1]).sum())}, true max {rm[r1]}")
