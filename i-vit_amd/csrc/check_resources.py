#!/usr/bin/env python3
"""Build-time guard for the inline-asm GEMM kernels (csrc/Makefile runs it on the remarks of
`hipcc -Rpass-analysis=kernel-resource-usage`).

gemm_i8_wreg_kernel / gemm_i8_pers_kernel issue global loads, LDS-DMA and ds_reads through inline asm and tie each
destination register to its consumer only by a later asm `s_waitcnt` with "+v" operands.  That is sound only while the
register allocator never spills or copies those registers between the issue and the wait, i.e. while the kernel fits its
register budget: no scratch, no VGPR spills, at most 256 VGPRs (two workgroups of four waves per CU).  A toolchain or flag
change that breaks the budget must fail the build here, on the CPU, not on the GPU."""
import re
import sys

GUARDED = ("gemm_i8_wreg_kernel", "gemm_i8_pers_kernel")
FIELDS = {"VGPRs": r"\bVGPRs: (\d+)", "AGPRs": r"AGPRs: (\d+)", "Scratch": r"ScratchSize \[bytes/lane\]: (\d+)",
          "Occupancy": r"Occupancy \[waves/SIMD\]: (\d+)", "VGPRSpill": r"VGPRs Spill: (\d+)", "SGPRSpill": r"SGPRs Spill: (\d+)",
          "LDS": r"LDS Size \[bytes/block\]: (\d+)"}


def parse(text):
    kernels, cur = {}, None
    for line in text.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        if cur is None:
            continue
        for k, pat in FIELDS.items():
            m = re.search(pat, line)
            if m and k not in cur:
                cur[k] = int(m.group(1))
    return kernels


def main(path):
    kernels = parse(open(path, errors="replace").read())
    bad, seen = [], 0
    for name, r in sorted(kernels.items()):
        if not any(g in name for g in GUARDED):
            continue
        seen += 1
        regs = r.get("VGPRs", 0) + r.get("AGPRs", 0)
        line = (f"{name}: VGPR+AGPR {regs}, scratch {r.get('Scratch')}, VGPR spill {r.get('VGPRSpill')}, "
                f"SGPR spill {r.get('SGPRSpill')}, occupancy {r.get('Occupancy')}, LDS {r.get('LDS')}")
        ok = (r.get("Scratch") == 0 and r.get("VGPRSpill") == 0 and regs <= 256 and r.get("Occupancy", 0) >= 2
              and r.get("LDS", 0) <= 81920)
        print(("ok   " if ok else "FAIL ") + line)
        if not ok:
            bad.append(name)
    if seen == 0:
        print("check_resources: no guarded kernel found in", path)
        return 1
    if bad:
        print("check_resources: register / scratch budget of the inline-asm GEMM kernels violated:", ", ".join(bad))
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
