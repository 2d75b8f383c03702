#!/usr/bin/env python3
"""Build-time guard for the inline-asm GEMM kernels (csrc/Makefile runs it on the remarks of
`hipcc -Rpass-analysis=kernel-resource-usage`).

gemm_i8_wreg_kernel / gemm_i8_pers_kernel issue global loads, LDS-DMA and ds_reads through inline asm and tie each
destination register to its consumer only by a later asm `s_waitcnt` with "+v" operands.  That is sound only while the
register allocator never spills or copies those registers between the issue and the wait, i.e. while the kernel fits its
register budget: no scratch, no VGPR spills, at most 256 VGPRs (two workgroups of four waves per CU).  A toolchain or flag
change that breaks the budget must fail the build here, on the CPU, not on the GPU.

"No spill" is a NECESSARY condition, not a sufficient one: the compiler may still copy such a register between the load and the
wait without spilling.  csrc/check_isa.py checks the emitted instruction stream for exactly that; the bit-exact GEMM tests on the
GPU remain the proof of the result.

Second use (`check_resources.py <remarks> attention|rowops|swin`): the occupancy the launchers of attention.hip / rowops.hip / swin.hip assume
when they size their grids (occupancy_rules below)."""
import re
import sys

GUARDED = ("gemm_i8_wreg_kernel", "gemm_i8_pers_kernel", "gemm_i8_wp_kernel")
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


def occupancy_rules(path, what):
    """attention.hip / rowops.hip: kernels whose launchers size their grids for a number of resident workgroups.  The number is a
    template argument (the OCC of attention_kernel<MODE, PB, OCC, ...>, the last argument of layernorm_i8_stream_kernel<LPR, NC, NG,
    COMPAT, OCC>): the compiler must reach it (waves per SIMD) and, for the variants on the headline path, without scratch."""
    kernels = parse(open(path, errors="replace").read())
    bad, seen = [], 0
    for name, r in sorted(kernels.items()):
        if what == "attention":
            m = re.search(r"attention_kernelILi(\d+)ELi(\d+)ELi(\d+)ELb([01])E", name)
            if not m:
                continue
            # scratch-free: the forms of 193-208 tokens (14 x 14 patches + cls: every BASELINE config); the general-T forms
            # (other geometries, "not tuned") may keep a few dwords of the Q prefetch in scratch
            occ, need_no_scratch = int(m.group(3)), m.group(4) == "0"
        elif what == "swin":
            if "window_attention_kernel" in name:
                occ, need_no_scratch = 4, True
            else:
                m = re.search(r"layernorm_i16_i8_tiled(_compat)?_kernelILi(\d+)ELi(\d+)E", name)
                if not m:
                    continue
                occ, need_no_scratch = (3 if m.group(1) and int(m.group(3)) >= 2 else 4), True      # = their __launch_bounds__
        else:
            m = re.search(r"layernorm_i8_stream_kernelILi(\d+)ELi(\d+)ELi(\d+)ELb([01])ELi(\d+)E", name)
            if not m:
                continue
            occ = int(m.group(5))
            # scratch-free: every plain form, and the natural-scale form of the widths the BASELINE configs run (C = 768, 384, 192);
            # the natural-scale C = 1024 form spills a few loop-invariant dwords into cold paths (ln_stream.h)
            need_no_scratch = m.group(4) == "0" or int(m.group(2)) == 3
        seen += 1
        ok = r.get("Occupancy", 0) >= occ and (r.get("Scratch") == 0 or not need_no_scratch)
        print(("ok   " if ok else "FAIL ") + f"{name}: VGPR {r.get('VGPRs')}, scratch {r.get('Scratch')}, occupancy {r.get('Occupancy')} (launcher assumes {occ}"
              f"{', no scratch' if need_no_scratch else ''})")
        if not ok:
            bad.append(name)
    if seen == 0:
        print("check_resources: no", what, "kernel found in", path)
        return 1
    if bad:
        print("check_resources: occupancy / scratch assumption of the launchers violated:", ", ".join(bad))
        return 1
    return 0


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
    sys.exit(occupancy_rules(sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else main(sys.argv[1]))
