#!/usr/bin/env python3
"""Build-time lint of the emitted gfx950 ISA for the kernels that issue ASYNCHRONOUS loads through inline asm
(csrc/Makefile runs it on `hipcc -S --cuda-device-only` output of gemm.hip; tests/test_host_logic.py feeds it broken streams).

gemm_i8_wreg_kernel / gemm_i8_pers_kernel load weight fragments (`global_load_dwordx4`) and token fragments (`ds_read_b128`)
with inline asm and tie each destination register to its consumer only by a later asm `s_waitcnt`.  The compiler sees neither
the load nor the wait: it will not insert a wait of its own, and nothing stops it from scheduling an instruction that touches a
destination register between the two.  check_resources.py guards the necessary condition (no spills, register budget); this
file checks the property itself, on the final instruction stream:

  (i)   no instruction reads or writes a VGPR that an inline-asm load has in flight: a register is in flight from the load
        until an `s_waitcnt` whose count proves the load complete -- vmcnt(N) / lgkmcnt(N) retire, in issue order, every
        operation of that counter with at least N younger ones.  Every vector-memory operation (loads, stores, LDS-DMA) counts
        for vmcnt, every DS operation for lgkmcnt, whoever issued it.  The analysis is a forward dataflow over the kernel's
        basic blocks (state: register -> number of younger operations, the MINIMUM over all paths), iterated to a fixed point,
        so loads that stay in flight around the main loop's back edge are followed.
  (i-b) no scalar memory load (SMEM returns out of order and shares lgkmcnt) is issued while an inline-asm ds_read is in flight:
        the counted lgkmcnt waits would no longer mean what they say.
  (ii)  an accumulator written by an INLINE-ASM v_mfma (tied accumulators: the hazard recognizer does not see them) is not read
        by another VALU instruction within MFMA_VALU_WAIT wait states, and is not the accumulator of another inline-asm MFMA closer
        than MFMA_MFMA_GAP instructions (MFMAs the compiler issued itself are its hazard recognizer's business);
  (iii) an inline-asm vector store is preceded, inside its asm statement, by an s_nop (its data comes straight out of
        v_permlane*_swap in the epilogue; the compiler pads its own stores, not these).
  (iv)  no VALU instruction writes an operand register of an inline-asm v_mfma within VALU_MFMA_GAP instructions in front of it
        (round 4, measured: `v_mov_b64 acc, bias` sunk by the scheduler to just in front of a tile's first MFMA left half of the C
        operand's dwords stale, by lane parity -- gemm_wp.h's first parity run; the hazard recognizer does not look into inline asm).

Exit status 1 and one line per finding if any guarded kernel violates a rule; the guarded kernels found are listed otherwise."""
import re
import sys
from collections import defaultdict

GUARDED = ("gemm_i8_wreg_kernel", "gemm_i8_pers_kernel", "gemm_i8_wp_kernel")
MFMA_VALU_WAIT = 19      # wait states between an MFMA's write and a VALU read of the result (16-pass bound: covers every shape here)
MFMA_MFMA_GAP = 4        # instructions between two MFMAs on the same accumulator
VALU_MFMA_GAP = 4        # instructions between a VALU write of a register and an inline-asm MFMA that reads it
CAP = 63                 # counters saturate: vmcnt is 6 bits wide on gfx9

REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


class Inst:
    __slots__ = ("op", "args", "asm", "line", "text")

    def __init__(self, op, args, asm, line, text):
        self.op, self.args, self.asm, self.line, self.text = op, args, asm, line, text


def is_vmem(op):
    return op.startswith(("global_", "buffer_", "flat_", "scratch_", "tbuffer_")) and not op.startswith(("buffer_inv", "buffer_wbl2", "buffer_gl"))


def is_ds(op):
    return op.startswith("ds_")


def is_smem_load(op):
    return op.startswith(("s_load_", "s_buffer_load_", "s_scratch_load_"))


def split_kernels(text):
    """-> {name: [Inst or ('label', name)]} for every function whose symbol contains a guarded name"""
    kernels, cur, name, in_asm = {}, None, None, False
    for ln, raw in enumerate(text.splitlines(), 1):
        line = raw.split("//")[0].rstrip()
        s = line.strip()
        if not s:
            continue
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if s.startswith(";") or s.startswith("//"):
            continue
        m = re.match(r"^([A-Za-z_.$][\w.$]*):", s)
        if m and not raw.startswith(("\t", " ")):
            lab = m.group(1)
            if lab.startswith(".L"):
                if cur is not None:
                    cur.append(("label", lab))
            else:
                cur, name = ([], lab) if any(g in lab for g in GUARDED) else (None, None)
                if cur is not None:
                    kernels[name] = cur
            continue
        if s.startswith("."):
            if s.startswith((".end_amdhsa_kernel", ".section", ".text")) and cur is not None and s.startswith(".section"):
                cur = None
            continue
        if cur is None:
            continue
        s = s.split(";")[0].strip()
        if not s:
            continue
        parts = s.split(None, 1)
        cur.append(Inst(parts[0], parts[1] if len(parts) > 1 else "", in_asm, ln, s))
        if parts[0] == "s_endpgm":
            pass
    return kernels


def build_blocks(items):
    """basic blocks: list of (label or None, [Inst]); successors by index"""
    blocks, cur, lab = [], [], None
    for it in items:
        if isinstance(it, tuple):
            if cur or lab is not None:
                blocks.append((lab, cur))
            cur, lab = [], it[1]
        else:
            cur.append(it)
            if it.op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
                blocks.append((lab, cur))
                cur, lab = [], None
    if cur or lab is not None:
        blocks.append((lab, cur))
    index = {lab: i for i, (lab, _) in enumerate(blocks) if lab is not None}
    succ = []
    for i, (lab, insts) in enumerate(blocks):
        s = []
        last = insts[-1] if insts else None
        if last is not None and last.op.startswith("s_branch"):
            s.append(index.get(last.args.strip()))
        elif last is not None and last.op.startswith("s_cbranch"):
            s.append(index.get(last.args.strip()))
            s.append(i + 1)
        elif last is not None and last.op.startswith(("s_endpgm", "s_setpc")):
            pass
        else:
            s.append(i + 1)
        succ.append([x for x in s if x is not None and x < len(blocks)])
    return blocks, succ


def waits(inst):
    vm = re.search(r"vmcnt\((\d+)\)", inst.args)
    lg = re.search(r"lgkmcnt\((\d+)\)", inst.args)
    if inst.op == "s_waitcnt" and not vm and not lg and re.match(r"^\s*(0x[0-9a-fA-F]+|\d+)\s*$", inst.args or ""):
        imm = int(inst.args.strip(), 0)        # raw immediate: vmcnt = bits 3:0 | 15:14, lgkmcnt = bits 11:8
        return (imm & 15) | ((imm >> 14) & 3) << 4, (imm >> 8) & 15
    return (int(vm.group(1)) if vm else None), (int(lg.group(1)) if lg else None)


def wait_states(inst):
    if inst.op == "s_nop":
        try:
            return int(inst.args.strip(), 0) + 1
        except ValueError:
            return 1
    return 1


def transfer(state, inst, findings, kname, record):
    """state: {reg: (kind, n, origin line)}: kind 'vm' / 'lg' = an inline-asm load in flight with n younger operations of its
    counter, kind 'mf' = written by an inline-asm MFMA n wait states ago (n < MFMA_VALU_WAIT); n is the MINIMUM over all paths.
    Returns the new state (a dict)."""
    touches_vgpr = inst.op.startswith(("v_", "ds_", "global_", "buffer_", "flat_", "scratch_"))
    used = regs_of(inst.args) if touches_vgpr else set()
    asm_mfma = inst.op.startswith("v_mfma") and inst.asm
    if inst.op != "s_waitcnt" and record:
        hit = sorted(r for r in used if r in state and state[r][0] != "mf")
        if hit:
            findings.append(f"{kname}: line {inst.line}: `{inst.text}` touches v{hit[0]}"
                            f"{'..' if len(hit) > 1 else ''}{hit[-1] if len(hit) > 1 else ''} while the inline-asm "
                            f"{'global load' if state[hit[0]][0] == 'vm' else 'ds_read'} of line {state[hit[0]][2]} into it may still be in flight "
                            f"({state[hit[0]][1]} younger operation(s) on some path) (rule i)")
        mf = sorted(r for r in used if r in state and state[r][0] == "mf")
        if mf and asm_mfma:
            acc = regs_of(inst.args.split(",")[0])
            bad = [r for r in mf if r in acc and state[r][1] < MFMA_MFMA_GAP]
            if bad:
                findings.append(f"{kname}: line {inst.line}: `{inst.text}` accumulates into v{bad[0]} {state[bad[0]][1]} wait state(s) after the "
                                f"MFMA of line {state[bad[0]][2]} on it (rule ii: at least {MFMA_MFMA_GAP})")
        elif mf and inst.op.startswith("v_") and not inst.op.startswith("v_mfma"):
            findings.append(f"{kname}: line {inst.line}: `{inst.text}` uses v{mf[0]} {state[mf[0]][1]} wait state(s) after the inline-asm MFMA of "
                            f"line {state[mf[0]][2]} wrote it (rule ii: at least {MFMA_VALU_WAIT})")
    if inst.op == "s_waitcnt":
        vm, lg = waits(inst)
        st = {r: v for r, v in state.items() if not ((v[0] == "vm" and vm is not None and v[1] >= vm) or (v[0] == "lg" and lg is not None and v[1] >= lg))}
    else:
        st = state
    if is_smem_load(inst.op) and any(v[0] == "lg" for v in state.values()) and record:
        findings.append(f"{kname}: line {inst.line}: `{inst.text}` is issued while an inline-asm ds_read is in flight: SMEM shares lgkmcnt "
                        "and returns out of order (rule i-b)")
    if is_vmem(inst.op) or is_ds(inst.op):
        c = "vm" if is_vmem(inst.op) else "lg"
        st = {r: ((v[0], min(v[1] + 1, CAP), v[2]) if v[0] == c else v) for r, v in st.items()}
        is_load = ("load" in inst.op and "_lds_" not in inst.op and " lds" not in inst.args) or inst.op.startswith(("ds_read", "ds_load"))
        if inst.asm and is_load:
            first = inst.args.split(",")[0]
            for r in regs_of(first):
                st[r] = (c, 0, inst.line)
    # MFMA results age by the wait states of every instruction; a result old enough, overwritten or consumed leaves the state
    ws = wait_states(inst)
    if any(v[0] == "mf" for v in st.values()) or asm_mfma:
        nst = {}
        for r, v in st.items():
            if v[0] != "mf":
                nst[r] = v
            elif r in used and not asm_mfma:
                continue                                  # consumed (reported above if too early) or overwritten
            elif v[1] + ws < MFMA_VALU_WAIT:
                nst[r] = ("mf", v[1] + ws, v[2])
        st = nst
        if asm_mfma:
            for r in regs_of(inst.args.split(",")[0]):
                st[r] = ("mf", 0, inst.line)
    return st


def check_async(kname, items, findings):
    blocks, succ = build_blocks(items)
    n = len(blocks)
    state_in = [None] * n
    state_in[0] = {}
    work = [0]
    rounds = 0
    while work:
        i = work.pop()
        rounds += 1
        if rounds > 200000:
            findings.append(f"{kname}: dataflow did not converge")
            return
        st = dict(state_in[i])
        for inst in blocks[i][1]:
            st = transfer(st, inst, findings, kname, False)
        for j in succ[i]:
            old = state_in[j]
            if old is None:
                state_in[j] = dict(st)
                work.append(j)
            else:
                new, changed = dict(old), False
                for r, v in st.items():
                    if r not in new or new[r][1] > v[1]:
                        new[r] = v
                        changed = True
                if changed:
                    state_in[j] = new
                    work.append(j)
    for i in range(n):          # final pass with the converged entry states: report
        if state_in[i] is None:
            continue
        st = dict(state_in[i])
        for inst in blocks[i][1]:
            st = transfer(st, inst, findings, kname, True)


def check_asm_stores(kname, items, findings):
    insts = [it for it in items if not isinstance(it, tuple)]
    for idx, inst in enumerate(insts):
        if inst.asm and inst.op.startswith(("global_store", "buffer_store")):
            prev = insts[idx - 1] if idx else None
            if prev is None or not (prev.asm and prev.op == "s_nop"):
                findings.append(f"{kname}: line {inst.line}: inline-asm `{inst.text}` without an s_nop in front of it inside its asm statement (rule iii)")


def check_valu_to_mfma(kname, items, findings):
    """rule (iv): straight-line look-back (a label or branch in between resets it: the gap is then at least a taken branch)"""
    recent = []      # (registers written, Inst) of the last VALU instructions
    for it in items:
        if isinstance(it, tuple):
            recent = []
            continue
        if it.op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_barrier")):
            recent = []
            continue
        if it.asm and it.op.startswith("v_mfma"):
            ops = [a.strip() for a in it.args.split(",")]
            reads = set()
            for a in ops[1:4]:
                reads |= regs_of(a)
            for back, (wr, w) in enumerate(reversed(recent[-VALU_MFMA_GAP:])):
                hit = wr & reads
                if hit:
                    findings.append(f"{kname}: line {it.line}: inline-asm `{it.text}` reads v{min(hit)}..{max(hit)} {back} instruction(s) behind "
                                    f"`{w.text}` (line {w.line}) that writes it (rule iv)")
        if it.op.startswith("v_") and not it.op.startswith(("v_mfma", "v_cmp", "v_readlane", "v_readfirstlane")):
            first = it.args.split(",")[0] if it.args else ""
            recent.append((regs_of(first), it))
        else:
            recent.append((set(), it))
        recent = recent[-8:]


def check_text(text, require_guarded=True):
    kernels = split_kernels(text)
    findings, report = [], []
    for name, items in sorted(kernels.items()):
        before = len(findings)
        check_async(name, items, findings)
        check_asm_stores(name, items, findings)
        check_valu_to_mfma(name, items, findings)
        n_asm = sum(1 for it in items if not isinstance(it, tuple) and it.asm and (is_vmem(it.op) or is_ds(it.op)))
        report.append(f"{'ok  ' if len(findings) == before else 'FAIL'} {name}: {sum(1 for it in items if not isinstance(it, tuple))} instructions, "
                      f"{n_asm} inline-asm memory operations")
    if require_guarded and not kernels:
        findings.append("check_isa: no guarded kernel found")
    return findings, report


def main(path):
    findings, report = check_text(open(path, errors="replace").read())
    print("\n".join(report))
    if findings:
        seen = []
        for f in findings:
            if f not in seen:
                seen.append(f)
        print("\n".join(seen[:60]))
        print(f"check_isa: {len(seen)} finding(s)")
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
