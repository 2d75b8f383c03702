"""Advisory ISA scan: `s_waitcnt vmcnt(0)` with a global / buffer load shortly before AND after it inside one kernel -- the
signature of loads that the compiler serialised (round 4: a run-time layout switch and `d < nd ? load : const` around every load of the
ShiftGELU table pass cost one HBM latency per load on the row-major path; a plain `dst[i] = src[i]` copy loop global -> LDS does the same).
Usage: hipcc --offload-arch=gfx950 -O3 ... --cuda-device-only -S -o x.s csrc/x.hip; python3 scripts/scan_serial_loads.py x.s [...]
Prints kernels with at least two such waits; a hit is a place to read, not a verdict (a dependent load needs its wait)."""
import re
import sys

for path in sys.argv[1:]:
    s = open(path).read()
    for name in re.findall(r"\.type\s+(_Z\w+),@function", s):
        i = s.find("\n" + name + ":")
        if i < 0:
            continue
        j = s.find(".Lfunc_end", i)
        ins = [ln.strip() for ln in s[i:j].split("\n") if ln.strip() and not ln.strip().startswith((";", "."))]
        hits = 0
        for k, ln in enumerate(ins):
            if ln.startswith("s_waitcnt vmcnt(0)"):
                after = any(x.startswith(("global_load", "buffer_load")) for x in ins[k + 1:k + 11])
                before = any(x.startswith(("global_load", "buffer_load")) for x in ins[max(0, k - 12):k])
                hits += after and before
        if hits >= 2:
            print(f"{path}: {hits:3d}  {name}")
