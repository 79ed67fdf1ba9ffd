#!/usr/bin/env python3
"""tools/loop_regs.py <file.s> <kernel symbol substring>: VGPRs of the kernel's innermost loop -- which are only read there
(loop-invariant addresses and constants: candidates for scalar operands or immediates), which are written, and the total --
to see what a register budget has to pay for.  Reads hipcc -S --cuda-device-only output."""
import re, sys
src, pat = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and pat in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith("; Occupancy"))
body = lines[start:end]
heads = [i for i, l in enumerate(body) if "Inner Loop Header" in l]
h = heads[-1] if len(sys.argv) < 4 else heads[int(sys.argv[3])]
label = body[h].split(":")[0]
t = max(i for i in range(h + 1, len(body)) if re.search(r"s_c?branch\w*\s+" + re.escape(label) + r"\b", body[i]))
def regs(tok):
    out = []
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(3): out.append(int(m.group(3)))
        else: out += list(range(int(m.group(1)), int(m.group(2)) + 1))
    return out
rd, wr = set(), set()
for l in body[h + 1:t]:
    l = l.split(";")[0].strip()
    if not l or l.endswith(":") or l.startswith("."): continue
    op, _, rest = l.partition(" ")
    ops = [o.strip() for o in rest.split(",")]
    store = op.startswith(("buffer_store", "global_store", "ds_write", "scratch_store", "s_")) or "lds" in l.split()[-1:]
    dst = [] if store or not ops else regs(ops[0])
    if op.startswith("buffer_load") and " lds" in " " + rest: dst = []
    srcs = [r for o in (ops if not dst else ops[1:]) for r in regs(o)]
    if op.startswith("v_mfma"): srcs += regs(ops[0]) if ops[0] == ops[-1] else []
    rd.update(srcs); wr.update(dst)
inv = sorted(rd - wr)
print(f"loop {label}: {t - h} lines; VGPRs touched {len(rd | wr)}, written {len(wr)}, read-only (invariant) {len(inv)}: {inv}")
for l in body[h + 1:t]:
    if any(re.search(rf"\bv{r}\b", l) for r in inv) and not l.strip().startswith("v_mfma"): print("   ", l.strip()[:110])
