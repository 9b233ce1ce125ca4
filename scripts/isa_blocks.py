#!/usr/bin/env python3
"""Instruction counts per basic block of one kernel's loops, from `hipcc -S --cuda-device-only` output.
usage: isa_blocks.py kern.s <substring of the mangled kernel name> [loop index]"""
import re
import sys

asm, name = sys.argv[1], sys.argv[2]
which = int(sys.argv[3]) if len(sys.argv) > 3 else None
lines = open(asm).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^[_A-Za-z].*:", l) and name in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
blocks, cur = [], ["entry", 0, []]
for l in body:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m or l.startswith("; %bb."):
        blocks.append(cur)
        cur = [m.group(1) if m else l.split()[1].rstrip(":"), 0, []]
        if "Loop Header" in l:
            cur[2].append("LOOP")
        continue
    if re.match(r"^\t[a-z]", l):
        cur[1] += 1
        op = l.split()[0]
        if op.startswith(("s_cbranch", "s_branch", "global_", "ds_", "flat_", "s_swappc", "s_waitcnt")):
            cur[2].append(op + (" " + l.split()[-1] if op.startswith("s_c") or op.startswith("s_b") else ""))
blocks.append(cur)
loops = [i for i, b in enumerate(blocks) if "LOOP" in b[2]]
print("kernel instructions:", sum(b[1] for b in blocks), "loops at blocks", loops)
rng = range(len(blocks)) if which is None else range(loops[which], min(len(blocks), loops[which] + 60))
for i in rng:
    b = blocks[i]
    print("%4d %-14s %3d  %s" % (i, b[0], b[1], " ".join(x for x in b[2] if x != "LOOP")[:150]))
