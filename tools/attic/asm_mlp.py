#!/usr/bin/env python3
"""Memory-level parallelism per wave, read off the compiler's assembly: for every kernel in a gfx950 .s file (hipcc
--save-temps), the sequence of global loads / stores / atomics, waits and barriers as one compact line per kernel, e.g.
`L L W0 B` = two loads in flight, then s_waitcnt vmcnt(0), then a barrier.  A run like `L W0 L W0 L W0` is a chain of round
trips to memory that the source did not ask for (typically loads behind `if (i < n)` in an unrolled loop).
Usage: python tools/asm_mlp.py file.s [kernel-substring]"""
import re
import sys

text = open(sys.argv[1]).read().splitlines()
want = sys.argv[2] if len(sys.argv) > 2 else ""
name, seq = None, []
for line in text:
    m = re.match(r"^(_Z\w+):", line)
    if m:
        name, seq = m.group(1), []
        continue
    if name is None:
        continue
    t = line.strip()
    if t.startswith(("global_load", "buffer_load", "flat_load")): seq.append("L")
    elif t.startswith(("global_store", "buffer_store", "flat_store")): seq.append("S")
    elif t.startswith(("global_atomic", "buffer_atomic", "flat_atomic")): seq.append("A")
    elif t.startswith("ds_"): seq.append("d")
    elif t.startswith("s_barrier"): seq.append("B")
    elif t.startswith("s_cbranch"): seq.append("^" if "BB" in t else "b")
    elif t.startswith("s_waitcnt"):
        m2 = re.search(r"vmcnt\((\d+)\)", t)
        if m2: seq.append("W" + m2.group(1))
    elif t.startswith("s_endpgm"):
        if want in name:
            out = " ".join(seq)
            out = re.sub(r"(?:d ?){2,}", lambda mm: "d*%d " % mm.group(0).count("d"), out)
            print(name[:60], "\n   ", out, "\n")
        name = None
