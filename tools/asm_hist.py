#!/usr/bin/env python3
"""Static instruction mix of a kernel in hipcc's assembly (-S --cuda-device-only): tools/asm_hist.py file.s kernel-substring"""
import re, sys, collections
txt = open(sys.argv[1]).read().splitlines()
name = sys.argv[2]
start = next(i for i, l in enumerate(txt) if re.match(r'^_ZN2a3\w*' + name + r'\w*:', l))
end = next(i for i in range(start, len(txt)) if txt[i].startswith(".Lfunc_end"))
c = collections.Counter()
for line in txt[start + 1:end + 1]:
    line = line.strip()
    if not line or line[0] in ';.' or line.endswith(':'): continue
    c[line.split()[0]] += 1
tot = sum(c.values()); v = sum(n for o, n in c.items() if o.startswith('v_'))
print(f"== {name}: {tot} static instructions, {v} VALU, {sum(n for o, n in c.items() if o.startswith('s_'))} SALU/branch, {sum(n for o, n in c.items() if o.startswith('ds_'))} LDS, {sum(n for o, n in c.items() if o.startswith(('global_', 'buffer_', 'flat_', 'scratch_')))} VMEM")
print('  ' + ', '.join(f"{o} {n}" for o, n in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 45)))
