#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel's ISA (hipcc -S output cut to the kernel)."""
import re
import sys
lines = open(sys.argv[1]).read().split('\n')
blocks = []
cur = ['entry', 0, 0, 0, 0, 0, '', 0]
for n, l in enumerate(lines):
    m = re.match(r'^(\.LBB\d+_\d+):(.*)', l)
    if m:
        blocks.append(cur)
        cur = [m.group(1), 0, 0, 0, 0, 0, m.group(2).strip(), n + 1]
        continue
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'):
        continue
    op = t.split()[0]
    if op.startswith('v_'):
        cur[1] += 1
    elif op.startswith('s_'):
        cur[2] += 1
    elif op.startswith('ds_'):
        cur[3] += 1
    elif op.startswith('global_') or op.startswith('buffer_'):
        cur[4] += 1
    cur[5] += 1
blocks.append(cur)
tot = [sum(b[i] for b in blocks) for i in range(1, 6)]
print('static totals VALU %d SALU %d LDS %d VMEM %d all %d' % tuple(tot))
for b in blocks:
    if b[5] >= int(sys.argv[2]) if len(sys.argv) > 2 else 6:
        print('%-10s line %5d  V%3d S%3d L%2d M%2d | %s' % (b[0], b[7], b[1], b[2], b[3], b[4], b[6][:70]))
