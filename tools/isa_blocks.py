#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel's ISA (hipcc -S output cut to the kernel).

    tools/isa_blocks.py kernel.s [min_instructions]

Columns: V = vector ALU, S = scalar ALU, L = LDS, M = vector memory, B = branches, X = vector instructions that
write scalar registers (compares, readlane, readfirstlane, div_scale, carry-out adds), P = S + B + X: what goes
through the CU's one scalar unit (tools/microbench_issue.hip: about 4.3 cycles each per SIMD, against 2.7 for V).
"""
import re
import sys

SGPR_WRITERS = ("v_cmp", "v_readlane", "v_readfirstlane", "v_div_scale", "v_add_co", "v_sub_co", "v_addc_co", "v_subb_co")
NOT_SALU = ("s_waitcnt", "s_nop", "s_endpgm", "s_barrier", "s_sleep", "s_setprio", "s_code_end")


def main():
    lines = open(sys.argv[1]).read().split('\n')
    least = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    blocks = []
    cur = dict(name='entry', V=0, S=0, L=0, M=0, B=0, X=0, all=0, note='', line=0)
    for n, l in enumerate(lines):
        m = re.match(r'^(\.LBB\d+_\d+):(.*)', l)
        if m:
            blocks.append(cur)
            cur = dict(name=m.group(1), V=0, S=0, L=0, M=0, B=0, X=0, all=0, note=m.group(2).strip(), line=n + 1)
            continue
        t = l.strip()
        if not t or t.startswith(';') or t.startswith('.'):
            continue
        op = t.split()[0]
        if op.startswith('v_'):
            cur['V'] += 1
            if op.startswith(SGPR_WRITERS):
                cur['X'] += 1
        elif op.startswith('s_cbranch') or op in ('s_branch', 's_setpc_b64'):
            cur['B'] += 1
        elif op.startswith('s_') and not op.startswith(NOT_SALU):
            cur['S'] += 1
        elif op.startswith('ds_'):
            cur['L'] += 1
        elif op.startswith(('global_', 'buffer_', 'scratch_', 'flat_')):
            cur['M'] += 1
        cur['all'] += 1
    blocks.append(cur)
    tot = {k: sum(b[k] for b in blocks) for k in ('V', 'S', 'L', 'M', 'B', 'X', 'all')}
    print('static totals VALU %(V)d SALU %(S)d LDS %(L)d VMEM %(M)d branch %(B)d sgpr-writing-VALU %(X)d all %(all)d' % tot)
    for b in blocks:
        if b['all'] >= least:
            print('%-10s line %5d  V%3d S%3d L%2d M%2d B%2d X%2d  P%3d | %s' % (b['name'], b['line'], b['V'], b['S'], b['L'], b['M'], b['B'], b['X'],
                                                                     b['S'] + b['B'] + b['X'], b['note'][:60]))


if __name__ == "__main__":
    main()
