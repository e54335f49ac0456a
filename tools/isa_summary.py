#!/usr/bin/env python3
"""Compressed view of one kernel's instruction stream in a hipcc .s file (runs of the same class collapsed):
    python tools/isa_summary.py file.s <substring of the mangled kernel name> [max chars]"""
import sys


def cls(line):
    l = line.strip()
    if l.startswith('v_mfma'): return 'MFMA'
    if l.startswith('ds_read'): return 'DSR'
    if l.startswith('ds_write'): return 'DSW'
    if l.startswith('global_load_lds') or (l.startswith('buffer_load') and l.endswith('lds')): return 'GLDS'
    if l.startswith('global_load') or l.startswith('buffer_load'): return 'GLD'
    if l.startswith('global_store') or l.startswith('buffer_store'): return 'GST'
    if l.startswith('scratch_'): return 'SCRATCH'
    if l.startswith('s_barrier'): return 'BAR'
    if l.startswith('s_waitcnt'): return 'W[' + l.split(None, 1)[1] + ']'
    if l.startswith('s_setprio'): return 'PRIO'
    if l.startswith('s_cbranch') or l.startswith('s_branch'): return 'BR:' + l.split()[-1]
    if l.startswith('.LBB'): return '\n' + l
    if l.startswith('v_exp') or l.startswith('v_rcp') or l.startswith('v_log') or l.startswith('v_rsq'): return 'TRANS'
    if l.startswith('v_permlane') or '_dpp' in l: return 'XLANE'
    if l.startswith('s_'): return 's'
    if l.startswith('v_'): return 'v'
    return None


def main():
    text = open(sys.argv[1]).read()
    key = sys.argv[2]
    limit = int(sys.argv[3]) if len(sys.argv) > 3 else 12000
    starts, pos = [], 0
    for line in text.split('\n'):
        if line.startswith('_Z') and ':' in line and key in line.split(':')[0]:
            starts.append(pos)
        pos += len(line) + 1
    if not starts:
        raise SystemExit('kernel not found')
    a = starts[0]
    b = text.index('.Lfunc_end', a)
    out, prev, cnt = [], None, 0
    for line in text[a:b].split('\n'):
        c = cls(line)
        if c is None:
            continue
        if c == prev:
            cnt += 1
        else:
            if prev:
                out.append(f'{prev}x{cnt}' if cnt > 1 else prev)
            prev, cnt = c, 1
    out.append(f'{prev}x{cnt}' if cnt > 1 else prev)
    print(text[a:text.find(':', a)])
    print(' '.join(out)[:limit])


if __name__ == '__main__':
    main()
