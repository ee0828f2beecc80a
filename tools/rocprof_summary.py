#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite output) as a short text table.

    python tools/rocprof_summary.py gpurun_out/prof_r1/c2_results.db > profiles/r01_c2_kernel_stats.txt
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    m = re.match(r'(?:void )?([\w:]+(?:<[^(]{0,40}>)?)', name)
    base = m.group(1) if m else name
    if 'distribution' in name:
        base += ' [normal]' if 'normal_kernel' in name else ' [uniform]'
    return base[:70]


def main(path):
    con = sqlite3.connect(path)
    rows = list(con.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
    print(f'# rocprofv3 --kernel-trace --stats  ({path})')
    print(f'{"kernel":70s} {"calls":>7s} {"total_us":>12s} {"avg_us":>10s} {"pct":>7s}')
    for name, calls, total, avg, pct in rows:
        print(f'{short(name):70s} {calls:7d} {total / 1e3 if total > 1e7 else total:12.1f} {avg:10.3f} {pct:7.2f}')


if __name__ == '__main__':
    main(sys.argv[1])
