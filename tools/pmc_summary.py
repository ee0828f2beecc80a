#!/usr/bin/env python
"""Average PMC counters per kernel from the csv output of tools/pmc_collect.sh.

    python tools/pmc_summary.py gpurun_out/pmc > profiles/rNN_pmc.txt
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.match(r'(?:void )?([\w:]+(?:<[^(]{0,30}>)?)', name)
    return (m.group(1) if m else name)[:40]


def main(root):
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = short(row['Kernel_Name'])
                acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
    for k in sorted(acc, key=lambda k: -sum(acc[k].get('SQ_WAVE_CYCLES', [0]))):
        if not (k.startswith('edge') or k.startswith('node') or k.startswith('knn')):
            continue
        print(f'== {k}')
        for c in sorted(acc[k]):
            v = acc[k][c]
            print(f'   {c:28s} mean {sum(v) / len(v):16.1f}  (n={len(v)})')


if __name__ == '__main__':
    main(sys.argv[1])
