#!/usr/bin/env python
"""Print the kernel timeline of the last sampling step of a `rocprofv3 --kernel-trace --output-format csv` run.

    python tools/step_timeline.py gpurun_out/ovl [first_row [rows]]
"""
import csv
import glob
import os
import re
import sys


def main(d, first=0, count=200):
    f = max(glob.glob(os.path.join(d, '**', '*_kernel_trace.csv'), recursive=True), key=os.path.getmtime)
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    idx = [i for i, r in enumerate(rows) if 'posterior' in r['Kernel_Name']]
    a, b = idx[-2], idx[-1]
    t0 = int(rows[a]['End_Timestamp'])
    print(f'# {f}: step of {(int(rows[b]["End_Timestamp"]) - t0) / 1e3:.1f} us, {b - a} kernels')
    print(f'{"start_us":>9s} {"end_us":>9s} {"dur_us":>8s} queue kernel')
    for r in rows[a + 1 + first:min(b + 1, a + 1 + first + count)]:
        nm = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])[:56]
        st, en = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
        print(f'{st / 1e3:9.1f} {en / 1e3:9.1f} {(en - st) / 1e3:8.1f} q{r["Queue_Id"]:>3s}  {nm}  grid {r["Grid_Size_X"]}')


if __name__ == '__main__':
    main(sys.argv[1], *(int(v) for v in sys.argv[2:4]))
