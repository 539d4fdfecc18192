""" Summarise rocprofv3 CSV output directories: per-kernel mean duration (kernel trace) and mean counter values. """
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    return name.split('(')[0][:70]


def main(dirs):
    for d in dirs:
        for path in sorted(glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)):
            dur = defaultdict(list)
            for row in csv.DictReader(open(path)):
                dur[short(row['Kernel_Name'])].append(int(row['End_Timestamp']) - int(row['Start_Timestamp']))
            print(f'== {path}')
            for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
                print(f'   {k:70s} calls {len(v):5d}  mean {sum(v) / len(v) / 1e3:10.2f} us  total {sum(v) / 1e6:9.3f} ms')
        for path in sorted(glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)):
            vals = defaultdict(lambda: defaultdict(list))
            for row in csv.DictReader(open(path)):
                vals[short(row['Kernel_Name'])][row['Counter_Name']].append(float(row['Counter_Value']))
            print(f'== {path}')
            for k, ctrs in vals.items():
                if 'tile' not in k:
                    continue
                print(f'   {k}')
                for c, v in sorted(ctrs.items()):
                    print(f'      {c:34s} mean {sum(v) / len(v):16.1f}  (n={len(v)})')


if __name__ == '__main__':
    main(sys.argv[1:])
