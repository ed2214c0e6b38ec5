#!/usr/bin/env python3
"""Per-kernel duration and inter-kernel gap statistics from a rocprofv3 kernel_trace.csv."""
import csv
import statistics
import sys


def main(path, pattern='step_kernel'):
    rows = [r for r in csv.DictReader(open(path)) if pattern in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    dur = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows]
    gap = [int(b['Start_Timestamp']) - int(a['End_Timestamp']) for a, b in zip(rows, rows[1:])]
    gap = [g for g in gap if g < 100_000]  # drop phase boundaries
    skip = len(dur) // 5
    d, g = dur[skip:], gap[skip:]
    print(f'{path}: {len(rows)} launches of *{pattern}*')
    print(f'  duration ns: mean {statistics.mean(d):.0f} median {statistics.median(d):.0f} min {min(d)} max {max(d)}')
    if g:
        print(f'  gap ns:      mean {statistics.mean(g):.0f} median {statistics.median(g):.0f} min {min(g)} max {max(g)}')
        print(f'  period ns:   {statistics.mean(d) + statistics.mean(g):.0f}')


if __name__ == '__main__':
    main(*sys.argv[1:])
