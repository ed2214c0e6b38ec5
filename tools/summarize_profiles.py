#!/usr/bin/env python3
"""Turn gpurun_out/<tag>/ (written by tools/profile_step.sh) into the committed profiles/ files."""
import csv
import glob
import json
import shutil
import statistics
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent


def main(tag, label):
    src = REPO / 'gpurun_out' / tag
    dst = REPO / 'profiles'
    dst.mkdir(exist_ok=True)
    stats = glob.glob(str(src / 'stats' / '*' / '*kernel_stats.csv'))
    if stats:
        shutil.copy(stats[0], dst / f'{label}_kernel_stats.csv')
    counters = {}
    for path in glob.glob(str(src / 'pmc_*' / '*' / '*counter_collection.csv')):
        rows = [r for r in csv.DictReader(open(path)) if 'step_kernel' in r['Kernel_Name']]
        by_name = {}
        for r in rows:
            by_name.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
        for name, vals in by_name.items():
            vals = vals[len(vals) // 4:]  # skip the warm-up launches
            counters[name] = {'mean': statistics.mean(vals), 'launches': len(vals)}
    summary = {'kernel': 'step_kernel<false>', 'boards_per_launch': 1 << 20, 'counters': counters}
    if 'FETCH_SIZE' in counters and 'WRITE_SIZE' in counters:
        # rocprofv3 reports KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide coalesced
        # reads -> double it (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact for 16-B stores.
        fetch = counters['FETCH_SIZE']['mean'] * 1024 * 2
        write = counters['WRITE_SIZE']['mean'] * 1024
        summary['hbm_bytes_per_launch'] = fetch + write
        summary['hbm_read_bytes'] = fetch
        summary['hbm_write_bytes'] = write
        summary['algorithmic_bytes_per_launch'] = 38 * (1 << 20)
        traffic_path = dst / 'traffic.json'
        table = json.loads(traffic_path.read_text()) if traffic_path.exists() else {}
        table[str(1 << 20)] = {'hbm_bytes_per_launch': fetch + write, 'source': f'profiles/{label}_pmc_summary.json',
                               'formula': '(2*FETCH_SIZE + WRITE_SIZE) KiB, separate --pmc passes'}
        traffic_path.write_text(json.dumps(table, indent=1) + '\n')
    if 'SQ_INSTS_VALU' in counters and 'SQ_WAVES' in counters:
        summary['valu_instructions_per_board'] = counters['SQ_INSTS_VALU']['mean'] * 64 / (1 << 20) / 64 * 64 / 64 * 1.0
        summary['valu_instructions_per_board'] = counters['SQ_INSTS_VALU']['mean'] / counters['SQ_WAVES']['mean'] / 4
    (dst / f'{label}_pmc_summary.json').write_text(json.dumps(summary, indent=1) + '\n')
    bench = src / 'bench_default.json'
    if bench.exists() and bench.stat().st_size:
        shutil.copy(bench, dst / f'{label}_bench.json')
    print(json.dumps(summary, indent=1)[:1500])


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
