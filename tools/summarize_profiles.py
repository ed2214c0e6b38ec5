#!/usr/bin/env python3
"""Turn gpurun_out/<tag>/ (written by tools/profile_step.sh) into the committed profiles/ files:
    profiles/<label>_{resident,resident_one_chain,cold}_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the bench command
                                                        (default plan = two half-batch chains; --single-chain; the cold leg)
    profiles/<label>_pmc_summary.json                   counter means per 2^20-board record: resident step launches, the shared cold launch
                                                        (step_many_kernel, 30 records per launch) and the same cold records as separate launches
    (tools/profile_step.sh adds <label>_bench_default.json / _bench_driver_shape.json next to them)
    profiles/step_pmc.json                              what bench.py reads (keyed by leg:boards, stamped with the kernel-source hash)
usage: tools/summarize_profiles.py <gpurun_out tag> <label> [<destination dir, default profiles/>]
tools/profile_step.sh runs this ON the GPU box with destination gpurun_out/<tag>/summary and then deletes the raw
per-dispatch CSVs (tens of MB), so only the summaries travel back; copy them into profiles/ to commit them."""
import csv
import glob
import json
import shutil
import statistics
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import bench  # noqa: E402  (kernel_source_sha only; nothing touches a GPU)

KERNEL = 'step_kernel<false, false, 4>'  # lean outputs, host actions, the large tiling (what 2^20-board launches run)
SHARED = 'step_many_kernel<false, false, 4>'  # the same block code over the records of an env2048_step_many call
BOARDS = 1 << 20
# leg -> (rocprof run it comes from, kernel): the --only-cold run holds BOTH cold forms.  Every figure is scaled to ONE 2^20-board
# step (record): a launch's share is boards-per-launch / 2^20, read off its grid size (512-thread blocks of 2048 boards: 4 boards
# per work-item) -- 1/30 of a shared cold launch, and TWO launches per step where env2048_step_sequence runs two half-batch chains.
DIRECT = 'env2048_step_resident_large'  # the resident variant the direct route dispatches on the library's own queues
LEGS = {'direct': ('direct', DIRECT), 'resident': ('resident', KERNEL), 'resident_one_chain': ('onechain', KERNEL), 'cold': ('cold', SHARED),
        'cold_separate': ('cold', KERNEL)}  # resident_one_chain: bench.py --single-chain (whole-batch launches: what a K < 32 run times)


def leg_counters(src: Path, leg: str):
    run, kernel = LEGS[leg]
    counters, per_launch_boards = {}, []
    for path in glob.glob(str(src / f'pmc_{run}_*' / '*' / '*counter_collection.csv')):
        rows = [r for r in csv.DictReader(open(path)) if kernel in r['Kernel_Name']]
        by_name = {}
        for r in rows:
            boards = float(r['Grid_Size']) * 4.0
            per_launch_boards.append(boards)
            by_name.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']) * BOARDS / boards)
        for name, vals in by_name.items():
            vals = vals[len(vals) // 4:]  # skip the warm-up launches
            counters[name] = {'mean': statistics.mean(vals), 'launches': len(vals)}  # per 2^20-board step
    boards = statistics.median(per_launch_boards) if per_launch_boards else float(BOARDS)
    return counters, boards


def kernel_avg_us(stats_csv: Path, leg: str, boards_per_launch: float):
    _, kernel = LEGS[leg]
    for r in csv.DictReader(open(stats_csv)):
        if kernel in r['Name']:
            return {'calls': int(r['Calls']), 'avg_us': float(r['AverageNs']) / 1e3, 'min_us': float(r['MinNs']) / 1e3,
                    'max_us': float(r['MaxNs']) / 1e3, 'boards_per_launch': boards_per_launch,
                    'avg_us_per_2pow20_boards': float(r['AverageNs']) / 1e3 * BOARDS / boards_per_launch}
    return None


def main(tag, label, dest=None):
    src = REPO / 'gpurun_out' / tag
    dst = Path(dest) if dest else REPO / 'profiles'
    dst.mkdir(parents=True, exist_ok=True)
    summary = {'kernel': KERNEL, 'boards_per_launch': BOARDS, 'kernel_source_sha': bench.kernel_source_sha(),
               'algorithmic_bytes_per_launch': 38 * BOARDS, 'legs': {}}
    table_path = dst / 'step_pmc.json'
    table = {'kernel_source_sha': bench.kernel_source_sha(), 'entries': {}, 'kernel_stats': {},
             'formula': 'fabric bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB from separate rocprofv3 --pmc passes (gfx950: FETCH_SIZE counts '
                        'wide coalesced reads at half their bytes; MI355X_MICROARCH.md, HBM section); counts Infinity-Cache hits'}
    for leg in LEGS:
        stats = glob.glob(str(src / f'stats_{LEGS[leg][0]}' / '*' / '*kernel_stats.csv'))
        counters, boards_per_launch = leg_counters(src, leg)
        entry = {'kernel': LEGS[leg][1], 'boards_per_launch': boards_per_launch, 'launches_per_2pow20_board_step': BOARDS / boards_per_launch,
                 'per': 'counters are scaled to ONE 2^20-board step (all the launches, or the share of a shared launch, that step it)'}
        if stats:
            if leg != 'cold_separate':
                shutil.copy(stats[0], dst / f'{label}_{leg}_kernel_stats.csv')
            entry['rocprof_kernel_stats'] = kernel_avg_us(Path(stats[0]), leg, boards_per_launch)
            if entry['rocprof_kernel_stats'] is not None:
                ks = dict(entry['rocprof_kernel_stats'], source=f'profiles/{label}_{leg}_kernel_stats.csv' if leg != 'cold_separate'
                          else f'profiles/{label}_cold_kernel_stats.csv')
                try:  # the step period bench.py itself measured in that PROFILED run (the profiler slows the enqueuing threads down)
                    line = (src / f'stats_{LEGS[leg][0]}.json').read_text().strip().splitlines()[-1]
                    ks['bench_event_us_per_step_in_the_profiled_run'] = json.loads(line)['timing']['event_us_per_step']['median']
                    # the profiler adds host time to every launch; when the run's own step period is far above the kernel time the
                    # profiled run was LAUNCH-BOUND: the GPU idled between kernels (clocks sag, kernels of two chains stop overlapping)
                    # and the average below is not the duration the kernel has in the timed, unprofiled run
                    per_step = ks['avg_us'] * (1 if leg != 'resident' else 1)
                    ks['launch_bound_in_the_profiled_run'] = ks['bench_event_us_per_step_in_the_profiled_run'] > 1.15 * per_step
                except Exception:
                    pass
                table['kernel_stats'][leg] = ks
        entry['counters'] = counters
        if 'FETCH_SIZE' in counters and 'WRITE_SIZE' in counters:
            fetch = counters['FETCH_SIZE']['mean'] * 1024 * 2
            write = counters['WRITE_SIZE']['mean'] * 1024
            entry.update(fabric_read_bytes=fetch, fabric_write_bytes=write, fabric_bytes_per_launch=fetch + write,
                         traffic_over_algorithmic=(fetch + write) / (38 * BOARDS))
            row = {'fabric_bytes_per_launch': fetch + write, 'source': f'profiles/{label}_pmc_summary.json', 'boards_per_kernel_launch': boards_per_launch,
                   'per': '2^20-board step'}
            if 'SQ_INSTS_VALU' in counters:
                row['valu_insts_per_launch'] = counters['SQ_INSTS_VALU']['mean']
            table['entries'][f'{leg}:{BOARDS}'] = row
        if 'SQ_INSTS_VALU' in counters and 'SQ_WAVES' in counters:
            entry['valu_instructions_per_board'] = counters['SQ_INSTS_VALU']['mean'] * 64 / BOARDS
        summary['legs'][leg] = entry
    (dst / f'{label}_pmc_summary.json').write_text(json.dumps(summary, indent=1) + '\n')
    if table['entries']:
        table_path.write_text(json.dumps(table, indent=1) + '\n')
    brief = {leg: {k: v for k, v in e.items() if k != 'counters'} for leg, e in summary['legs'].items()}
    print(json.dumps(brief, indent=1))


if __name__ == '__main__':
    main(*sys.argv[1:4])
