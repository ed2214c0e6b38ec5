#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel stats + PMC passes of the bench command, then
# the default bench.  Everything lands in gpurun_out/$1/ ; tools/summarize_profiles.py turns it into
# the committed files under profiles/.
set -u
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $R/bench.py --steps 300 --warmup 100 --no-cpu-baseline"
# kernel stats over the SAME step counts as the default bench run, so that the averages are comparable
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline > $O/stats.json 2> $O/stats.err
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  name=$(echo $set | tr " " "_" | cut -c1-32)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$name -- $BENCH > $O/pmc_$name.json 2> $O/pmc_$name.err || echo "PMC pass failed: $set"
done
cd $R
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
cat $O/bench_default.json
