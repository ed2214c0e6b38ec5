#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel stats + PMC passes of the bench command for BOTH legs of
# bench.py (cache-resident headline, HBM-cold), then the default bench.  Everything lands in gpurun_out/$1/ ;
# tools/summarize_profiles.py turns it into the committed files under profiles/ (incl. profiles/step_pmc.json,
# which bench.py reads for roofline.traffic and the VALU instruction count).
# Counter passes are separate runs with --kernel-trace only (never combined with other trace domains).
set -u
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
RES="python3 $R/bench.py --no-cpu-baseline --no-cold --no-other-configs"
COLD="python3 $R/bench.py --no-cpu-baseline --only-cold"
# kernel stats over the SAME step counts as the default bench run, so that the averages are comparable
# the default run takes the DIRECT route (the library's own queues); --no-direct = the stream route (two chains; --single-chain: one)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_direct -- $RES > $O/stats_direct.json 2> $O/stats_direct.err
echo "stats direct rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_resident -- $RES --no-direct > $O/stats_resident.json 2> $O/stats_resident.err
echo "stats resident rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_onechain -- $RES --no-direct --single-chain > $O/stats_onechain.json 2> $O/stats_onechain.err
echo "stats resident (one chain) rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cold -- $COLD > $O/stats_cold.json 2> $O/stats_cold.err
echo "stats cold rc=$?"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  name=$(echo $set | tr " " "_" | cut -c1-32)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_direct_$name -- $RES --steps 300 --warmup 100 --max-blocks 3 > $O/pmc_direct_$name.json 2> $O/pmc_direct_$name.err || echo "PMC pass failed: direct $set"
  echo "pmc direct $name done"
done
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $set | tr " " "_" | cut -c1-32)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_resident_$name -- $RES --no-direct --steps 300 --warmup 100 --max-blocks 3 > $O/pmc_resident_$name.json 2> $O/pmc_resident_$name.err || echo "PMC pass failed: resident $set"
  echo "pmc resident $name done"
done
# the one-chain shape (whole-batch launches): what the driver's --steps 20 run times (K < 32 never splits)
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $set | tr " " "_" | cut -c1-32)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_onechain_$name -- $RES --no-direct --single-chain --steps 300 --warmup 100 --max-blocks 3 > $O/pmc_onechain_$name.json 2> $O/pmc_onechain_$name.err || echo "PMC pass failed: one chain $set"
  echo "pmc onechain $name done"
done
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"; do
  name=$(echo $set | tr " " "_" | cut -c1-32)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_cold_$name -- $COLD > $O/pmc_cold_$name.json 2> $O/pmc_cold_$name.err || echo "PMC pass failed: cold $set"
  echo "pmc cold $name done"
done
cd $R
LABEL=${2:-$(basename $TAG)}
# summaries first (then the raw per-dispatch CSVs, tens of MB, are dropped), and the fresh counter table goes where bench.py
# reads it, so that the bench lines below carry the traffic / VALU counts of THIS kernel build
python3 tools/summarize_profiles.py $TAG $LABEL $O/summary > $O/summary.log 2>&1 && test -s $O/summary/step_pmc.json && rm -rf $O/pmc_*/ $O/stats_*/
echo "summary rc=$?"
cp $O/summary/step_pmc.json $R/profiles/step_pmc.json
python3 bench.py > $O/summary/${LABEL}_bench_default.json 2> $O/bench_default.err
echo "bench default rc=$?"
python3 bench.py --steps 20 --warmup 5 > $O/summary/${LABEL}_bench_driver_shape.json 2> $O/bench_driver_shape.err
echo "bench driver-shape rc=$?"
