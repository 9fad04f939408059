#!/bin/bash
# Evidence for profiles/: run on the GPU box from the repo root:  bash tools/profile_round.sh r01c
# 1. rocprofv3 --kernel-trace --stats over bench.py (kernel durations; bench.py's own HIP-event figures must agree)
# 2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only, as the MI355X guide prescribes) -> <tag>_pmc_hbm.json
set -e
TAG=${1:-r01x}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- python3 $R/bench.py --steps 50 --warmup 5 --cpu-seconds 10 > $OUT/${TAG}_bench.json 2> $OUT/bench.err
cp $OUT/stats/*kernel_stats.csv $OUT/${TAG}_kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_$C -o pmc --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/pmc_$C.log 2>&1
done
python3 $R/tools/pmc_summarize.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE > $OUT/${TAG}_pmc_hbm.json
tail -c 600 $OUT/${TAG}_bench.json; echo; cat $OUT/${TAG}_pmc_hbm.json
