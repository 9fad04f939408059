#!/bin/bash
# Evidence for profiles/: run on the GPU box from the repo root:  bash tools/profile_round.sh r04
# 1. rocprofv3 --kernel-trace --stats over bench.py's default command (kernel durations; bench.py's own HIP-event figures must agree)
# 2. separate --pmc passes, kernel-trace only, as the MI355X guide prescribes: FETCH_SIZE, WRITE_SIZE -> <tag>_pmc_hbm.json;
#    SQ passes (MFMA instructions / busy cycles, wave cycles, LDS bank conflicts, instruction-cache misses) -> <tag>_pmc_sq.json
set -e
TAG=${1:-r04x}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# --no-extras: the extra workloads launch the same kernels on other shapes, and the per-kernel averages must be the headline workload's
timeout -k 10 420 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- python3 $R/bench.py --steps 200 --warmup 20 --cpu-seconds 8 --no-extras > $OUT/${TAG}_bench.json 2> $OUT/bench.err
# the default command, unprofiled (headline + extras + both CPU baselines)
timeout -k 10 420 python3 $R/bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/bench_default.err
cp $OUT/stats/*kernel_stats.csv $OUT/${TAG}_kernel_stats.csv
BARGS="--steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-extras"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_$C -o pmc --output-format csv -- python3 $R/bench.py $BARGS > $OUT/pmc_$C.log 2>&1
done
python3 $R/tools/pmc_summarize.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE > $OUT/${TAG}_pmc_hbm.json
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $line -d $OUT/sq$i -o sq --output-format csv -- python3 $R/bench.py $BARGS > $OUT/sq$i.log 2>&1
done <<'PASSES'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES
SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_SALU GRBM_GUI_ACTIVE
PASSES
python3 $R/tools/pmc_summarize.py $OUT/sq1 $OUT/sq2 $OUT/sq3 > $OUT/${TAG}_pmc_sq.json
tail -c 600 $OUT/${TAG}_bench.json; echo; cat $OUT/${TAG}_pmc_hbm.json; head -c 3000 $OUT/${TAG}_pmc_sq.json
