#!/bin/bash
# rocprofv3 counter passes over tools/prof_step.py; writes gpurun_out/pmc/<pass>/...  (run on the GPU box from the repo root)
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1 || true
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $line --kernel-trace -d $OUT/p$i -o p$i --output-format csv -- python3 $R/tools/prof_step.py ${RAYS:-1000} 6 > $OUT/p$i.log 2>&1
done <<'PASSES'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
PASSES
ls -R $OUT | head -40
