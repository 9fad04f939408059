#!/bin/bash
# round-4 helper: bit-level A/B of libnsk_prev.so against libnsk.so, then the same-box timing A/B
R=$PWD
NSK_LIB=$R/nice-slam-cpp_amd/csrc/libnsk_prev.so python tools/ab_outputs.py 2>&1 | grep -v amdgpu.ids > gpurun_out/ab_prev.txt
python tools/ab_outputs.py 2>&1 | grep -v amdgpu.ids > gpurun_out/ab_new.txt
if diff gpurun_out/ab_prev.txt gpurun_out/ab_new.txt > gpurun_out/ab_diff.txt; then echo "BIT-IDENTICAL"; else echo "DIFFERENT"; cat gpurun_out/ab_diff.txt; fi
cat gpurun_out/ab_new.txt
bash tools/ab_bench.sh gpurun_out/r4_ab2 "$@"
