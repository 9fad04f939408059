#!/bin/bash
# same-box A/B of the K5 loop (Tracker + bundle-adjustment iterations) and of a Tracker iteration: libnsk_prev.so against libnsk.so
for lib in prev new prev new; do
  if [ $lib = prev ]; then export NSK_LIB=$PWD/nice-slam-cpp_amd/csrc/libnsk_prev.so; else unset NSK_LIB; fi
  python bench.py --k5-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', 'ms/frame', round(d['ms_per_frame'],3), d['kernels_us_per_frame'], 'loss', d['final_ba_loss'])"
  python tools/track_times.py 200 2>/dev/null | tail -3
done
