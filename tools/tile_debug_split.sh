# needs a library built with the knobs compiled in:  make -C cs348b-pbrt_amd/csrc clean && make -C cs348b-pbrt_amd/csrc EXTRA=-DPVOL_TIMING_KNOBS
for d in 0 2 4 6 8; do
  PVOL_TILE_DEBUG=$d python bench.py --no-cpu-baseline --steps 1 --warmup 1 2>/dev/null > gpurun_out/tdbg_$d.json
  python3 -c "
import json,sys
d=json.load(open('gpurun_out/tdbg_$d.json')); print('dbg $d', d['ms_per_step'], d['roofline']['kernel_avg_ms'])"
done
