# where the multi-wave tile pre-pass spends a pixel's time, at rank 0's share of an N-GPU frame (bench.py --emulate-ranks): needs a library
# with the knobs compiled into pvol_tile.hip (-DPVOL_TIMING_KNOBS) given as PVOL_LIB; 1 = no swaps, 2 = no draw count, 4 = no stream advance
for d in ${KNOBS:-0 1 2 4 7}; do
  PVOL_TILE_DEBUG=$d timeout -k 10 200 python bench.py --emulate-ranks ${WORLDS:-8} --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/tmw_$d.json 2> gpurun_out/tmw_$d.err
  python3 - <<PY
import json
d=json.load(open('gpurun_out/tmw_$d.json'))
for x in d['worlds']:
    if x['n_gpus'] == 1: continue
    r=x['ranks'][0]; print('knob $d N', x['n_gpus'], 'tile %.1f ms' % r['phases_ms']['tile_prepass'])
PY
done
