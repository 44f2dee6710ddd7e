# tile pre-pass timing by waves per task and emulated world size (bench.py --emulate-ranks): picks the heuristic of tile_waves_per_task
for w in ${WAVES:-1 2 4 8}; do
  PVOL_TILE_WAVES=$w timeout -k 10 400 python bench.py --emulate-ranks ${WORLDS:-2,4,8} --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/c4_w$w.json 2> gpurun_out/c4_w$w.err
  python3 - <<PY
import json
d=json.load(open('gpurun_out/c4_w$w.json'))
for x in d['worlds']:
    r=x['ranks'][0]
    print('waves $w N',x['n_gpus'],'tile %.1f march %.1f step %.1f eff %.3f'%(r['phases_ms']['tile_prepass'],r['phases_ms']['march_gather'],x['projected_step_ms'],x['projected_efficiency']))
PY
done
