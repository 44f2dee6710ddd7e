
for w in 16; do
  PVOL_TILE_WAVES=$w timeout -k 10 400 python -m pytest tests/test_gpu_render.py -m gpu -x -q > gpurun_out/b1_render_w$w.log 2>&1; tail -2 gpurun_out/b1_render_w$w.log
done
for w in 1 4 8 16; do
  PVOL_TILE_WAVES=$w timeout -k 10 300 python bench.py --emulate-ranks 8 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/b1_emul_w$w.json 2> gpurun_out/b1_emul_w$w.err
  python3 - <<PY
import json
d=json.load(open('gpurun_out/b1_emul_w$w.json'))
for x in d['worlds']:
    r=x['ranks'][0]
    print('waves $w N',x['n_gpus'],'tile %.1f march %.1f step %.1f eff %.3f'%(r['phases_ms']['tile_prepass'],r['phases_ms']['march_gather'],x['projected_step_ms'],x['projected_efficiency']))
PY
done
