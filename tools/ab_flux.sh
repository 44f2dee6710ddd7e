# A/B of li_group_kernel builds on the C2 frame: libpvol.so against an alternative build given as PVOL_ALT (same ABI), each with the phase
# counters of the stats build (cycles per phase summed over waves)
set -e
B="python bench.py --no-cpu-baseline --steps 2 --warmup 1"
timeout -k 10 200 $B > gpurun_out/ab_new.json 2> gpurun_out/ab_new.err
timeout -k 10 200 $B --stats > gpurun_out/ab_new_stats.json 2> gpurun_out/ab_new_stats.err
if [ -n "$PVOL_ALT" ]; then
PVOL_LIB=$PWD/$PVOL_ALT timeout -k 10 200 $B > gpurun_out/ab_alt.json 2> gpurun_out/ab_alt.err
PVOL_LIB=$PWD/$PVOL_ALT timeout -k 10 200 $B --stats > gpurun_out/ab_alt_stats.json 2> gpurun_out/ab_alt_stats.err
fi
python3 - <<'PY'
import json, os
for n in ['ab_new','ab_new_stats','ab_alt','ab_alt_stats']:
    f='gpurun_out/%s.json'%n
    if not os.path.exists(f): continue
    d=json.load(open(f))
    print(n, 'value %.2f ms %.1f kernel %.1f'%(d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms']), {k:v for k,v in d.get('gpu_counters',{}).items() if k.startswith('cy')})
PY
