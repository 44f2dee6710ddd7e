# li_fixup_group_kernel radius policy on the reduced C3 frame: "widen aim" pairs in $SWEEP (PVOL_FXG_WIDEN / PVOL_FXG_AIM)
SWEEP=${SWEEP:-1.3:1.4 1.15:1.25 1.1:1.15 1.2:1.1 1.5:1.4}
for wa in $SWEEP; do
  w=${wa%%:*}; a=${wa##*:}
  PVOL_FXG_WIDEN=$w PVOL_FXG_AIM=$a timeout -k 10 200 python tools/measure_configs.py C3 --no-li --no-parity > gpurun_out/c3_sw_${w}_${a}.jsonl 2> gpurun_out/c3_sw.err || exit 1
  python3 - <<PY
import json
d=json.loads(open('gpurun_out/c3_sw_${w}_${a}.jsonl').read().strip().splitlines()[-1]); fr=d['frame']
print('widen $w aim $a frame %.2f s' % fr['frame_s'], fr['handed_over_shared_bucket'], fr['handed_over_exact_pass'], fr['bucket_stagings'], 'rgb %.6f' % fr['mean_rgb'])
PY
done
