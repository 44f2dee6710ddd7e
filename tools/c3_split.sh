# C3 reduced frame (8 spp) on the product library, then on a -DPVOL_FXG_TIME build (PVOL_ALT) for li_fixup_group_kernel's cycle split
set -e
timeout -k 10 300 python tools/measure_configs.py C3 --no-li --no-parity --raw-stats > gpurun_out/c3_red.jsonl 2> gpurun_out/c3_red.err
if [ -n "$PVOL_ALT" ]; then PVOL_LIB=$PWD/$PVOL_ALT timeout -k 10 300 python tools/measure_configs.py C3 --no-li --no-parity --raw-stats > gpurun_out/c3_red_time.jsonl 2> gpurun_out/c3_red_time.err; fi
python3 - <<'PY'
import json, os
for n in ['c3_red', 'c3_red_time']:
    f = 'gpurun_out/%s.jsonl' % n
    if not os.path.exists(f): continue
    d = json.loads(open(f).read().strip().splitlines()[-1])
    fr = d['frame']
    print(n, 'shoot %.2f frame %.2f s  %.3f Msamples/s  march %.2f s' % (d['shoot_s'], fr['frame_s'], fr['Msamples_per_s_whole_pipeline'], fr['march_kernels_s']),
          {k: fr[k] for k in ('handed_over_shared_bucket', 'handed_over_exact_pass', 'bucket_stagings')}, fr['exact_pass_because'])
    if n.endswith('time'):
        w = fr['raw_stats']
        tot = w['cy_total']
        print('  fxg split: probe %.3f stage %.3f select %.3f flux %.3f slow %.3f  runs %d skipped %d' % (
            w['n_tested'] / tot, w['cy_search'] / tot, w['cy_select'] / tot, w['cy_flux'] / tot, w['n_kept'] / tot, w['n_lookups_lt10'], w['n_shadow_unoccluded']))
PY
