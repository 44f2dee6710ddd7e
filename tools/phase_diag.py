"""Prints li_group_kernel's phase shares from a `bench.py --stats` line made with a -DGRP_PHASE_DIAG=1|2 build (timing only)."""
import json, sys
d = json.load(open(sys.argv[1]))
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
c = d['gpu_counters']
tot = c['cy_total']
print('kernel ms', d['roofline']['kernel_avg_ms'])
for k in ['cy_search', 'cy_select', 'cy_flux', 'cy_fallback']:
    print(k, '%.3f' % (c[k] / tot))
names = {1: [('group_plan_skipped', 'pass1'), ('group_guess_failed', 'scan'), ('group_deferred_overflow', 'pass2'), ('group_deferred_too_few', 'rank')],
         2: [('group_guess_failed', 'step geometry + direct lighting'), ('group_plan_skipped', 'gather (guess, attempts, hand-over)'), ('group_deferred_overflow', 'recurrence')]}[mode]
for k, n in names:
    print(n, '%.3f' % (c[k] / tot))
