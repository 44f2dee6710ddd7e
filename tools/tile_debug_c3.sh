# needs a library built with the knobs compiled in:  make -C cs348b-pbrt_amd/csrc clean && make -C cs348b-pbrt_amd/csrc EXTRA=-DPVOL_TIMING_KNOBS
# timing split of the FUSED tile pre-pass on the reduced C3 frame (PVOL_TILE_DEBUG knobs: results are wrong with any of them)
for d in ${PVOL_DBG_LIST:-0 32 64 128 160 256}; do
  PVOL_TILE_DEBUG=$d timeout -k 10 200 python tools/measure_configs.py C3 --no-li 2>/dev/null > gpurun_out/c3dbg_$d.jsonl
  python3 -c "
import json
for l in open('gpurun_out/c3dbg_$d.jsonl'):
    if l.startswith('{'):
        f=json.loads(l)['frame']; print('dbg $d frame_s', f['frame_s'])"
done
