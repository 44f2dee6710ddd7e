#!/bin/bash
# Round profiles on the MI355X box (run through gpurun from the repo root):
#   bash tools/run_profiles.sh r01g
# 1. kernel trace of the default bench.py run (full C2)        -> gpurun_out/<tag>_trace, <tag>_bench_traced.json
# 2. PMC passes of the native driver tools/pvol_prof (rocprofv3's counter mode is unreliable under python+torch
#    on this pool), separate passes, no trace domains combined with --pmc -> gpurun_out/<tag>_pmc_<first counter>
# then `python tools/summarize_pmc.py <tag>` (on either side) condenses them into profiles/.
set -u
TAG=${1:-r02}
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
python3 tools/make_prof_inputs.py /tmp/prof_in --spp 256 --xres 640 --yres 360 > $OUT/${TAG}_prof_inputs.log 2>&1 || exit 1
./tools/pvol_prof /tmp/prof_in 2 > $OUT/${TAG}_prof_plain.json 2> $OUT/${TAG}_prof_plain.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > $OUT/${TAG}_bench_traced.json 2> $OUT/${TAG}_bench_traced.err
echo "trace_exit=$?"
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD" \
            "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"; do
    name=${pass%% *}
    timeout -k 10 300 rocprofv3 --pmc $pass --output-format csv -d $OUT/${TAG}_pmc_$name -- ./tools/pvol_prof /tmp/prof_in 1 > $OUT/${TAG}_pmc_$name.log 2>&1
    echo "$name exit=$?"
done
