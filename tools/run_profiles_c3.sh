#!/bin/bash
# PMC passes of the C3 hand-over kernels (li_fixup_group_kernel, li_fixup_kernel, li_group_kernel's REPLAY form) on the reduced C3
# frame: pinkfloyd 1920x1080 at 8 spp, 4 M photons requested, nused 500 -- Li() alone through the native driver tools/pvol_prof.
#   bash tools/run_profiles_c3.sh r03_c3        then  python tools/summarize_pmc_kernels.py r03_c3
set -u
TAG=${1:-r03_c3}
SPP=${2:-8}
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
python3 tools/make_prof_inputs.py /tmp/prof_c3 --scene pinkfloyd --photons 4000000 --shoot-tasks 256 --xres 1920 --yres 1080 --spp $SPP > $OUT/${TAG}_prof_inputs.log 2>&1 || exit 1
./tools/pvol_prof /tmp/prof_c3 1 > $OUT/${TAG}_prof_plain.json 2> $OUT/${TAG}_prof_plain.err || exit 1
cat $OUT/${TAG}_prof_plain.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- ./tools/pvol_prof /tmp/prof_c3 1 > $OUT/${TAG}_trace.log 2>&1
echo "trace exit=$?"
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD" \
            "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    name=${pass%% *}
    timeout -k 10 300 rocprofv3 --pmc $pass --output-format csv -d $OUT/${TAG}_pmc_$name -- ./tools/pvol_prof /tmp/prof_c3 1 > $OUT/${TAG}_pmc_$name.log 2>&1
    echo "$name exit=$?"
done
