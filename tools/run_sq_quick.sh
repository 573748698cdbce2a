# two SQ counter passes of one 64-clip UNet eval forward (bf16x3): usage: tools/run_sq_quick.sh <tag> [extra bench args]
export TMPDIR=/tmp
TAG=$1; shift
O=gpurun_out/sq_$TAG; mkdir -p $O
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $O/pmc_s1 -o p --output-format csv -- python3 bench.py --clips 64 --steps 1 --warmup 0 --cpu-seconds 0 --no-configs "$@" > $O/pmc_s1.log 2>&1 &&
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU -d $O/pmc_s2 -o p --output-format csv -- python3 bench.py --clips 64 --steps 1 --warmup 0 --cpu-seconds 0 --no-configs "$@" > $O/pmc_s2.log 2>&1 &&
python tools/summarize_sq.py $O/pmc_s1 conv_mfma_kernel,convT_mfma_kernel,conv_wd16_kernel $O/pmc_sq1.json > $O/pmc_sq1.txt 2>&1 &&
python tools/summarize_sq.py $O/pmc_s2 conv_mfma_kernel,convT_mfma_kernel,conv_wd16_kernel $O/pmc_sq2.json > $O/pmc_sq2.txt 2>&1
rm -rf $O/pmc_s1 $O/pmc_s2
ls $O
