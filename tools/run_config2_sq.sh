# SQ counters of the STFT -> pick chain's three kernels at 256 clips (two rocprofv3 --pmc passes): gpurun_out/config2_sq/{sq1,sq2}_256.json
export TMPDIR=/tmp
O=gpurun_out/config2_sq; mkdir -p $O
B=256
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $O/p1_$B -o p --output-format csv -- python3 bench.py --no-unet --clips $B --steps 2 --warmup 1 --cpu-seconds 0 --no-configs --no-extras > $O/p1_$B.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES -d $O/p2_$B -o p --output-format csv -- python3 bench.py --no-unet --clips $B --steps 2 --warmup 1 --cpu-seconds 0 --no-configs --no-extras > $O/p2_$B.log 2>&1
python tools/summarize_sq.py $O/p1_$B stft_kernel,prep_sum_kernel,prune_kernel $O/sq1_$B.json > $O/sq1_$B.txt 2>&1
python tools/summarize_sq.py $O/p2_$B stft_kernel,prep_sum_kernel,prune_kernel $O/sq2_$B.json > $O/sq2_$B.txt 2>&1
rm -rf $O/p1_$B $O/p2_$B
cat $O/sq1_$B.txt $O/sq2_$B.txt | cut -c1-700
