# SQ counters of the Audfprint picker kernels (stft / prepare / prune) at 256 and 8192 clips: two rocprofv3 --pmc passes per size
export TMPDIR=/tmp
O=gpurun_out/prune_sq; mkdir -p $O
for B in 256 8192; do
  python tools/time_small_kernels.py $B 2>&1 | head -5 > $O/time_$B.txt
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $O/p1_$B -o p --output-format csv -- python3 bench.py --no-unet --clips $B --steps 2 --warmup 1 --cpu-seconds 0 --no-configs > $O/p1_$B.log 2>&1 &&
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES -d $O/p2_$B -o p --output-format csv -- python3 bench.py --no-unet --clips $B --steps 2 --warmup 1 --cpu-seconds 0 --no-configs > $O/p2_$B.log 2>&1
  python tools/summarize_sq.py $O/p1_$B stft_kernel,prepare_kernel,prune_kernel $O/sq1_$B.json > $O/sq1_$B.txt 2>&1
  python tools/summarize_sq.py $O/p2_$B stft_kernel,prepare_kernel,prune_kernel $O/sq2_$B.json > $O/sq2_$B.txt 2>&1
  rm -rf $O/p1_$B $O/p2_$B
done
cat $O/time_256.txt $O/time_8192.txt; cat $O/sq1_256.txt $O/sq1_8192.txt | cut -c1-600
