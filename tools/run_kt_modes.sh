# rocprofv3 kernel stats of the train / demucs / demucs-train bench modes (outputs: gpurun_out/kt_<mode>_kernel_stats.csv + the line under the profiler)
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
for m in train demucs demucs-train; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$m -o p -- python3 bench.py --mode $m --steps 10 --warmup 3 --cpu-seconds 0 > $O/kt_${m}_line.json 2> $O/kt_$m.err || exit 1
  cp $(find $O/kt_$m -name "*kernel_stats.csv" | head -1) $O/kt_${m}_kernel_stats.csv
  rm -rf $O/kt_$m
done
