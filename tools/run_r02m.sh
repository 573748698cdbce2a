export TMPDIR=/tmp
O=gpurun_out/r02m; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_peaks.py tests/test_gpu_fullsize.py tests/test_gpu_hashes.py tests/test_gpu_stft.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout -k 10 300 python tools/time_small_kernels.py 256 2>>$O/err.log | tee $O/small_kernels.txt | head -8
for i in 1 2; do timeout -k 10 300 python bench.py --no-unet --steps 50 --warmup 5 > $O/bench_nounet_$i.json 2>>$O/err.log; python -c "import json;d=json.load(open('$O/bench_nounet_$i.json'));print(d['value'],d['ms_per_step'])"; done
