export TMPDIR=/tmp
O=gpurun_out/r02g; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_unet.py tests/test_gpu_fullsize.py tests/test_gpu_train.py tests/test_gpu_graph.py -x -q > $O/tests.log 2>&1; tail -4 $O/tests.log
timeout -k 10 300 python tools/exp_conv.py 2>>$O/err.log | tee $O/conv_new.txt
for i in 1 2; do for v in pipe0 ""; do L=musicfpaugment_amd/libmfpa${v:+_$v}.so; timeout -k 10 300 python bench.py --cpu-seconds 0 --no-configs --steps 5 --lib $L > $O/bench_${v:-new}_$i.json 2>>$O/err.log; python -c "import json;d=json.load(open('$O/bench_${v:-new}_$i.json'));print('${v:-new}',d['value'],d['roofline']['kernel_ms_per_step'])"; done; done
