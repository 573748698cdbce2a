export TMPDIR=/tmp
O=gpurun_out/r02r; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_unet.py tests/test_gpu_fullsize.py tests/test_gpu_train.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
for i in 1 2 3; do for v in prev ""; do L=musicfpaugment_amd/libmfpa${v:+_$v}.so; timeout -k 10 300 python bench.py --cpu-seconds 0 --no-configs --steps 5 --lib $L > $O/b_${v:-new}_$i.json 2>>$O/err.log; python -c "import json;d=json.load(open('$O/b_${v:-new}_$i.json'));print('${v:-new}',d['value'],d['roofline']['kernel_ms_per_step'],d['other_precision']['value'])"; done; done
