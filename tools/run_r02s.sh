export TMPDIR=/tmp
O=gpurun_out/r02s; mkdir -p $O
timeout -k 10 300 python tools/exp_wgrad_bf16.py 2>>$O/err.log | tee $O/wgrad_bf16.txt
timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_fullsize.py -x -q -k "train or wgrad" > $O/tests.log 2>&1; tail -3 $O/tests.log
for i in 1 2; do for v in prev ""; do L=musicfpaugment_amd/libmfpa${v:+_$v}.so; [ "$v" = prev ] && continue; timeout -k 10 300 python bench.py --mode train --steps 5 --warmup 2 > $O/train_${v:-new}_$i.json 2>>$O/err.log; python -c "import json;d=json.load(open('$O/train_${v:-new}_$i.json'));print('${v:-new}',d['value'],d['ms_per_step'])"; done; done
