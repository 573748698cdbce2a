export TMPDIR=/tmp
O=gpurun_out/r02o; mkdir -p $O
for i in 1 2 3; do for v in prev ""; do L=musicfpaugment_amd/libmfpa${v:+_$v}.so; timeout -k 10 300 python bench.py --cpu-seconds 0 --no-configs --steps 5 --lib $L > $O/bench_${v:-new}_$i.json 2>>$O/err.log; python -c "import json;d=json.load(open('$O/bench_${v:-new}_$i.json'));print('${v:-new}',d['value'],d['roofline']['kernel_ms_per_step'])"; done; done
