export TMPDIR=/tmp
O=gpurun_out/r02q; mkdir -p $O
for i in 1 2; do for v in "" conda; do L=musicfpaugment_amd/libmfpa${v:+_$v}.so; echo "== ${v:-new} $i"; timeout -k 10 300 python tools/exp_conv.py --lib $L 2>>$O/err.log | grep -E "d1.3|up3.0|up1.0|d4.3|sum"; done; done
for i in 1 2; do for v in "" conda; do L=musicfpaugment_amd/libmfpa${v:+_$v}.so; timeout -k 10 300 python bench.py --cpu-seconds 0 --no-configs --steps 5 --lib $L > $O/b_${v:-new}_$i.json 2>>$O/err.log; python -c "import json;d=json.load(open('$O/b_${v:-new}_$i.json'));print('${v:-new}',d['value'],d['other_precision']['value'])"; done; done
