export TMPDIR=/tmp
O=gpurun_out/r02l; mkdir -p $O
for i in 1 2; do for v in prev "" noslp; do L=musicfpaugment_amd/libmfpa${v:+_$v}.so; echo "== ${v:-new} $i"; timeout -k 10 300 python tools/exp_conv.py --lib $L 2>>$O/err.log | tee $O/conv_${v:-new}_$i.txt | grep -E "inc.3|up4.0|d1.3|up1.0|sum"; done; done
