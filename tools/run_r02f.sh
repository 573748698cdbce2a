export TMPDIR=/tmp
O=gpurun_out/r02f; mkdir -p $O
for d in 0 32 64 1 0; do echo "== MFPA_CONV_DBG=$d"; MFPA_CONV_DBG=$d python tools/exp_conv.py --lib musicfpaugment_amd/libmfpa_exp.so 2>>$O/err.log | tee $O/conv_b_dbg$d.txt | grep -E "up4.0|d1.3|up2.0|up1.0|d4.3|sum"; done
