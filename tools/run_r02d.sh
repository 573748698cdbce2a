export TMPDIR=/tmp
O=gpurun_out/r02d; mkdir -p $O
python -m pytest tests/test_gpu_unet.py tests/test_gpu_fullsize.py -x -q > $O/tests.log 2>&1; tail -2 $O/tests.log
for v in pipe0 "" big64 bn8; do L=musicfpaugment_amd/libmfpa${v:+_$v}.so; echo "== ${v:-pipe1}"; python tools/exp_conv.py --lib $L 2>>$O/err.log | tee $O/conv_${v:-pipe1}.txt; done
for i in 1 2; do for v in pipe0 ""; do L=musicfpaugment_amd/libmfpa${v:+_$v}.so; python bench.py --cpu-seconds 0 --no-configs --steps 5 --lib $L > $O/bench_${v:-pipe1}_$i.json 2>>$O/err.log; python -c "import json;d=json.load(open('$O/bench_${v:-pipe1}_$i.json'));print('${v:-pipe1}',d['value'],d['roofline']['kernel_ms_per_step'])"; done; done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $O/pmc1 -o p --output-format csv -- python3 tools/exp_conv.py --reps 1 > $O/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $O/pmc2 -o p --output-format csv -- python3 tools/exp_conv.py --reps 1 > $O/pmc2.log 2>&1
python tools/summarize_sq.py $O/pmc1 conv_mfma_kernel $O/pmc1.json > $O/pmc1_summary.txt 2>&1
python tools/summarize_sq.py $O/pmc2 conv_mfma_kernel $O/pmc2.json > $O/pmc2_summary.txt 2>&1
rm -rf $O/pmc1 $O/pmc2
tail -3 $O/pmc1.log
