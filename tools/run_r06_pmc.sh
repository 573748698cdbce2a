# PMC passes of ONE 64-clip UNet eval forward (bf16x3), nothing else in the process (--no-extras); round 6: conv_up_kernel (the folded decoder levels) in the family
export TMPDIR=/tmp
O=gpurun_out/r06p; mkdir -p $O
CMD="python3 bench.py --clips 64 --steps 1 --warmup 0 --cpu-seconds 0 --no-configs --no-extras"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_f -o p --output-format csv -- $CMD > $O/pmc_f.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_w -o p --output-format csv -- $CMD > $O/pmc_w.log 2>&1
python tools/summarize_pmc.py $O/pmc_f $O/pmc_w $O/pmc_traffic_bf16x3.json 64 conv_mfma_kernel,convT_mfma_kernel,conv_wd16_kernel,conv_ws64_kernel,conv_up_kernel "the MFMA convolution launches of ONE {clips}-clip UNet eval forward (bf16x3)" '(, 1(, (false|true)(, [0-9]+)?(, (false|true))?)?>$)|(conv_wd16_kernel)|(conv_ws64_kernel)|(conv_up_kernel)' > $O/pmc_traffic.log 2>&1
rm -rf $O/pmc_f $O/pmc_w
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $O/pmc_s1 -o p --output-format csv -- $CMD > $O/pmc_s1.log 2>&1
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU -d $O/pmc_s2 -o p --output-format csv -- $CMD > $O/pmc_s2.log 2>&1
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum -d $O/pmc_s3 -o p --output-format csv -- $CMD > $O/pmc_s3.log 2>&1
python tools/summarize_sq.py $O/pmc_s1 conv_mfma_kernel,convT_mfma_kernel,conv_wd16_kernel,conv_ws64_kernel,conv_up_kernel $O/pmc_sq_pass1.json > $O/pmc_sq1.txt 2>&1
python tools/summarize_sq.py $O/pmc_s2 conv_mfma_kernel,convT_mfma_kernel,conv_wd16_kernel,conv_ws64_kernel,conv_up_kernel $O/pmc_sq_pass2.json > $O/pmc_sq2.txt 2>&1
python tools/summarize_sq.py $O/pmc_s3 conv_mfma_kernel,convT_mfma_kernel,conv_wd16_kernel,conv_ws64_kernel,conv_up_kernel $O/pmc_sq_pass3.json > $O/pmc_sq3.txt 2>&1
python tools/sq_table.py $O/pmc_sq_pass1.json $O/pmc_sq_pass2.json > $O/pmc_sq_table.md 2>&1
rm -rf $O/pmc_s1 $O/pmc_s2 $O/pmc_s3
tail -2 $O/pmc_traffic.log; cat $O/pmc_sq_table.md; head -c 1500 $O/pmc_sq3.txt
