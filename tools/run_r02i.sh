export TMPDIR=/tmp
O=gpurun_out/r02i; mkdir -p $O
for d in 0 1 16 17; do
MFPA_CONV_DBG=$d rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU -d $O/pmc_d$d -o p --output-format csv -- python3 tools/exp_conv.py --reps 1 --lib musicfpaugment_amd/libmfpa_exp.so > $O/pmc_d$d.log 2>&1
python tools/summarize_sq.py $O/pmc_d$d conv_mfma_kernel $O/pmc_d$d.json > $O/pmc_d${d}_summary.txt 2>&1
rm -rf $O/pmc_d$d
done
tail -2 $O/pmc_d0.log
