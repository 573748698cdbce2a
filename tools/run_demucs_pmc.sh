export TMPDIR=/tmp
O=gpurun_out/dmpmc; mkdir -p $O
CMD="python3 bench.py --mode demucs --clips 256 --steps 1 --warmup 1"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $O/s1 -o p --output-format csv -- $CMD > $O/s1.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU -d $O/s2 -o p --output-format csv -- $CMD > $O/s2.log 2>&1
python tools/summarize_sq.py $O/s1 gemm_,lstm_seq,glu_convT,c1_glu $O/p1.json > $O/sq1.txt 2>&1
python tools/summarize_sq.py $O/s2 gemm_,lstm_seq,glu_convT,c1_glu $O/p2.json > $O/sq2.txt 2>&1
rm -rf $O/s1 $O/s2
