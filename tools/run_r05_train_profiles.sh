# The train-step part of tools/run_r05_profiles.sh (re-run after the late round-5 train-step changes): config 4's own command, its variants, the
# kernel stats of the same command and the ordered launch list of one pre-mixed step.  Outputs under gpurun_out/r05p (then tools/copy_r05_profiles.sh).
export TMPDIR=/tmp
O=gpurun_out/r05p; mkdir -p $O
timeout -k 10 300 python bench.py --mode train --precision bf16 --wgrad bf16 --augment --steps 20 --warmup 5 > $O/train_step_bf16_bench_line.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --mode train --precision bf16x3 --wgrad bf16 --augment --steps 20 --warmup 5 > $O/train_step_bf16x3_bench_line.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --mode train --precision bf16 --wgrad bf16 --augment --no-z16 --steps 20 --warmup 5 > $O/train_step_bf16_f32act_bench_line.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --mode train --precision bf16 --wgrad bf16 --steps 20 --warmup 5 > $O/train_step_bf16_premixed_bench_line.json 2>> $O/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktt -o p -- python3 bench.py --mode train --precision bf16 --wgrad bf16 --augment --steps 10 --warmup 3 --cpu-seconds 0 > $O/train_step_bf16_bench_under_rocprof.json 2>> $O/bench.err
cp $(find $O/ktt -name "*kernel_stats.csv" | head -1) $O/train_step_bf16_kernel_stats.csv; rm -rf $O/ktt
cd /tmp; rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/ktr -o tr -- python3 $GRAFT_REPO_ROOT/bench.py --mode train --precision bf16 --wgrad bf16 --steps 4 --warmup 2 --no-configs --no-extras --cpu-seconds 0 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT; python tools/train_trace.py $O/ktr/tr_kernel_trace.csv > $O/train_step_launches.txt 2>&1; rm -rf $O/ktr
timeout -k 10 700 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2>> $O/bench.err
