# rocprofv3 kernel stats of the UNet train step (gpurun_out/kt_train_kernel_stats.csv)
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_train -o p -- python3 bench.py --mode train --steps 10 --warmup 3 --cpu-seconds 0 > $O/kt_train_line.json 2> $O/kt_train.err || exit 1
cp $(find $O/kt_train -name "*kernel_stats.csv" | head -1) $O/kt_train_kernel_stats.csv
rm -rf $O/kt_train
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/kt_train_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total ms/step %.2f"%(tot/1e6/13))
for r in rows[:16]:
    print("%7.2f ms/step %5d calls %9.1f us  %s"%(float(r["TotalDurationNs"])/1e6/13, int(r["Calls"]), float(r["AverageNs"])/1e3, r["Name"][:100]))
PY
