# per-launch durations of the GEMM-family kernels of ONE Demucs forward (256 clips): gpurun_out/demucs_gemm_calls.txt
export TMPDIR=/tmp
O=gpurun_out/ktd; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O -o p -- python3 bench.py --mode demucs --steps 1 --warmup 1 --cpu-seconds 0 > $O/line.json 2> $O/err.log || exit 1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/ktd/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
keep = [r for r in rows if any(k in r["Kernel_Name"] for k in ("gemm_", "lstm_", "c1_glu", "glu_convT"))]
half = keep[len(keep) // 2:]          # the timed step (the warm-up step comes first)
out = []
for r in half:
    nm = r["Kernel_Name"].split("(")[0].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    out.append("%-44s grid %6s %5s %5s wg %4s  %9.1f us" % (nm[:44], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
open("gpurun_out/demucs_gemm_calls.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
rm -rf $O
