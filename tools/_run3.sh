cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/s4e; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python bench.py --steps 3 --warmup 2 --skip-cpu-baseline > $O/prof_bench.log 2>&1
T=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python tools/rocprof_pair_breakdown.py $T --pair -3 --phases --timeline 10 --encoders > $O/pair_breakdown.txt 2>&1
python - "$T" > $O/tail.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "corr1d_build" in r["Kernel_Name"]]
i = idx[-2]
t0 = int(rows[i - 40]["Start_Timestamp"])
for r in rows[i - 40:i + 40]:
    print("%9.1f %8.1f q%s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id"), r["Kernel_Name"][:90]))
PY
rm -rf $O/prof
cat $O/tail.txt
