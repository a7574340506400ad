cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_lookup8
run() { n=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_lookup8/$n -o $n -- python $R/tools/prof_conv.py lookup8 3 > $R/gpurun_out/pmc_lookup8/$n.log 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE
cd $R; python - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob("gpurun_out/pmc_lookup8/*/*counter_collection.csv")):
    d=collections.defaultdict(list); t=[]
    for r in csv.DictReader(open(f)):
        if "lookup" in r["Kernel_Name"]:
            d[r["Counter_Name"]].append(float(r["Counter_Value"])); t.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
    for k,v in d.items(): print(k, v, "kernel us", t)
PY
