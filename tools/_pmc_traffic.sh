# HBM traffic of the two roofline kernels: separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_traffic
run() { n=$1; k=$2; shift 2; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_traffic/$n -o $n -- python $R/tools/prof_conv.py $k 3 > $R/gpurun_out/pmc_traffic/$n.log 2>&1; }
run conv_fetch zr_gate FETCH_SIZE
run conv_write zr_gate WRITE_SIZE
run look_fetch lookup FETCH_SIZE
run look_write lookup WRITE_SIZE
cd $R; python - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob("gpurun_out/pmc_traffic/*/*counter_collection.csv")):
    d=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "conv2d_f16s" in r["Kernel_Name"] or "lookup" in r["Kernel_Name"]:
            d[(r["Kernel_Name"][:50],r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k,v in d.items(): print(f.split("/")[2], k, v)
PY
