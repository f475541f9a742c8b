# Run ON THE GPU BOX from the repo root:  bash tools/prof_cmd.sh <tag> <python script and args...>
# rocprofv3 --kernel-trace --stats of one command; prints the kernel summary and keeps it as gpurun_out/<tag>_kernel_stats.csv
T=$1; shift
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${T}_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prof -- python3 "$@" > $R/gpurun_out/${T}_prof.log 2>&1
cd $R
f=$(ls gpurun_out/${T}_prof/*/*kernel_stats.csv | head -1)
cp $f gpurun_out/${T}_kernel_stats.csv
rm -rf gpurun_out/${T}_prof
python3 - <<PY
import csv
for r in list(csv.DictReader(open("gpurun_out/${T}_kernel_stats.csv")))[:16]:
    print("%-70s calls %6s avg %10.1f us  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
