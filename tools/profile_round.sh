# Run ON THE GPU BOX from the repo root:  bash tools/profile_round.sh [tag]
# bench lines (both workloads), rocprofv3 kernel-trace stats of the same bench command, HBM traffic passes, SQ counter passes,
# vector-issue calibration -> gpurun_out/<tag>_*  (copy what is to be kept into profiles/).
set -e
R=$PWD
T=${1:-r03}
mkdir -p gpurun_out
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
echo "bench euroc done"
python bench.py --config tumvi > gpurun_out/${T}_bench_tumvi.json 2> gpurun_out/${T}_bench_tumvi.err
echo "bench tumvi done"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${T}_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prof -- python3 $R/bench.py --no-cpu-baseline --no-host-fed > $R/gpurun_out/${T}_prof.log 2>&1
# one pipeline, synchronised nowhere but at the step ends: the HIP-event stage times of the bench line and rocprofv3's dispatch
# durations must agree here (with four streams the events also see the wait for CUs held by the other streams)
rm -rf $R/gpurun_out/${T}_s1_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_s1_prof -- python3 $R/bench.py --streams 1 --steps 4 --warmup 1 --no-cpu-baseline --no-host-fed > $R/gpurun_out/${T}_s1_bench.json 2> $R/gpurun_out/${T}_s1_prof.log
cd $R
cp $(ls gpurun_out/${T}_s1_prof/*/*kernel_stats.csv | head -1) gpurun_out/${T}_s1_kernel_stats.csv
rm -rf gpurun_out/${T}_s1_prof
python bench.py --no-match --no-cpu-baseline --no-host-fed > gpurun_out/${T}_bench_extract_only.json 2> /dev/null
echo "kernel trace done"
cp $(ls gpurun_out/${T}_prof/*/*kernel_stats.csv | head -1) gpurun_out/${T}_kernel_stats.csv
rm -rf gpurun_out/${T}_prof      # the per-dispatch trace is large; the summary is what is kept
python tools/collect_traffic.py > gpurun_out/${T}_traffic.log 2>&1
cp gpurun_out/traffic.json gpurun_out/${T}_traffic.json
echo "traffic done"
python tools/collect_sq.py > gpurun_out/${T}_sq.log 2>&1
cp gpurun_out/sq_counters.json gpurun_out/${T}_sq_counters.json
echo "sq done"
python tools/collect_valu_calib.py --out gpurun_out/${T}_valu_calib.json > gpurun_out/${T}_valu_calib.log 2>&1
echo "calib done"
