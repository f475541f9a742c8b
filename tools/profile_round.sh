set -e
R=$PWD
python bench.py > gpurun_out/v5_bench.json 2> gpurun_out/v5_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/v5_prof -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/v5_prof.log 2>&1
cd $R
python tools/collect_traffic.py > gpurun_out/v5_traffic.log 2>&1
