# Run ON THE GPU BOX from the repo root:  bash tools/profile_round.sh [tag]
# bench line, rocprofv3 kernel-trace stats of the same command, HBM traffic passes, SQ counter passes -> gpurun_out/<tag>_*
set -e
R=$PWD
T=${1:-v10}
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prof -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/${T}_prof.log 2>&1
cd $R
python tools/collect_traffic.py > gpurun_out/${T}_traffic.log 2>&1
python tools/collect_sq.py > gpurun_out/${T}_sq.log 2>&1
