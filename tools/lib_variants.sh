# Run ON THE GPU BOX: the bench line with each of the given library builds (ORBHIP_LIB).  bash tools/lib_variants.sh tag lib1.so lib2.so ...
T=$1; shift
mkdir -p gpurun_out/$T
for l in "$@"; do
  n=$(basename $l .so)
  ORBHIP_LIB=$PWD/$l python bench.py --no-cpu-baseline --no-host-fed $BENCH_OPTS > gpurun_out/$T/$n.json 2>gpurun_out/$T/$n.err || { echo "$l failed"; tail -3 gpurun_out/$T/$n.err; exit 1; }
  python - <<P
import json; d=json.load(open("gpurun_out/$T/$n.json")); print("$n", "->", d["value"], d["roofline"]["stage_ms_per_batch"])
P
done
