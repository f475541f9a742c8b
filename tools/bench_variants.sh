# Run ON THE GPU BOX: the bench line under a list of option sets.  bash tools/bench_variants.sh tag "opts1" "opts2" ...
T=$1; shift
mkdir -p gpurun_out/$T
i=0
for o in "$@"; do
  python bench.py --no-cpu-baseline --no-host-fed $o > gpurun_out/$T/v$i.json 2>gpurun_out/$T/v$i.err || { echo "variant $i failed"; tail -3 gpurun_out/$T/v$i.err; exit 1; }
  python - <<P
import json; d=json.load(open("gpurun_out/$T/v$i.json")); print("$o", "->", d["value"], d.get("verified_frames"), d["roofline"]["stage_ms_per_batch"])
P
  i=$((i+1))
done
