# Run ON THE GPU BOX: per-section instruction counts of k_fast by difference of early-exit diagnostic builds
# (build/liborbhip_stop{0..3}.so = -DFAST_STOP=n, plus the product build).  Output: gpurun_out/fast_sections.txt
set -e
R=$PWD
mkdir -p gpurun_out
: > gpurun_out/fast_sections.txt
for n in 0 1 2 3 full; do
  if [ "$n" = full ]; then unset ORBHIP_LIB; else export ORBHIP_LIB=$R/build/liborbhip_stop$n.so; fi
  timeout -k 10 120 python tools/collect_sq.py "SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAIT_INST_ANY,SQ_WAIT_ANY,SQ_ACTIVE_INST_ANY" --no-match > gpurun_out/fast_sections_$n.log 2>&1 || true
  echo "== stop after section $n" >> gpurun_out/fast_sections.txt
  grep -A9 "^k_fast" gpurun_out/fast_sections_$n.log >> gpurun_out/fast_sections.txt || true
done
cat gpurun_out/fast_sections.txt
