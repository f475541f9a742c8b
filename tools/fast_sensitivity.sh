# Run ON THE GPU BOX: how much does k_fast's time move per extra instruction of each class?  Diagnostic builds
# build/liborbhip_pad{1,2,3}.so (-DFAST_PAD=n) add 16 instructions of one class to every pass-1 trip (~4.4 trips per wavefront,
# 4 wavefronts per cell: ~70 extra wave-instructions per wavefront, +17 % on k_fast's 412 vector instructions per wavefront).
# 1 = v_add_u32 (issues every 2 cycles in isolation), 2 = v_pk_max_i16 (every 4), 3 = s_add_u32 (scalar unit).
set -e
for n in base 1 2 3; do
  if [ "$n" = base ]; then unset ORBHIP_LIB; else export ORBHIP_LIB=$PWD/build/liborbhip_pad$n.so; fi
  python bench.py --no-cpu-baseline --no-host-fed --no-match --steps 6 --warmup 1 > gpurun_out/fast_pad_$n.json 2> gpurun_out/fast_pad_$n.err
  python -c "
import json,sys; d=json.loads(open('gpurun_out/fast_pad_$n.json').read().strip().splitlines()[-1]); print('$n', 'k_fast alone ms', d['roofline']['isolated']['stage_ms_per_batch']['fast'], 'value', d['value'])"
done
