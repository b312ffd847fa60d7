mkdir -p gpurun_out/exp
for v in base 16 1 2 3 4 8; do
  if [ $v = base ]; then unset PMK_LIB; else export PMK_LIB=$PWD/patchmixturekriging_amd/csrc/libpmk_hip_v$v.so; fi
  PMK_BENCH_NOCHECK=1 timeout -k 10 200 python bench.py --no-cpu --steps 5 --warmup 2 --nq 65536 > gpurun_out/exp/v$v.json 2> gpurun_out/exp/v$v.err || echo "variant $v failed"
done
