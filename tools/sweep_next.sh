#!/bin/bash
# First GPU call of the next round: what round 2 changed after its GPU time was spent, measured before anything else.
#   gpurun --timeout 900 -- 'bash tools/sweep_next.sh'
# 1. the tests of the kernels whose build changed (signed-A int8 BCSC at bn = 16: two waves per SIMD, no scratch), 2. that variant's time next to the
# unsigned-A one, 3. the generated packed CSR kernel of config #3 at one / two / four elements per lane (61 / 108 / 166 registers: 8 / 4 / 3 waves).
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_sparse_gpu.py -m gpu -q -x -p no:cacheprovider > gpurun_out/sweep_tests.log 2>&1; echo "sparse tests rc=$?"; tail -3 gpurun_out/sweep_tests.log
timeout 300 python tools/bench_paths.py --only bcsc_i8u8 --steps 50 > gpurun_out/sweep_bcsc_i8u8.jsonl 2> gpurun_out/sweep_bcsc_i8u8.err; echo "bcsc rc=$?"; cat gpurun_out/sweep_bcsc_i8u8.jsonl
for v in 1 2 4; do
  LIBXSMM_HIP_JIT_VEC=$v timeout 300 python tools/bench_paths.py --only csr,fsspmdm --steps 50 > gpurun_out/sweep_csr_vec$v.jsonl 2> gpurun_out/sweep_csr_vec$v.err
  echo "csr vec=$v rc=$?"; cat gpurun_out/sweep_csr_vec$v.jsonl
done
# 4. vector-valued reductions inside generated equation kernels (generated and CPU-checked in round 2, off by default): the equation tests with the switch on
LIBXSMM_HIP_MEQN_VECRED=1 timeout 600 python -m pytest tests/test_meqn.py -m gpu -q -p no:cacheprovider -k "reduce_bcast" > gpurun_out/sweep_vecred.log 2>&1; echo "vecred rc=$?"; tail -5 gpurun_out/sweep_vecred.log
