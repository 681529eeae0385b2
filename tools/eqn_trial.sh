#!/bin/bash
# one-off probe: the equation sample drivers that exercise gather / matmul nodes (output to gpurun_out/eqn_trial.log)
D=oracle/_ref/drivers; mkdir -p gpurun_out; L=gpurun_out/eqn_trial.log; : > $L
run() { echo "=== $*" >> $L; LIBXSMM_VERBOSE=1 timeout 20 "$@" >> $L 2>&1; echo "--- rc=$?" >> $L; }
run $D/equation_gather_reduce 37 21 40 0 0 2
run $D/equation_gather_reduce 64 32 64 1 1 2
run $D/equation_gather_dot 1024 48 64 16 2
run $D/equation_gather_bcstmul_add 1024 48 64 16 2
run $D/equation_matmul 5 32 16 32 1 32 16 32 1 32 64 32 4 64 16 64 4 32 16 32 1 0 0 2
run $D/equation_matmul 5 64 16 64 1 64 16 64 1 64 64 64 3 64 16 64 3 64 16 64 1 1 1 2
grep -n "rc=\|Check-norm\|refused\|error\|Error" $L | head -80
