for shape in 0 824 842; do LIBXSMM_HIP_BM_SHAPE=$shape timeout 300 python tools/bb_sweep.py --sizes 4096x4096x4096,4096x4096x16384 2>&1 | grep -v amdgpu.ids; done
LIBXSMM_HIP_BM_SHAPE=824 LIBXSMM_HIP_BB_ABL=3 timeout 300 python tools/bb_sweep.py --sizes 4096x4096x16384 2>&1 | grep -v amdgpu.ids
