set -x
export PYTHONPATH=.
timeout 1200 python -m pytest tests/test_gemm_gpu.py -q -m gpu -k "interleaved_4bit" 2>&1 | tail -4
MX='bp.brgemm_mx4i8(api, 64, 2 ** 17);;bp.brgemm_mx4i8(api, 64, 2 ** 17, DT.F32)'
TAG=mx4i8_cvt WL="$MX" timeout 300 python tools/_one.py 2>&1 | tail -2
TAG=mx4i8_old LIBXSMM_HIP_MX4I8_PIPE=0 WL="$MX" timeout 300 python tools/_one.py 2>&1 | tail -2
