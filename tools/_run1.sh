export PYTHONPATH=.
for h in 0 1 2; do
TAG=stream$h LIBXSMM_HIP_STREAMING=$h WL='bp.bcsc(api, dtype="u8i8");;bp.bcsc(api, dtype="i8u8")' timeout 200 python tools/time_one.py 2>&1 | grep '^{' | cut -c1-230
done
