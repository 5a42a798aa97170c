timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_beside_another_stream.py -m gpu -q -x -k "bn or batch or norm or stats or channel_sums or beside" 2>&1 | tail -3
for v in 0 1; do echo "M2D_BN_VEC_ROWS=$v"; M2D_BN_VEC_ROWS=$v python tools/bn_bwd_time.py 2>&1 | tail -25; done
