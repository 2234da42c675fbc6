cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_stress_gpu.py -m gpu -x -q 2>&1 | tail -4
for v in 3 5; do MNERF_CV_VARIANT=$v timeout 300 python tools/exp/frame_time.py c2 4 2>&1 | tail -1; done
MNERF_CV_VARIANT=5 timeout 300 python tools/exp/frame_time.py c5 2 2>&1 | tail -1
for c in c2 c5; do MNERF_LIB=$PWD/matchnerf_amd/libmnerf_hip_cvst.so timeout 200 python tools/exp/cvt_stats.py $c 2>&1 | tail -11; done
