#!/bin/bash
# round-3 soak (GPU box): the kernels that changed this round, full-size encodes word for word against the any-parameter
# kernels + every decode against the input, then the randomised model campaign
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03_soak.txt; mkdir -p gpurun_out
echo "round 3: tools/stress_fast_kernels.py, every encode compared word for word with the any-parameter kernel, every decode with the input" > $O
run() { echo "== $*" >> $O; env "$@" python tools/stress_fast_kernels.py 2>&1 | tail -1 >> $O; }
run MODEL=order1 REPS=100
run MODEL=order1_k256 NCHUNKS=65536 REPS=20
run MODEL=tans REPS=60
run SCL_TANS_KERNELS=table MODEL=tans REPS=30
run SCL_AEC_WIDE=dense MODEL=order1_k256 NCHUNKS=65536 REPS=8
run MODEL=rans REPS=60
run MODEL=range REPS=15
run MODEL=fixed REPS=10
run MODEL=iid NCHUNKS=65536 REPS=10
echo >> $O; echo "randomised model tests (tests/test_gpu_batch.py -k random; SCL_RANDOM_SEEDS=1500)" >> $O
SCL_RANDOM_SEEDS=1500 python -m pytest tests/test_gpu_batch.py tests/test_gpu_wide_alphabets.py -q -m gpu -k "random" -n 4 2>&1 | tail -2 >> $O
cat $O
