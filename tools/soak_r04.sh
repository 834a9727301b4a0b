#!/bin/bash
# round-4 soak (GPU box): decoder fuzzing under guard bands, full-size encodes of the kernels that changed this round word
# for word against the any-parameter kernels (+ every decode against the input), the randomised model campaign
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04_soak.txt; mkdir -p gpurun_out
echo "round 4 soak" > $O
echo "== decoder fuzzing: SCL_FUZZ_SEEDS=${FUZZ:-30} rounds x 18 decoder families x 4 kinds of damage (tests/test_gpu_decoder_fuzz.py)" >> $O
SCL_FUZZ_SEEDS=${FUZZ:-30} timeout 3000 python -m pytest tests/test_gpu_decoder_fuzz.py -q -m gpu 2>&1 | tail -2 >> $O
echo "== tools/stress_fast_kernels.py: every encode compared word for word with the any-parameter kernel, every decode with the input" >> $O
run() { echo "== $*" >> $O; env "$@" timeout 1500 python tools/stress_fast_kernels.py 2>&1 | tail -1 >> $O; }
run MODEL=rans REPS=60
run MODEL=rans_b8 REPS=40
run MODEL=tans REPS=30
run MODEL=range REPS=15
run MODEL=order1 REPS=60
run MODEL=iid NCHUNKS=65536 REPS=60
run MODEL=iid NCHUNKS=262144 REPS=10
run MODEL=fixed REPS=40
run MODEL=fixed_k64 REPS=10
run MODEL=order1_k256 NCHUNKS=65536 REPS=3
run MODEL=rans_k64 REPS=10
run MODEL=rans_m3000 REPS=10
echo >> $O; echo "randomised model tests (tests/test_gpu_batch.py -k random; SCL_RANDOM_SEEDS=${SEEDS:-1500})" >> $O
SCL_RANDOM_SEEDS=${SEEDS:-1500} timeout 3000 python -m pytest tests/test_gpu_batch.py tests/test_gpu_wide_alphabets.py -q -m gpu -k "random" -n 4 2>&1 | tail -2 >> $O
cat $O
