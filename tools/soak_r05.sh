#!/bin/bash
# round-5 soak (GPU box): decoder fuzzing under guard bands, full-size encodes of the kernels that changed this round word
# for word against the any-parameter kernels (+ every decode against the input), the randomised model campaign
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_soak.txt; mkdir -p gpurun_out
echo "round 5 soak" > $O
echo "== decoder fuzzing: SCL_FUZZ_SEEDS=${FUZZ:-30} rounds x 18 decoder families x 4 kinds of damage (tests/test_gpu_decoder_fuzz.py)" >> $O
SCL_FUZZ_SEEDS=${FUZZ:-30} timeout 3000 python -m pytest tests/test_gpu_decoder_fuzz.py -q -m gpu 2>&1 | tail -2 >> $O
echo "== tools/stress_fast_kernels.py: every encode compared word for word with the any-parameter kernel, every decode with the input" >> $O
run() { echo "== $*" >> $O; env "$@" timeout 1500 python tools/stress_fast_kernels.py 2>&1 | tail -1 >> $O; }
# (round 5 changed no arithmetic: the launch choice of the rANS kernels was refactored, the lone-wave arithmetic decoders got an
# exec-aware wave minimum, the row relay a guard -- every family once, the touched ones longer)
run MODEL=rans REPS=30
run MODEL=rans_b8 REPS=10
run MODEL=tans REPS=10
run MODEL=range REPS=20
run MODEL=range_uniform1 REPS=20
run MODEL=order1 REPS=40
run MODEL=iid NCHUNKS=65536 REPS=40
run MODEL=fixed REPS=10
run MODEL=order1_k256 NCHUNKS=65536 REPS=2
run MODEL=rans_k64 REPS=5
run MODEL=rans_m3000 REPS=5
echo "== round-5 tests with partial waves, damaged chunks, refused rows, 20 repetitions" >> $O
for i in $(seq 20); do timeout 600 python -m pytest tests/test_gpu_round5.py -q -m gpu -x 2>&1 | tail -1; done | sort | uniq -c >> $O
echo >> $O; echo "randomised model tests (tests/test_gpu_batch.py -k random; SCL_RANDOM_SEEDS=${SEEDS:-1500})" >> $O
SCL_RANDOM_SEEDS=${SEEDS:-1500} timeout 3000 python -m pytest tests/test_gpu_batch.py tests/test_gpu_wide_alphabets.py -q -m gpu -k "random" -n 4 2>&1 | tail -2 >> $O
cat $O
