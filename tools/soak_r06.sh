#!/bin/bash
# round-6 soak (GPU box), on the final sources: decoder fuzzing under guard bands incl. the four striped-slot readers, the
# randomised model campaign (every tuned rANS / tANS model also on striped slots: same descriptors, same dense and framed
# bytes, same decode), the striped layout's own tests repeated (1 GiB word for word, lanes that drift apart, lengths that lie)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06_soak.txt; mkdir -p gpurun_out
echo "round 6 soak" > $O
echo "== decoder fuzzing: SCL_FUZZ_SEEDS=${FUZZ:-120} rounds x 24 decoder families x 4 kinds of damage (tests/test_gpu_decoder_fuzz.py)" >> $O
SCL_FUZZ_SEEDS=${FUZZ:-120} timeout 3000 python -m pytest tests/test_gpu_decoder_fuzz.py -q -m gpu -n 4 2>&1 | tail -2 >> $O
echo "== randomised model tests (tests/test_gpu_batch.py tests/test_gpu_wide_alphabets.py -k random; SCL_RANDOM_SEEDS=${SEEDS:-5000})" >> $O
SCL_RANDOM_SEEDS=${SEEDS:-5000} timeout 3000 python -m pytest tests/test_gpu_batch.py tests/test_gpu_wide_alphabets.py -q -m gpu -k "random" -n 4 2>&1 | tail -2 >> $O
echo "== tests/test_gpu_striped.py, ${REPS:-25} repetitions" >> $O
for i in $(seq ${REPS:-25}); do timeout 900 python -m pytest tests/test_gpu_striped.py -q -m gpu -x 2>&1 | tail -1; done | sed 's/ in [0-9.]*s.*//' | sort | uniq -c >> $O
echo "== full-size stress of every tuned kernel family (word for word against the any-parameter kernels + oracle samples), 3 repetitions each" >> $O
timeout 3000 python -m pytest tests/test_gpu_batch.py -q -m gpu -k "full_occupancy_stress" 2>&1 | tail -2 >> $O
cat $O
