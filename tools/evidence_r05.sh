#!/bin/bash
# round-5 evidence batch (GPU box): the driver-style line, the full default line (other_configs + CPU baselines), bench
# lines of the other workloads, and rocprofv3 trace / PMC summaries + stamped traffic entries of every kernel family quoted
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05; mkdir -p $O
python bench.py --steps 20 --warmup 5 2>/dev/null | grep '^{' > $O/bench_driver_style.json
python bench.py 2>/dev/null | grep '^{' > $O/bench_headline_full.json
run() { name=$1; shift; python bench.py --no-other-configs "$@" 2>/dev/null | grep '^{' > $O/bench_$name.json; python - <<PY
import json
d=json.loads(open("$O/bench_$name.json").read())
print("$name", d["value"], d["roofline_encode"]["avg_launch_ms"], d["roofline_encode"]["frac"], d["roofline_decode"]["avg_launch_ms"], d["roofline_decode"]["frac"], d["dense_output"]["compact_ms"])
PY
}
run gather --gather --no-cpu-baseline
run uniform --table uniform --no-cpu-baseline
run 192Ki --chunks 196608 --no-cpu-baseline
run tans --coder tans
run tans_markov1 --coder tans --source markov1 --no-cpu-baseline
run range_markov1 --coder range --source markov1 --no-cpu-baseline
run aec_static --coder aec --aec-model fixed
run aec_iid --coder aec --aec-model iid --chunks 65536
run aec_k256_256Ki --coder aec --aec-K 256 --chunks 262144 --steps 2 --warmup 1 --no-cpu-baseline
[ "${PART:-all}" = "lines" ] && exit 0   # PART=lines: only the bench lines above (after traffic.json was refreshed)
for spec in "rans_headline:" "rans_markov1:--source markov1" "config2_64Ki:--chunks 65536" "tans:--coder tans" \
            "range_uniform1:--coder range --table uniform1" "range_t256:--coder range --table t256" \
            "rans_b8:--num-bits-out 8 --range-factor 256" "aec_k16:--coder aec --steps 5 --warmup 2" \
            "aec_k256_sparse:--coder aec --aec-K 256 --chunks 65536 --steps 3 --warmup 1" \
            "aec_static:--coder aec --aec-model fixed" "aec_iid:--coder aec --aec-model iid --chunks 65536"; do
  BENCH_ARGS="${spec#*:}" bash tools/prof_bench.sh ${spec%%:*} > /dev/null 2>&1
  python - <<PY
import json
d=json.loads(open("gpurun_out/prof_${spec%%:*}/bench.json").read())
print("${spec%%:*}", d["value"], d["roofline_encode"]["avg_launch_ms"], d["roofline_encode"]["frac"], d["roofline_decode"]["avg_launch_ms"], d["roofline_decode"]["frac"])
PY
done
