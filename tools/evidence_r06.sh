#!/bin/bash
# round-6 evidence batch (GPU box): the line the driver records (compact, < 6 KB) + its full record, the default line with
# every configuration, bench lines of the other workloads, rocprofv3 trace / PMC summaries + stamped traffic entries of
# every kernel family quoted, and the striped-vs-linear same-box A/B of the headline batch.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06; mkdir -p $O
python tools/try_striped_rt.py 2>/dev/null | grep -v amdgpu.ids > $O/striped_ab.txt
python tools/try_striped_rt.py 65536 4096 2>/dev/null | grep -v amdgpu.ids >> $O/striped_ab.txt
python tools/try_striped_rt.py 131072 4096 2>/dev/null | grep -v amdgpu.ids >> $O/striped_ab.txt
if [ "${PART:-all}" != "prof" ]; then
python bench.py --steps 20 --warmup 5 --full-json $O/bench_driver_style.json 2>/dev/null | grep '^{' > $O/bench_line_driver_style.json
python bench.py --full-json $O/bench_headline_full.json 2>/dev/null | grep '^{' > $O/bench_line_default.json
wc -c $O/bench_line_driver_style.json $O/bench_line_default.json
run() { name=$1; shift; python bench.py --full-line --no-other-configs "$@" 2>/dev/null | grep '^{' > $O/bench_$name.json; python - <<PY
import json
d=json.loads(open("$O/bench_$name.json").read())
print("$name", d["value"], d["roofline_encode"]["avg_launch_ms"], d["roofline_encode"]["frac"], d["roofline_decode"]["avg_launch_ms"], d["roofline_decode"]["frac"], d["dense_output"]["compact_ms"])
PY
}
run gather --gather --no-cpu-baseline
run headline_linear --layout linear --no-cpu-baseline
run uniform --table uniform --no-cpu-baseline
run 192Ki --chunks 196608 --no-cpu-baseline
run tans --coder tans
run tans_markov1 --coder tans --source markov1 --no-cpu-baseline
run rans_markov1 --source markov1 --no-cpu-baseline
run range_markov1 --coder range --source markov1 --no-cpu-baseline
run aec_static --coder aec --aec-model fixed
run aec_iid --coder aec --aec-model iid --chunks 65536
fi
[ "${PART:-all}" = "lines" ] && exit 0   # PART=lines: only the bench lines above (after traffic.json was refreshed)
for spec in "rans_headline:" "rans_headline_linear:--layout linear" "rans_markov1:--source markov1" "config2_64Ki:--chunks 65536" \
            "tans:--coder tans" "range_uniform1:--coder range --table uniform1" "range_t256:--coder range --table t256" \
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
