#!/bin/bash
# collect_profiles.sh: gpurun_out/prof_<name>/ and gpurun_out/r03/ -> profiles/r03_* (the files the docs cite)
cd "$(dirname "$0")/.."
for d in gpurun_out/prof_*/; do
  n=$(basename $d); n=${n#prof_}
  [ -s $d/bench.json ] || continue
  python3 - "$d" <<'PY' || continue
import json,sys
d=sys.argv[1]
b=json.loads(open(d+"/bench.json").read())
sys.exit(0 if "traffic_key" in b else 1)
PY
  cp $d/bench.json profiles/r03_bench_$n.json
  [ -s $d/kernel_trace_summary.txt ] && cp $d/kernel_trace_summary.txt profiles/r03_${n}_kernel_trace_summary.txt
  [ -s $d/pmc_summary.txt ] && grep -v "^find:" $d/pmc_summary.txt > profiles/r03_${n}_pmc_summary.txt
  [ -s profiles/r03_${n}_pmc_summary.txt ] || rm -f profiles/r03_${n}_pmc_summary.txt
done
for f in gpurun_out/r03/bench_*.json; do [ -s $f ] && cp $f profiles/r03_$(basename $f); done
ents=$(for d in gpurun_out/prof_*/; do f=$d/traffic_entry.json; [ -s $f ] && grep -q '"encode"' $f && echo $f; done)
python3 tools/make_traffic_json.py --merge profiles/traffic.json $ents
ls profiles | grep r03 | wc -l
