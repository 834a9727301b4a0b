#!/bin/bash
# collect_profiles.sh: gpurun_out/prof_<name>/ and gpurun_out/$R/ -> profiles/$R_* (the files the docs cite); R=r04 by default
R=${R:-r06}
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
  cp $d/bench.json profiles/${R}_bench_$n.json
  [ -s $d/kernel_trace_summary.txt ] && cp $d/kernel_trace_summary.txt profiles/${R}_${n}_kernel_trace_summary.txt
  [ -s $d/pmc_summary.txt ] && grep -v "^find:" $d/pmc_summary.txt > profiles/${R}_${n}_pmc_summary.txt
  [ -s profiles/${R}_${n}_pmc_summary.txt ] || rm -f profiles/${R}_${n}_pmc_summary.txt
done
for f in gpurun_out/${R}/bench_*.json; do [ -s $f ] && cp $f profiles/${R}_$(basename $f); done
ents=$(for d in gpurun_out/prof_*/; do f=$d/traffic_entry.json; [ -s $f ] && grep -q '"encode"' $f && echo $f; done)
python3 tools/make_traffic_json.py --merge profiles/traffic.json $ents
ls profiles | grep ${R} | wc -l
