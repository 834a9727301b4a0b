#!/bin/bash
# one short line from bench.py: value, encode ms, decode ms  (arguments are passed to bench.py)
python bench.py --full-line --no-cpu-baseline --steps 30 --warmup 10 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['coder'], d['value'], 'MB/s  enc', d['roofline_encode']['avg_launch_ms'], 'ms  dec', d['roofline_decode']['avg_launch_ms'], 'ms')"
