"""one short line per bench.py run: encode ms, decode ms, value (for tools/abn.sh A/B timing)"""
import json, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--full-line", "--no-cpu-baseline"] + sys.argv[1:], capture_output=True, text=True).stdout
d = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][0])
print(f"enc {d['roofline_encode']['avg_launch_ms']:.4f} ms  dec {d['roofline_decode']['avg_launch_ms']:.4f} ms  value {d['value']:.0f} MB/s")
