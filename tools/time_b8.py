"""encode / decode times of NUM_BITS_OUT = 8, RANGE_FACTOR = 2^8 (and 16 / 2^8), one line (for tools/abn.sh)"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = []
for name, args in (("b8", ["--num-bits-out", "8", "--range-factor", "256"]), ("b16", ["--num-bits-out", "16", "--range-factor", "256"]), ("b4", ["--num-bits-out", "4", "--range-factor", "4096"])):
    o = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-other-configs", "--steps", "20", "--warmup", "5"] + args, capture_output=True, text=True).stdout
    ln = [l for l in o.splitlines() if l.startswith("{")]
    if not ln: out.append(name + " failed"); continue
    d = json.loads(ln[0])
    out.append(f"{name} {d['roofline_encode']['avg_launch_ms']:.4f}/{d['roofline_decode']['avg_launch_ms']:.4f}")
print("  ".join(out))
