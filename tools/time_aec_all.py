"""encode / decode times of the tuned arithmetic-coder kernel families, one line (for tools/abn.sh)"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = []
for name, args in (("static", ["--coder", "aec", "--aec-model", "fixed"]), ("iid", ["--coder", "aec", "--aec-model", "iid", "--chunks", "65536"]),
                   ("k16", ["--coder", "aec"]), ("k256", ["--coder", "aec", "--aec-K", "256", "--chunks", "65536", "--steps", "3", "--warmup", "1"])):
    o = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-other-configs", "--steps", "10", "--warmup", "3"] + args,
                       capture_output=True, text=True).stdout
    d = json.loads([ln for ln in o.splitlines() if ln.startswith("{")][0])
    out.append(f"{name} {d['roofline_encode']['avg_launch_ms']:.3f}/{d['roofline_decode']['avg_launch_ms']:.3f}")
print("  ".join(out))
