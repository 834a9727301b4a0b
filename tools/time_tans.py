import json,subprocess,sys,os
root=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out=[]
for name,args in (("tans",["--coder","tans"]),("rans_rf256",["--range-factor","256"]),("rans_k16m1024?",["--coder","tans","--source","markov1"])):
    o=subprocess.run([sys.executable,os.path.join(root,"bench.py"),"--no-cpu-baseline","--no-other-configs","--steps","20","--warmup","5"]+args,capture_output=True,text=True).stdout
    d=json.loads([l for l in o.splitlines() if l.startswith("{")][0])
    out.append(f"{name} {d['roofline_encode']['avg_launch_ms']:.4f}/{d['roofline_decode']['avg_launch_ms']:.4f}")
print("  ".join(out))
