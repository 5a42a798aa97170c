"""Dev tool: A/B two builds of libm2d_hip.so on ONE box (boxes differ by ~10 %): alternate
`bench.py` runs with M2D_LIB pointing at each library and print ms/step per run.
    python tools/ab_bench.py music2dance_amd/lib_old/libm2d_hip.so music2dance_amd/lib/libm2d_hip.so [rounds] [bench args]"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:3]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
extra = sys.argv[4:]
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, M2D_LIB=os.path.join(root, l), M2D_AB_TOLERANT="1")
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-prof"] + extra,
                             env=env, capture_output=True, text=True)
        line = [x for x in out.stdout.splitlines() if x.startswith("{")]
        if not line:
            print(l, "FAILED", out.stderr[-500:]); continue
        d = json.loads(line[-1]); res[l].append(d["ms_per_step"])
        print(r, l, d["ms_per_step"], flush=True)
print({l: (min(v) if v else None) for l, v in res.items()})
