import os, sys, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import synth
with tempfile.TemporaryDirectory(dir="/dev/shm") as d:
    kins = []
    for i in range(3):
        g, _ = synth.family(i, 2_000_000)
        p = os.path.join(d, f"s{i:02d}.fa"); g.tofile(p)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "indexer.py"), p, f"s{i}", "11"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        kins.append(p + ".11.kin")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "merger.py"), os.path.join(d, "proj"), *kins, "--threads", "8"], capture_output=True, text=True)
    print(r.stdout[-1500:]); print(r.stderr[-3000:])
