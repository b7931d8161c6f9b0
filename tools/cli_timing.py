import os, subprocess, sys, tempfile, time
ROOT = os.getcwd(); sys.path.insert(0, ROOT)
import synth
with tempfile.TemporaryDirectory(dir="/dev/shm") as d:
    for n in (20_000_000, 800_000_000):
        g, _ = synth.family(1, n) if n < 100_000_000 else synth.c2(n)
        p = os.path.join(d, "g.fa"); g.tofile(p)
        for rep in range(2):
            t0 = time.perf_counter()
            r = subprocess.run([sys.executable, os.path.join(ROOT, "indexer.py"), p, "g", "15"], capture_output=True, text=True, env=dict(os.environ, PK_TIMING="1"))
            print(n, rep, round(time.perf_counter() - t0, 3)); print(r.stderr[-1500:])
        os.remove(p + ".15.kin")
