"""probe: wall time of the CLI under two environment settings, alternating, same inputs.
usage: ab_env.py N_PAIRS VAR A B [reps]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import e2e_bench as eb
n, var, a, b = int(sys.argv[1]), sys.argv[2], sys.argv[3], sys.argv[4]
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 4
with tempfile.TemporaryDirectory(dir="/dev/shm") as d:
    p1, p2 = eb.write_pair(d, n)
    def run(val):
        env = dict(os.environ); env[var] = val
        for o in ("o1", "o2", "os"):  # (truncating a multi-GB tmpfs file costs ~0.4 s by itself)
            if os.path.exists(d + "/" + o):
                os.remove(d + "/" + o)
        if os.environ.get("AB_MODE") == "se":
            cmd = [eb.NEW, "se", "-f", p1, "-t", "sanger", "-o", d + "/o1", "-a", "1"]
        else:
            cmd = [eb.NEW, "pe", "-f", p1, "-r", p2, "-t", "sanger", "-o", d + "/o1", "-p", d + "/o2", "-s", d + "/os", "-a", "1"]
        t0 = time.perf_counter()
        subprocess.run(cmd, capture_output=True, env=env, check=True)
        return time.perf_counter() - t0
    run(a)
    res = {a: [], b: []}
    for _ in range(reps):
        for v in (a, b):
            res[v].append(run(v))
    for v in (a, b):
        print("%s=%s: %s  median %.3f s" % (var, v, " ".join("%.3f" % t for t in res[v]), sorted(res[v])[len(res[v]) // 2]))
