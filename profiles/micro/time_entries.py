import importlib.util, os, sys, time, json
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from levelsetfortran_amd import _lib
t0 = time.time()
ent = bench._slab_entries(2, 512, 20, 5, "fast", devices_of=lambda nd: [0] * nd)
t1 = time.time()
print("slab entries (1, 2 slabs on one device, G=512, K=20): %.1f s" % (t1 - t0), [(e.get("n_gpus"), e.get("value"), (e.get("parity") or {}).get("ok"), e.get("error")) for e in ent])
ent = bench._single_process_entries(_lib.load(), 1, 512, 20, 5, "fast", transports=("peer",))
t2 = time.time()
print("one-process entries (1 device, G=512): %.1f s" % (t2 - t1), [(e.get("n_gpus"), e.get("value"), (e.get("parity") or {}).get("ok")) for e in ent])
