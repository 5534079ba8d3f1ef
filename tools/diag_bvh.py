"""Diagnostic: device builder vs oracle on larger inputs, distinct keys vs ties."""
import importlib, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding as ob
pta = importlib.import_module("path-tracer_amd")
wd = pta.work_dispatcher
rd = pta.PtkRenderingDevice(0)
o = ob.OracleBvhBuilder(); d = wd.DeviceBvhBuilder(rd)
rng = np.random.default_rng(0)
def run(tag, aabbs, t):
    n = aabbs.shape[0]
    ids = (np.arange(n) + (3 << 24)).astype(np.int32)
    cn, ci = o.read(o.build(8, t, aabbs, ids))
    gn, gi = d.read(d.build(8, t, aabbs, ids))
    bad = np.argwhere(ci != gi).ravel()
    print(tag, "n", n, "type", t, "ids_ok", len(bad) == 0, "nodes_ok", cn == gn, "first", bad[:6].tolist())
for n in [3000, 4000, 4096, 4097, 5000, 8224, 20000]:
    c = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    aabbs = np.concatenate([c - 0.01, np.full((n, 3), 0.02, np.float32)], axis=1).astype(np.float32)
    run("distinct", aabbs, 1)
    c = rng.integers(0, 50, (n, 3)).astype(np.float32)
    aabbs = np.concatenate([c - 0.25, np.full((n, 3), 0.5, np.float32)], axis=1).astype(np.float32)
    run("ties50", aabbs, 1)
    s = rng.uniform(0.01, 0.3, (n, 3)).astype(np.float32)
    c = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    run("varsize", np.concatenate([c, s], axis=1).astype(np.float32), 1)
