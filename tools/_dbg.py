import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from cora_amd import _lib
ctx = _lib.get_context()
rng = np.random.default_rng(103)
st = rng.bit_generator.state["state"]
n = 4000
g, nraw = ctx.normals_pcg64(st["state"], st["inc"], n)
dev = g.cpu().numpy(); ref = rng.standard_normal(n + 100)
bad = np.flatnonzero(dev != ref[:n])
print("nraw", nraw, "first bad", bad[:10])
i0 = bad[0]
for d in range(-40, 41):
    if np.array_equal(dev[i0 + 20: i0 + 60], ref[i0 + 20 + d: i0 + 60 + d]): print("dev[i] == ref[i%+d] after first bad" % d)
# where does each dev value come from in ref
idx = {v: i for i, v in enumerate(ref)}
src = [idx.get(v, -1) for v in dev[980:1040]]
print(src)
