import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); orc = g.load_oracle()
A, b = orc.advdiff(12, 1000.0)
dA = pkg.HipCSR(A.n, A.n, A.colptr, A.rowval, A.nzval, index_base=A.index_base)
for knob in (4, 0):
    pkg.lib().mik_set_tuning(5, knob)
    x, ch = pkg.gmres(dA, pkg.HipVector.from_numpy(b), restart=10, log=True)
    print("knob5", knob, "iters", ch.iters, "first residuals", [float(v) for v in ch["resnorm"][:4]], flush=True)
