"""Latency of one host-visible scalar (mik_dot on a 1,024-element vector: partial kernel + level-2 kernel + the read): the
publish-kernel mailbox (default) against hipMemcpyAsync + event spin (development knob 10 = 1).  GPU box.
    python scripts/scalar_read_latency.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
x = pkg.HipVector.from_numpy(np.arange(1024, dtype=np.float64))
for knob in (0, 1, 0, 1):
    pkg.lib().mik_set_tuning(10, knob)
    for _ in range(200):
        pkg.dot(x, x)
    t0 = time.perf_counter()
    for _ in range(2000):
        v = pkg.dot(x, x)
    dt = (time.perf_counter() - t0) / 2000
    print(f"knob 10 = {knob}: {dt * 1e6:.2f} us per dot (value {v})")
pkg.lib().mik_set_tuning(10, 0)
