#!/bin/bash
# Development knob 27 (direction of the streaming launches of a CG step) x cache-hint masks of the vector kernels, 256^3: it/s and
# per-kernel launch times (GPU box).  Result of round 2: no effect at the default mask 248.
for d in 0 8 2 5 1 4 3 6 7; do echo "== knob 27 = $d"; ONLY=1 SWEEP_DIR=$d python scripts/hint_sweep.py 248 121 -1 9 2>&1 | grep "round"; done
