for d in 0 8 2 5 1 4 3 6 7; do echo "== knob 27 = $d"; ONLY=1 SWEEP_DIR=$d python scripts/hint_sweep.py 248 121 -1 9 2>&1 | grep "round"; done
