for s in 12 16 24 32; do for a in 0 1 2; do echo -n "signals $s aux $a: "; HYPAD_AUX_STREAMS=$a python scripts/time_signals.py --spg $s --reps 10 | tail -2 | tr '\n' ' '; echo; done; done
