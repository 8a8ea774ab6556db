for rep in 1 2; do
echo -n "default (stand-alone launches, zero-copy noise): "; python scripts/time_drop_in.py 2>&1 | tail -1
echo -n "H2D copy of the noise:                           "; HYPAD_DROPIN_ZEROCOPY=0 python scripts/time_drop_in.py 2>&1 | tail -1
echo -n "phase form of the critic iterations:             "; HYPAD_ENGINE_ITER_PHASE=1 python scripts/time_drop_in.py 2>&1 | tail -1
done
