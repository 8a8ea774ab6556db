for g in 8192 2560 1280 8192 2560 1280; do echo -n "grid $g: "; HYPAD_KDE_GRID=$g python scripts/time_kde.py | cut -c1-60; done
