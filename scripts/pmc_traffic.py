"""profiles/*_pmc_traffic.json from two rocprofv3 counter passes of bench.py:

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out_f -o pmc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scoring
    rocprofv3 --pmc WRITE_SIZE ...                               -d out_w ...
    python scripts/pmc_traffic.py out_f out_w > profiles/rNN_pmc_traffic.json

Memory-side bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KB, medians over the launches of each kernel (the gfx950
correction of MI355X_MICROARCH.md: FETCH_SIZE counts half of a wide coalesced read stream)."""
import csv, glob, json, re, sys
import numpy as np

sys.path.insert(0, ".")
KERNELS = ["critic_persistent_kernel", "critic_iteration_kernel", "critic_phase_precompute_kernel", "dw_adam_kernel", "gen_kernel", "pack_generator_kernel",
           "score_forward_packed_kernel", "critic_rows_kernel", "kde_mode_kernel", "unroll_median_kernel", "dtw_error_kernel", "unary_rows", "mobius_add_rows", "rowdist_rows"]


def medians(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    vals = {k: [] for k in KERNELS}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        for k in KERNELS:
            if re.search(r"\b" + k + r"\b", r["Kernel_Name"]):
                vals[k].append(float(r["Counter_Value"]))
    return {k: (float(np.median(v)), len(v)) for k, v in vals.items() if v}


fetch, write = medians(sys.argv[1], "FETCH_SIZE"), medians(sys.argv[2], "WRITE_SIZE")
from hypad_amd.build import source_digest
out = {"source_sha256": source_digest(),
       "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 3 --warmup 1 "
                  "--no-cpu-baseline (two separate passes; scripts/pmc_traffic.py)",
       "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE reads half of a wide coalesced read stream on gfx950 "
                     "(MI355X_MICROARCH.md, HBM)",
       "kernels": {k: {"launches_sampled": fetch[k][1], "FETCH_SIZE_KB_median": fetch[k][0], "WRITE_SIZE_KB_median": write[k][0],
                       "hbm_bytes_per_launch": (2 * fetch[k][0] + write[k][0]) * 1024} for k in KERNELS if k in fetch and k in write}}
print(json.dumps(out, indent=1))
