"""profiles/r0N_mfma_util.json from one rocprofv3 counter pass over scripts/profile_mfma_target.py:

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d out -o pmc -- python3 scripts/profile_mfma_target.py
    python scripts/mfma_util.py out > profiles/r0N_mfma_util.json

MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs): the share of the chip's SIMD-cycles
during the dispatch in which the matrix pipe was busy (the gfx94x MfmaUtil formula; ROCm 7.2 ships no gfx950 derived counters).
The algorithmic figure next to it: fp32 MFMAs the kernel's algorithm needs (SURVEY.md §8d MACs / 1024 per v_mfma_f32_16x16x4_f32)
x 32 cycles each, over the same SIMD-cycles -- what the counter would read with no padding, no redundant tiles, no idle CUs."""
import csv, glob, json, re, sys
import numpy as np

sys.path.insert(0, ".")
KERNELS = {"score_forward_packed_kernel": (340312.0 / 2 - 3220.0) * 125_000,       # MACs per launch (algorithmic; critic_x's 3 220 per window are the launch below)
           "critic_rows_kernel": 3220.0 * 125_000,
           "lstm_fwd(_lds[0-9]?)?_kernel": None,
           "gen_kernel": 561824.0 * 64, "dw_adam_kernel": 281872.0 * 64,
           "critic_persistent_kernel": (40400.0 + 156936.0) * 64 * 145}
d = sys.argv[1]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
acc = {}
lstm_ids = set()
rows_all = list(csv.DictReader(open(f)))
for r in rows_all:
    if re.search("lstm_fwd(_lds[0-9]?)?_kernel", r["Kernel_Name"]):
        lstm_ids.add(int(r["Dispatch_Id"]))
lstm_sorted = sorted(lstm_ids)
lstm_first = set(lstm_sorted[: len(lstm_sorted) // 2])           # the target runs the 100 -> 2x50 shape first, then 128 -> 2x64
LSTM_MACS = {"lstm_bidir_fwd (100 -> 2 x 50, 200 000 rows)": 2 * 3 * 50 * 100 * 200_000.0, "lstm_bidir_fwd (128 -> 2 x 64, 200 000 rows)": 2 * 3 * 64 * 128 * 200_000.0}      # (i, g, o: the f gate meets c0 = 0 and is never formed)
for r in rows_all:
    for k in KERNELS:
        if re.search(k, r["Kernel_Name"]):
            key = k
            if k == "lstm_fwd(_lds[0-9]?)?_kernel":
                key = list(LSTM_MACS)[0 if int(r["Dispatch_Id"]) in lstm_first else 1]
            acc.setdefault(key, {}).setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
            acc[key][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
out = {"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -- "
                  "python3 scripts/profile_mfma_target.py", "formula": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 * 4)",
       "kernels": {}}
try:
    from hypad_amd.build import source_digest
    out["source_sha256"] = source_digest()
except Exception:
    pass
for k, c in acc.items():
    med = {name: float(np.median(list(v.values()))) for name, v in c.items()}
    ent = {"launches_sampled": len(next(iter(c.values()))), **{name + "_median": v for name, v in med.items()}}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in med and med.get("GRBM_GUI_ACTIVE"):
        simd_cycles = med["GRBM_GUI_ACTIVE"] / 8.0 * 256 * 4
        ent["mfma_util"] = med["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles
        macs = KERNELS.get(k) if k in KERNELS else LSTM_MACS.get(k)
        if macs:
            ent["algorithmic_mfma_cycles"] = macs / 1024.0 * 32.0
            ent["algorithmic_util"] = ent["algorithmic_mfma_cycles"] / simd_cycles
            ent["issued_over_algorithmic"] = med["SQ_VALU_MFMA_BUSY_CYCLES"] / ent["algorithmic_mfma_cycles"]
    out["kernels"][k] = ent
print(json.dumps(out, indent=1))
