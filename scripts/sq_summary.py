"""Per-kernel medians of every counter found under the given rocprofv3 output directories (one --pmc pass each):
    python scripts/sq_summary.py DIR [DIR ...] [--match regex]"""
import csv, glob, json, re, sys
import numpy as np
args = sys.argv[1:]
match = "score_forward_packed|kde_mode|unroll_median|precompute|gen_kernel|dw_adam|critic_persistent"
if "--match" in args:
    i = args.index("--match"); match = args[i + 1]; del args[i:i + 2]
acc = {}
for d in args:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(match, r["Kernel_Name"])
            if not m:
                continue
            a = acc.setdefault(m.group(0), {}).setdefault(r["Counter_Name"], {})
            a[r["Dispatch_Id"]] = a.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
out = {k: {c: float(np.median(list(v.values()))) for c, v in sorted(cs.items())} for k, cs in acc.items()}
# derived: shares of the dispatch's SIMD-cycles (GRBM_GUI_ACTIVE sums the 8 XCDs; SQ_*_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* count quad-cycles)
for k, c in out.items():
    if not c.get("GRBM_GUI_ACTIVE"):
        continue
    simd = c["GRBM_GUI_ACTIVE"] / 8.0 * 256 * 4
    d = {"simd_cycles": simd}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c: d["mfma_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / simd
    if "SQ_ACTIVE_INST_VALU" in c: d["valu_busy_incl_mfma_issue"] = 4 * c["SQ_ACTIVE_INST_VALU"] / simd
    if "SQ_INSTS_VALU" in c and "SQ_INSTS_MFMA" in c:
        d["valu_instructions_per_mfma"] = (c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / c["SQ_INSTS_MFMA"] if c["SQ_INSTS_MFMA"] else None
        d["non_mfma_valu_cycles_at_4_per_instruction"] = 4 * (c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / simd
    if "SQ_INSTS_VALU_TRANS_F32" in c: d["transcendental_instructions_per_simd_cycle"] = c["SQ_INSTS_VALU_TRANS_F32"] / simd
    if "SQ_WAVE_CYCLES" in c: d["mean_waves_per_simd"] = 4 * c["SQ_WAVE_CYCLES"] / simd
    if "SQ_VALU_MFMA_COEXEC_CYCLES" in c: d["mfma_valu_coexec"] = c["SQ_VALU_MFMA_COEXEC_CYCLES"] / simd
    c["derived"] = d
try:
    sys.path.insert(0, ".")
    from hypad_amd.build import source_digest
    out["source_sha256"] = source_digest()
except Exception:
    pass
print(json.dumps(out, indent=1))
