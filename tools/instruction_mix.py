"""profiles/<R>_<family>_pmc_summary.txt (+ the family's bench line for the symbol count) -> profiles/<R>_instruction_mix.txt (R=r06):
per-symbol instruction counts and unit occupancy of every coder kernel.  Formulas in the header it writes."""
import glob, json, os, re, sys
R = os.environ.get("R", "r06")  # the round whose profiles are summarised
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "profiles")
HEADER = f"""Instruction mix and unit occupancy of every kernel family, from the {R} PMC passes (profiles/{R}_<family>_pmc_summary.txt;
batches as in profiles/{R}_bench_<family>.json; tools/instruction_mix.py: counters / symbols x 64 lanes; valu_busy =
4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); lds_busy = SQ_LDS_IDX_ACTIVE / (256 CUs x cycles))
"""
out = [HEADER]
order = ["rans_headline", "rans_headline_linear", "rans_markov1", "rans_b8", "tans", "range_uniform1", "range_t256", "aec_static", "aec_iid", "aec_k16",
         "aec_k256_wide", "aec_k256_sparse", "config2_64Ki"]
fams = [os.path.basename(f)[4:-16] for f in glob.glob(os.path.join(prof, f"{R}_*_pmc_summary.txt"))]
for fam in [f for f in order if f in fams] + sorted(f for f in fams if f not in order):
    bench = os.path.join(prof, f"{R}_bench_{fam}.json")
    if not os.path.exists(bench):
        continue
    cfg = json.load(open(bench))["config"]
    symbols = cfg["chunks_per_gpu"] * cfg["chunk_len"]
    rows = {}
    for ln in open(os.path.join(prof, f"{R}_{fam}_pmc_summary.txt")):
        m = re.match(r"(.+?)\s+([A-Z_a-z0-9]+)\s+mean/dispatch=([0-9.e+]+)\s+dispatches", ln)
        if m and re.search(r"rans_|tans_|range_|aec_", m.group(1)):
            rows.setdefault(m.group(1).strip(), {})[m.group(2)] = float(m.group(3))
    for kern, c in sorted(rows.items(), key=lambda kv: ("decode" in kv[0], kv[0])):
        if "SQ_INSTS_VALU" not in c or "GRBM_GUI_ACTIVE" not in c or not re.search("encode|decode", kern):
            continue
        per = lambda k: c.get(k, 0.0) * 64 / symbols
        cycles = c["GRBM_GUI_ACTIVE"] / 8
        valu_busy = 4 * c.get("SQ_ACTIVE_INST_VALU", 0) / (1024 * cycles)
        lds_busy = c.get("SQ_LDS_IDX_ACTIVE", 0) / (256 * cycles)
        conf = c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 0), 1)
        side = "decode" if "decode" in kern else "encode"
        out.append(f"{fam:15s} {side} VALU/sym={per('SQ_INSTS_VALU'):6.1f} SALU/sym={per('SQ_INSTS_SALU'):5.1f} "
                   f"LDS/sym={per('SQ_INSTS_LDS'):5.1f} VMEM/sym={per('SQ_INSTS_VMEM_RD') + per('SQ_INSTS_VMEM_WR'):5.2f} "
                   f"waves={int(c.get('SQ_WAVES', 0)):6d} valu_busy={valu_busy:.2f} lds_busy={lds_busy:.2f} "
                   f"lds_conflict_share={conf:.2f}   {kern}")
open(os.path.join(prof, f"{R}_instruction_mix.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out[1:]))
