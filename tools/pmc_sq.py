#!/usr/bin/env python
"""SQ / GRBM / TCC counters per kernel from rocprofv3 --pmc passes (one counter_collection.csv per pass) -> one JSON with
the derived fractions DESIGN.md §3 quotes beside the time-derived roofline fractions.

usage: pmc_sq.py out.json pass1.csv [pass2.csv ...]

Units (MI355X_MICROARCH.md "rocprofv3 PMC slots" and the s_memtime row of the cycle-constants table):
  * GRBM_GUI_ACTIVE is summed over the 8 XCDs -> kernel cycles = GRBM_GUI_ACTIVE / 8;
  * SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe busy CYCLES summed over all SIMDs (1024): mfma_busy = that / (1024 x cycles);
  * SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count QUAD-cycles summed over waves: reported as shares of SQ_WAVE_CYCLES
    (WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES);
  * SQ_INSTS_* are instruction counts (per wave-instruction), reported per launch and as ratios to SQ_INSTS_MFMA;
  * FETCH_SIZE / WRITE_SIZE as in tools/pmc_traffic.py (x1024 B, fetch x2 on gfx950 for 16-B-lane reads).
Only the kernels named in KEEP are summarised (prefix match on the demangled name)."""
import collections
import csv
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEEP = ("gemm_w8a8_fi_kernel", "gemm_w8a8_kernel", "gemm_w8a8_m32_kernel", "attn_kernel", "attn_i8_q64_kernel",
        "vae_conv3_kernel", "vae_conv2_kernel", "vae_conv_kernel", "vae_chan_rms_kernel", "gemm_bf16_kernel", "ln_apply_quant_kernel", "qk_norm_rope_kernel",
        "sage_quant_pool_kernel", "linear_out_kernel", "linear_kv_partial_kernel")
N_SIMD = 1024
N_XCD = 8


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:90]


def main():
    out_path, paths = sys.argv[1], sys.argv[2:]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in paths:
        for r in csv.DictReader(open(p)):
            k = short(r["Kernel_Name"])
            if not k.startswith(KEEP):
                continue
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    kernels = {}
    for k, cs in sorted(agg.items()):
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        d = {"launches": max(len(v) for v in cs.values()), "counters_per_launch": {c: round(v, 1) for c, v in sorted(m.items())}}
        cyc = m.get("GRBM_GUI_ACTIVE", 0.0) / N_XCD
        if cyc:
            d["kernel_cycles"] = round(cyc)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
                d["mfma_busy_frac"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (N_SIMD * cyc), 4)
            if "SQ_BUSY_CYCLES" in m:
                d["sq_busy_over_kernel_cycles"] = round(m["SQ_BUSY_CYCLES"] / cyc, 3)
        wc = m.get("SQ_WAVE_CYCLES")
        if wc:
            for c, name in (("SQ_WAIT_ANY", "wave_parked_share"), ("SQ_WAIT_INST_ANY", "issue_stall_share"),
                            ("SQ_ACTIVE_INST_ANY", "issuing_share"), ("SQ_ACTIVE_INST_VALU", "issuing_valu_share"),
                            ("SQ_ACTIVE_INST_LDS", "issuing_lds_share"), ("SQ_WAIT_INST_LDS", "lds_issue_stall_share")):
                if c in m:
                    d[name] = round(m[c] / wc, 4)
        nm = m.get("SQ_INSTS_MFMA")
        if nm:
            for c, name in (("SQ_INSTS_VALU", "valu_per_mfma"), ("SQ_INSTS_LDS", "lds_insts_per_mfma"),
                            ("SQ_INSTS_VMEM_RD", "vmem_rd_per_mfma"), ("SQ_INSTS_SALU", "salu_per_mfma")):
                if c in m:
                    d[name] = round(m[c] / nm, 3)   # (SQ_INSTS_VALU includes the MFMAs themselves)
        if "SQ_LDS_BANK_CONFLICT" in m and "SQ_ACTIVE_INST_LDS" in m and m["SQ_ACTIVE_INST_LDS"]:
            d["lds_bank_conflict_over_lds_active"] = round(m["SQ_LDS_BANK_CONFLICT"] / m["SQ_ACTIVE_INST_LDS"], 4)
        if "FETCH_SIZE" in m:
            d["fetch_bytes_per_launch"] = 2.0 * 1024.0 * m["FETCH_SIZE"]
        if "WRITE_SIZE" in m:
            d["write_bytes_per_launch"] = 1024.0 * m["WRITE_SIZE"]
        kernels[k] = d
    head = None
    hp = os.path.join(ROOT, ".td_head")
    if os.path.exists(hp):
        head = open(hp).read().strip()
    sys.path.insert(0, ROOT)
    try:
        from bench import kernel_source_digest
        dig = kernel_source_digest()
    except Exception:
        dig = None
    json.dump({"commit": head, "kernel_source_digest": dig,
               "note": "rocprofv3 --pmc passes (SQ sets + GRBM_GUI_ACTIVE; FETCH_SIZE / WRITE_SIZE in their own passes) over "
                       "bench.py --num-steps 1 --no-graph (token split off) and one VAE decode; units in tools/pmc_sq.py",
               "kernels": kernels}, open(out_path, "w"), indent=1)
    for k, d in kernels.items():
        print(f"{k[:70]:70s} n={d['launches']:4d} mfma_busy {d.get('mfma_busy_frac')}  issuing {d.get('issuing_share')} "
              f"(valu {d.get('issuing_valu_share')}) parked {d.get('wave_parked_share')} stall {d.get('issue_stall_share')} "
              f"valu/mfma {d.get('valu_per_mfma')} lds/mfma {d.get('lds_insts_per_mfma')}")


if __name__ == "__main__":
    main()
