"""Figures read out of the COMMITTED rocprofv3 profiles of the default bench command (labelled as such in the JSON line: bench.py
cannot read hardware counters itself)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PMC_SUMMARY = "profiles/round6_pmc_summary.json"      # tools/pmc_summary.py over rocprofv3 --pmc passes of the default command
KERNEL_STATS = "profiles/round6_kernel_stats.csv"     # rocprofv3 --kernel-trace --stats of the default command


# ---- figures out of the committed profiles (labelled as such) --------------------------------------------------------------
def traffic_from_profile(kernel, batch):
    """HBM bytes per launch of the kernel whose name contains `kernel`, from the committed rocprofv3 PMC summary of the default
    bench command (separate --pmc passes for FETCH_SIZE and WRITE_SIZE, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes
    for gfx950).  bench.py cannot read PMC counters itself: this is a figure FROM A PROFILE of the same command, labelled as
    such; None when the summary is missing, was taken at another batch size, or lacks the kernel."""
    try:
        with open(os.path.join(ROOT, PMC_SUMMARY)) as f:
            j = json.load(f)
        if int(j.get("batch", -1)) != int(batch):
            return None
        for name, row in j.items():
            if isinstance(row, dict) and kernel in name and "hbm_read_bytes" in row and "hbm_write_bytes" in row:
                return {"hbm_bytes_per_launch": float(row["hbm_read_bytes"]) + float(row["hbm_write_bytes"]),
                        "hbm_read_bytes": float(row["hbm_read_bytes"]), "hbm_write_bytes": float(row["hbm_write_bytes"]),
                        "kernel": name, "dispatches": row.get("dispatches"), "source": PMC_SUMMARY}
    except Exception:
        pass
    return None


def valu_from_profile(kernel, batch):
    """Vector-ALU issue occupancy of the kernel whose name contains `kernel`, from the committed PMC summary: SQ_ACTIVE_INST_VALU (in
    units of 4 cycles, summed over all SIMDs) x 4 / 1024 SIMDs = cycles a SIMD's vector pipe was issuing, against GRBM_GUI_ACTIVE / 8
    XCDs = cycles of one launch.  The roofline of a kernel whose arithmetic is fixed by bit-exactness (binary64 in the reference's
    order: nothing to trade) is its instruction issue, not HBM."""
    try:
        with open(os.path.join(ROOT, PMC_SUMMARY)) as f:
            j = json.load(f)
        if int(j.get("batch", -1)) != int(batch):
            return None
        for name, row in j.items():
            if isinstance(row, dict) and kernel in name and "SQ_ACTIVE_INST_VALU" in row and "GRBM_GUI_ACTIVE" in row:
                active = float(row["SQ_ACTIVE_INST_VALU"]) * 4.0 / 1024.0
                launch = float(row["GRBM_GUI_ACTIVE"]) / 8.0
                return {"valu_issue_cycles_per_simd": active, "launch_cycles": launch, "frac_of_issue_bound": active / launch,
                        "valu_instructions_per_launch": float(row.get("SQ_INSTS_VALU", 0.0)),
                        "lds_bank_conflict_share_of_lds_active": (float(row["SQ_LDS_BANK_CONFLICT"]) / float(row["SQ_LDS_IDX_ACTIVE"])
                                                                  if row.get("SQ_LDS_IDX_ACTIVE") else None),
                        "lds_active_share_of_launch": (float(row["SQ_LDS_IDX_ACTIVE"]) * 4.0 / 1024.0 / launch if row.get("SQ_LDS_IDX_ACTIVE") else None),
                        "kernel": name, "source": PMC_SUMMARY}
    except Exception:
        pass
    return None


def clock_from_profile(kernel, flops, batch):
    """Effective shader clock and MFMA cycle fraction of the kernel whose name contains `kernel` inside the step, from the committed
    profiles: GRBM_GUI_ACTIVE (busy cycles, summed over the 8 XCDs by rocprofv3) of the PMC summary / 8 = cycles of one launch;
    / the average duration of the kernel-trace summary = the clock the power management granted (MI355X_MICROARCH.md, "DVFS
    give-back"); the MFMA work of `flops` is flops / (1024 SIMDs x 1024 flop per cycle) cycles.  None when a profile lacks it."""
    try:
        import csv
        with open(os.path.join(ROOT, PMC_SUMMARY)) as f:
            j = json.load(f)
        if int(j.get("batch", -1)) != int(batch):
            return None
        cyc = next(float(r["GRBM_GUI_ACTIVE"]) / 8.0 for n, r in j.items() if isinstance(r, dict) and kernel in n and "GRBM_GUI_ACTIVE" in r)
        with open(os.path.join(ROOT, KERNEL_STATS)) as f:
            ns = next(float(r["AverageNs"]) for r in csv.DictReader(f) if kernel in r["Name"])
        mfma = flops / (1024.0 * 1024.0)
        return {"busy_cycles_per_launch": cyc, "effective_clock_ghz": cyc / ns, "mfma_cycles": mfma, "mfma_cycle_frac": mfma / cyc,
                "note": "the chip clocks down under dense MFMA work on random operands (DESIGN.md 3.10): `frac` is against the 2.4 GHz peak, "
                        "`mfma_cycle_frac` is the share of the launch's cycles that are MFMA issue cycles", "source": PMC_SUMMARY + " + " + KERNEL_STATS}
    except Exception:
        return None


def stats_from_profile(kernel):
    """Average duration of the kernel whose name contains `kernel` in the committed rocprofv3 --kernel-trace --stats summary of the
    default bench command: what the live in-step figure must agree with."""
    try:
        import csv
        with open(os.path.join(ROOT, KERNEL_STATS)) as f:
            for row in csv.DictReader(f):
                if kernel in row["Name"]:
                    return {"avg_kernel_ms": float(row["AverageNs"]) * 1e-6, "calls": int(row["Calls"]), "kernel": row["Name"], "source": KERNEL_STATS}
    except Exception:
        pass
    return None
