"""The roofline block of a bench line: algorithmic bytes against the HBM peak, PMC traffic, VALU issue."""
from .pmc import pmc_bytes
from .workload import HBM_PEAK_GBS, VALU_PEAK


def roofline_entry(job, launch_s, steps_in_launch, counters, bytes_per_step):
    """roofline entries of one kind of launch"""
    alg = bytes_per_step * job.particles_rank * steps_in_launch
    eq = alg / launch_s / 1e9 / HBM_PEAK_GBS
    e = {"avg_launch_ms": launch_s * 1e3, "steps_per_launch": steps_in_launch,
         "ms_per_step": launch_s * 1e3 / steps_in_launch,
         "achieved": alg / launch_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "equivalent_frac": eq,
         "algorithmic_bytes_per_launch": alg}
    tb = pmc_bytes(counters)
    e["traffic"] = tb
    if tb is not None:
        e["hbm_physical"] = {"achieved": tb / launch_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": tb / launch_s / 1e9 / HBM_PEAK_GBS, "bytes_over_algorithmic": tb / alg,
                             "counted_at": "the L2s' memory-side interface (TCC_EA0 read / write requests: every read of the integrator is "
                                           "128 B and DRAM-destined - profiles/r5_a_single_step_memory_path_pmc.txt); requests the 256 MB "
                                           "Infinity Cache serves are in it - the decoded flow plane lives there, so of a single step's "
                                           "traffic HBM itself moves the streams (state in + out, 4 B of slot order per particle) and the "
                                           "plane once: ~1.15 x the algorithmic bytes"}
    if counters and "SQ_INSTS_VALU" in counters:
        v = counters["SQ_INSTS_VALU"]
        e["valu"] = {"wave_insts_per_launch": v, "per_wave_step": v / (job.particles_rank / 64.0 * steps_in_launch),
                     "achieved": v / launch_s, "peak": VALU_PEAK, "unit": "wave-instr/s", "frac": v / launch_s / VALU_PEAK}
        if "clock_ghz" in counters:      # under the profiler (launches run a few % slower there)
            e["valu"]["clock_ghz"] = counters["clock_ghz"]
            e["valu"]["profiled_launch_ms"] = counters.get("profiled_launch_ms")
            e["valu"]["issue_utilization_at_held_clock"] = counters.get("valu_issue_utilization")
            e["valu"]["note"] = "peak = 256 CU x 4 SIMD x 2.4 GHz / 2 cycles per wave64 instruction; clock_ghz = GRBM_GUI_ACTIVE / 8 / launch " \
                                "duration and issue_utilization = 2 x SQ_INSTS_VALU / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), both per " \
                                "dispatch in the PMC child run: the clock the chip held under this load and the share of its issue " \
                                "slots the launch used at that clock"
    if counters and "SQ_LDS_IDX_ACTIVE" in counters and counters["SQ_LDS_IDX_ACTIVE"] > 0:
        e["lds"] = {"bank_conflict_share": counters.get("SQ_LDS_BANK_CONFLICT", 0.0) / counters["SQ_LDS_IDX_ACTIVE"]}
    return e


def bind(e, fused):
    """`bound` and `frac` of an entry: the fraction of the bound it names, never the equivalent bandwidth of a
    register-resident launch.  (bench.py relabels ONE entry after this - the line's headline, to the bench contract's
    `frac = achieved / peak` against the HBM roofline, keeping what this function found as `limited_by` / `limited_by_frac`;
    its docstring says so, and `frac_is` of every entry names the rule it follows.)  One step per launch streams its algorithmic bytes: HBM, frac = algorithmic / peak.
    A fused launch is bound by whichever of VALU issue and physical HBM traffic it uses more of (PMC child runs);
    without counters the bound is not known and frac stays null."""
    if not fused:
        e["bound"], e["frac"] = "hbm", e["equivalent_frac"]
        e["frac_is"] = "algorithmic bytes / launch duration / HBM peak (a single-step launch streams them)"
        return e
    v = (e.get("valu") or {}).get("frac")
    h = (e.get("hbm_physical") or {}).get("frac")
    if v is None and h is None:
        e["bound"], e["frac"] = "valu", None
        e["frac_is"] = "unknown: the PMC child runs gave no counters (equivalent_frac is the SURVEY.md 8d figure)"
    elif h is None or (v is not None and v >= h):
        e["bound"], e["frac"] = "valu", v
        e["frac_is"] = "valu.frac: wave64 VALU instructions per second / the chip's issue peak at 2.4 GHz"
    else:
        e["bound"], e["frac"] = "hbm", h
        e["frac_is"] = "hbm_physical.frac: PMC bytes (2 x FETCH_SIZE + WRITE_SIZE) / launch duration / HBM peak"
    return e


