"""rocprofv3 PMC child passes of bench.py: short child runs of bench.py itself (--pmc-child), one pass per counter group."""
import json
import os
import subprocess
import sys

from .workload import ROOT

BENCH = os.path.join(ROOT, "bench.py")

# ---- rocprofv3 PMC child passes ----------------------------------------------------------------------------
PMC_PASSES = (("FETCH_SIZE", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE"), ("WRITE_SIZE", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"))


def kernel_class(name):
    """Which of the measured launches a kernel-trace row belongs to (template arguments: <FAST, NOISE, ...>)."""
    for base in ("logic_fused_packed_kernel", "logic_fused_kernel", "logic_packed_kernel", "logic_sorted_kernel", "logic_kernel"):
        if "th::" + base + "<" in name:
            args = name.split(base + "<", 1)[1].split(",")
            noise = len(args) > 1 and args[1].strip().startswith("true")
            fused = "fused" in base
            return ("fused" if fused else "single") + ("" if noise else "_flow_only")
    return None


def measure_pmc(extra_args, launch_len):
    """PMC counters of the integrator launches from rocprofv3, as MI355X_MICROARCH.md (HBM) prescribes: FETCH_SIZE
    and WRITE_SIZE in separate --pmc passes of the same workload (short child runs of this script with launches of
    `launch_len` steps like the timed region), FETCH_SIZE doubled when turned into bytes (gfx950 tallies the 128-B
    requests of a wide coalesced stream at 64 B), both in KiB.  Runs before this process touches the GPU.
    Returns {class: {counter: mean per launch}} and a note."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return {}, "rocprofv3 not found"
    out_all = {}
    notes = []
    for group in PMC_PASSES:
        out = tempfile.mkdtemp(prefix="th_pmc_", dir="/tmp")
        cmd = [prof, "--pmc"] + list(group) + ["--kernel-trace", "--output-format", "csv", "-d", out, "--",
               sys.executable, BENCH, "--pmc-child", str(launch_len), "--no-cpu", "--no-traffic"] + extra_args
        try:
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL, timeout=300, check=True)
            dur = {}
            for f in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    dur[row["Dispatch_Id"]] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
            per_dispatch = {}
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    cls = kernel_class(row.get("Kernel_Name", ""))
                    if cls:
                        per_dispatch.setdefault((cls, row["Dispatch_Id"]), {})[row["Counter_Name"]] = float(row["Counter_Value"])
            for (cls, did), cs in per_dispatch.items():
                d = out_all.setdefault(cls, {})
                for k, v in cs.items():
                    d.setdefault(k, []).append(v)
                if cs.get("GRBM_GUI_ACTIVE", 0) > 0:
                    cyc = cs["GRBM_GUI_ACTIVE"] / 8.0            # summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back)
                    if dur.get(did, 0) > 0:
                        d.setdefault("clock_ghz", []).append(cyc / dur[did])          # cycles per ns
                        d.setdefault("profiled_launch_ms", []).append(dur[did] * 1e-6)
                    if "SQ_INSTS_VALU" in cs:      # issue slots used: 2 cycles per wave64 instruction on each of 1024 SIMDs
                        d.setdefault("valu_issue_utilization", []).append(cs["SQ_INSTS_VALU"] * 2.0 / (cyc * 1024.0))
        except (subprocess.SubprocessError, OSError) as e:
            notes.append("pass %s failed: %s" % ("+".join(group), type(e).__name__))
        finally:
            shutil.rmtree(out, ignore_errors=True)
    res = {cls: {k: sum(v) / len(v) for k, v in cs.items()} for cls, cs in out_all.items()}
    note = "rocprofv3 PMC, child runs with %d-step launches; bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB" % launch_len
    if notes:
        note += "; " + "; ".join(notes)
    return res, note


def pmc_bytes(c):
    if not c or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        return None
    return (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0


