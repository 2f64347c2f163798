#!/usr/bin/env python3
"""Condenses a tools/profile_bench.sh (or profile_gpu.sh) output directory into profiles/<tag>_<name>.{md,json}.

    tools/summarize_profile.py <dir> <tag> <n | name> [cells] [steps_profiled]


 - per-kernel time from rocprofv3 --kernel-trace --stats (kernel_stats.csv)
 - per-launch HBM traffic of each kernel from the two PMC passes: FETCH_SIZE and WRITE_SIZE are reported in KiB
   per dispatch.  MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reads exactly 1/2 of the bytes of wide
   (16 B/lane) streaming reads; other widths are uncalibrated -> we calibrate the read factor on a kernel of known
   traffic in OUR access pattern (8 B/lane): stepper_kernel<3> (cache_previous_tendencies: reads 3 fields, writes 3).
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name[:60]


def load_stats(d):
    f = glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True)
    rows = []
    for r in csv.DictReader(open(f[0])):
        rows.append(dict(name=short(r["Name"]), calls=int(r["Calls"]), total_ms=float(r["TotalDurationNs"]) / 1e6,
                         avg_us=float(r["AverageNs"]) / 1e3, pct=float(r["Percentage"])))
    return rows


MARKER = "profile_marker_kernel"  # ocn_profile_marker: bench.py launches one right before and one right after its timed steps


def load_pmc(d, sub, counter, windowed=False):
    """per kernel: the counter values of its dispatches (windowed: only those between the two marker dispatches, if the run has them)"""
    f = glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True)
    rows = [r for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == counter or MARKER in r["Kernel_Name"]]
    lo = hi = None
    if windowed:
        marks = sorted({int(r["Dispatch_Id"]) for r in rows if MARKER in r["Kernel_Name"]})
        if len(marks) >= 2:
            lo, hi = marks[0], marks[1]
    acc = defaultdict(list)
    for r in rows:
        if MARKER in r["Kernel_Name"] or r["Counter_Name"] != counter:
            continue
        if lo is not None and not (lo < int(r["Dispatch_Id"]) < hi):
            continue
        acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    acc["__windowed__"] = lo is not None
    return acc


def load_trace_window(d):
    """per-kernel (calls, total ms) between the two marker dispatches of the kernel trace, or None"""
    f = glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True)
    if not f:
        return None
    rows = list(csv.DictReader(open(f[0])))
    marks = sorted(int(r["Start_Timestamp"]) for r in rows if MARKER in r["Kernel_Name"])
    if len(marks) < 2:
        return None
    out = defaultdict(lambda: [0, 0.0])
    for r in rows:
        t = int(r["Start_Timestamp"])
        if marks[0] < t < marks[1] and MARKER not in r["Kernel_Name"]:
            e = out[short(r["Kernel_Name"])]
            e[0] += 1
            e[1] += (int(r["End_Timestamp"]) - t) / 1e6
    return {k: tuple(v) for k, v in out.items()}


SQ_COUNTERS = ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_INSTS_SALU",
               "SQ_INSTS_LDS", "SQ_WAIT_INST_ANY")


STALL_COUNTERS = ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_LDS",
                  "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE")
OCC_COUNTERS = ("SQ_WAVES", "SQ_LEVEL_WAVES", "SQ_BUSY_CYCLES", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_ACTIVE_INST_VMEM",
                "SQ_ACTIVE_INST_LDS")


def main():
    d, tag, name = sys.argv[1], sys.argv[2], sys.argv[3]
    n = int(name) if name.isdigit() else None
    stats = load_stats(d)
    fetch = load_pmc(d, "pmc_fetch", "FETCH_SIZE")
    write = load_pmc(d, "pmc_write", "WRITE_SIZE")
    # the same two passes restricted to the timed steps (marker-delimited), for the step totals
    fetch_w = load_pmc(d, "pmc_fetch", "FETCH_SIZE", windowed=True)
    write_w = load_pmc(d, "pmc_write", "WRITE_SIZE", windowed=True)
    windowed = bool(fetch_w.pop("__windowed__")) and bool(write_w.pop("__windowed__"))
    fetch.pop("__windowed__", None); write.pop("__windowed__", None)
    trace_w = load_trace_window(d)
    cells = int(sys.argv[4]) if len(sys.argv) > 4 else n ** 3
    steps_profiled = int(sys.argv[5]) if len(sys.argv) > 5 else None
    sq = {}
    if glob.glob(os.path.join(d, "pmc_sq", "**", "*counter_collection.csv"), recursive=True):
        sq = {c: load_pmc(d, "pmc_sq", c) for c in SQ_COUNTERS}
    stall, occ = {}, {}
    if glob.glob(os.path.join(d, "pmc_stall", "**", "*counter_collection.csv"), recursive=True):
        stall = {c: load_pmc(d, "pmc_stall", c) for c in STALL_COUNTERS}
    if glob.glob(os.path.join(d, "pmc_occ", "**", "*counter_collection.csv"), recursive=True):
        occ = {c: load_pmc(d, "pmc_occ", c) for c in OCC_COUNTERS}
    field_bytes = cells * 8.0
    # calibration kernel of known traffic in our 8 B/lane pattern:
    #   stepper_kernel<3> (cache_previous_tendencies): reads 3 fields, writes 3;  or, when the host swaps the G buffers
    #   and that kernel never runs, pressure_correct_kernel: reads p, u, v, w (4 fields), writes u, v, w (3).
    med = lambda v: sorted(v)[len(v) // 2]
    read_factor = write_factor = None
    #   hydrostatic workloads: hydrostatic_pressure_kernel reads T, S and writes pHY' (columns 0 .. N+1: +0.4 % at 1024^2, neglected)
    for cal, nr, nw in (("ocn::stepper_kernel<3>", 3, 3), ("ocn::pressure_correct_kernel", 4, 3), ("ocn::hydrostatic_pressure_kernel", 2, 1)):
        if cal in fetch and cal in write:
            read_factor = nr * field_bytes / (med(fetch[cal]) * 1024)
            write_factor = nw * field_bytes / (med(write[cal]) * 1024)
            break
    command = open(os.path.join(d, "command.txt")).read().strip() if os.path.exists(os.path.join(d, "command.txt")) else None
    out = {"tag": tag, "n": n, "name": name, "cells": cells, "command": command, "steps_profiled": steps_profiled,
           "read_calibration_factor": read_factor, "write_calibration_factor": write_factor, "kernels": []}
    # FETCH_SIZE undercounts by the access width (MI355X_MICROARCH.md, HBM section): exactly x2 for 16 B / lane streaming reads, and the
    # factor calibrated above for this code's 8 B / lane reads.  Kernels whose loads are complex numbers (16 B / lane) take the guide's 2.
    WIDE = ("colfft_kernel", "colfft_io_kernel", "rowfft_c2r_kernel", "realfft_y_inv_kernel", "xtri_forward_kernel", "xtri_backward_kernel",
            "tridiag_z_kernel")
    rf = lambda name: 2.0 if any(w in name for w in WIDE) else read_factor
    out["read_factor_wide_loads"] = 2.0
    step_bytes = 0.0
    if windowed and read_factor:
        # exact: every dispatch between the markers, each with its own counter value (no medians, no set! / warm-up launches)
        step_bytes = sum(sum(v) * rf(k) for k, v in fetch_w.items()) * 1024 + sum(sum(v) for v in write_w.values()) * 1024 * write_factor
        out["window"] = "timed steps only: dispatches between the two ocn_profile_marker launches of bench.py"
        if trace_w:
            out["window_kernels"] = {k: {"calls": c, "total_ms": t} for k, (c, t) in sorted(trace_w.items(), key=lambda kv: -kv[1][1])}
            out["window_gpu_ms"] = sum(t for _, t in trace_w.values())
    for s in stats:
        k = dict(s)
        if s["name"] in fetch and s["name"] in write and read_factor:
            rd = med(fetch[s["name"]]) * 1024 * rf(s["name"])
            wr = med(write[s["name"]]) * 1024 * write_factor
            k.update(read_bytes=rd, write_bytes=wr, traffic_bytes=rd + wr, traffic_bytes_per_cell=(rd + wr) / cells,
                     hbm_GBps=(rd + wr) / (s["avg_us"] * 1e-6) / 1e9,
                     raw_fetch_KiB=med(fetch[s["name"]]), raw_write_KiB=med(write[s["name"]]))
            if not windowed:
                step_bytes += (rd + wr) * s["calls"]
            else:
                k["window_calls"] = len(fetch_w.get(s["name"], []))
                k["window_traffic_bytes"] = sum(fetch_w.get(s["name"], [])) * 1024 * rf(s["name"]) + sum(write_w.get(s["name"], [])) * 1024 * write_factor
        for c in SQ_COUNTERS:
            if c in sq and s["name"] in sq[c]:
                k[c] = med(sq[c][s["name"]])
        for group, prefix in ((stall, "stall_"), (occ, "occ_")):
            for c, per_kernel in group.items():
                if s["name"] in per_kernel:
                    k[prefix + c] = med(per_kernel[s["name"]])
        wc = k.get("stall_SQ_WAVE_CYCLES")
        if wc:
            # fractions of the wave cycles: issuing, stalled at issue (of which LDS), parked at s_waitcnt / barrier
            k["frac_active"] = k.get("stall_SQ_ACTIVE_INST_ANY", 0.0) / wc
            k["frac_issue_stall"] = k.get("stall_SQ_WAIT_INST_ANY", 0.0) / wc
            k["frac_issue_stall_lds"] = k.get("stall_SQ_WAIT_INST_LDS", 0.0) / wc
            k["frac_parked"] = k.get("stall_SQ_WAIT_ANY", 0.0) / wc
            if k.get("stall_SQ_LDS_IDX_ACTIVE"):
                k["lds_conflict_frac"] = k.get("stall_SQ_LDS_BANK_CONFLICT", 0.0) / k["stall_SQ_LDS_IDX_ACTIVE"]
        if k.get("occ_SQ_LEVEL_WAVES") and k.get("occ_SQ_BUSY_CYCLES"):
            k["mean_waves_in_flight_per_SE_unit"] = k["occ_SQ_LEVEL_WAVES"] / k["occ_SQ_BUSY_CYCLES"]
        if "SQ_ACTIVE_INST_VALU" in k and k.get("GRBM_GUI_ACTIVE"):
            # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the 1024 SIMDs, GRBM_GUI_ACTIVE cycles summed over the 8 XCDs
            k["valu_busy"] = k["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * k["GRBM_GUI_ACTIVE"] / 8)
            k["clock_GHz_under_pmc"] = k["GRBM_GUI_ACTIVE"] / 8 / (s["avg_us"] * 1e3)
            k["valu_wave_instr_per_cell"] = k["SQ_INSTS_VALU"] / cells
        out["kernels"].append(k)
    if steps_profiled:
        # all launches of the profiled command (steps + warm-up + set!) over the steps it ran: an upper bound of one step's traffic
        out["step_bytes_per_cell"] = step_bytes / steps_profiled / cells
    os.makedirs("profiles", exist_ok=True)
    base = os.path.join("profiles", f"{tag}_{name}")
    json.dump(out, open(base + ".json", "w"), indent=1)
    with open(base + ".md", "w") as f:
        f.write(f"# rocprofv3 summary {tag}: `{command or 'bench.py --size ' + str(n)}`\n\n")
        f.write("Source: `rocprofv3 --kernel-trace --stats` and separate `--pmc` passes (FETCH_SIZE; WRITE_SIZE; SQ counters) of the same "
                "command; see tools/profile_bench.sh / tools/summarize_profile.py.  valu busy = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x "
                "GRBM_GUI_ACTIVE / 8 XCDs); VALU instr/cell = SQ_INSTS_VALU (wave instructions) / cells.\n\n")
        if "step_bytes_per_cell" in out:
            if windowed:
                f.write(f"Measured HBM traffic of the TIMED steps (every dispatch between bench.py's two marker launches) / ({steps_profiled} steps x "
                        f"{cells} cells) = **{out['step_bytes_per_cell']:.0f} B per cell per step**.\n\n")
                if "window_kernels" in out:
                    f.write("Kernels of the timed steps (kernel trace, same window): " + ", ".join(
                        f"`{k.split('::')[-1][:48]}` {v['calls']} x {v['total_ms'] / max(v['calls'], 1) * 1e3:.0f} us" for k, v in list(out["window_kernels"].items())[:12])
                        + f"; GPU time {out['window_gpu_ms'] / steps_profiled:.2f} ms per step.\n\n")
            else:
                f.write(f"Measured HBM traffic of the whole command / ({steps_profiled} steps x {cells} cells) = "
                        f"**{out['step_bytes_per_cell']:.0f} B per cell per step** (includes set! and warm-up launches).\n\n")
        f.write(f"PMC calibration on `{cal}` (known fields read / written): read x{read_factor}, write x{write_factor}; kernels that load complex numbers "
                f"(16 B per lane: column FFTs, c2r rows, x solve, z tridiagonal) take the guide's exact x2 for FETCH_SIZE.\n\n")
        f.write("| kernel | calls | avg us | % GPU time | HBM traffic/launch (B/cell) | HBM GB/s | VALU wave-instr/cell | valu busy | GHz under PMC |\n|---|---|---|---|---|---|---|---|---|\n")
        for k in out["kernels"][:25]:
            t = f"{k['traffic_bytes_per_cell']:.1f}" if "traffic_bytes" in k else "-"
            b = f"{k['hbm_GBps']:.0f}" if "hbm_GBps" in k else "-"
            vi = f"{k['valu_wave_instr_per_cell']:.2f}" if "valu_wave_instr_per_cell" in k else "-"
            vb = f"{k['valu_busy']:.2f}" if "valu_busy" in k else "-"
            gh = f"{k['clock_GHz_under_pmc']:.2f}" if "clock_GHz_under_pmc" in k else "-"
            f.write(f"| `{k['name']}` | {k['calls']} | {k['avg_us']:.1f} | {k['pct']:.2f} | {t} | {b} | {vi} | {vb} | {gh} |\n")
        if any("frac_parked" in k for k in out["kernels"]):
            f.write("\n## Where the wave cycles go (SQ_WAVE_CYCLES = issuing + stalled at issue + parked at s_waitcnt / barrier)\n\n"
                    "| kernel | issuing | issue stall (LDS part) | parked (s_waitcnt, barrier) | LDS bank-conflict cycles / LDS cycles | LDS instr / VMEM rd instr per launch |\n|---|---|---|---|---|---|\n")
            for k in out["kernels"][:14]:
                if "frac_parked" not in k:
                    continue
                lc = f"{k['lds_conflict_frac']:.3f}" if "lds_conflict_frac" in k else "-"
                f.write(f"| `{k['name']}` | {k['frac_active']:.2f} | {k['frac_issue_stall']:.2f} ({k['frac_issue_stall_lds']:.2f}) | {k['frac_parked']:.2f} | {lc} | "
                        f"{k.get('stall_SQ_INSTS_LDS', 0):.3g} / {k.get('occ_SQ_INSTS_VMEM_RD', 0):.3g} |\n")
    print(open(base + ".md").read())


if __name__ == "__main__":
    main()
