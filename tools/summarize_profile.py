#!/usr/bin/env python3
"""Condenses a tools/profile_gpu.sh output directory into profiles/<tag>_<n>.{md,json}.

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


def load_pmc(d, sub, counter):
    f = glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == counter:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc


def main():
    d, tag, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
    stats = load_stats(d)
    fetch = load_pmc(d, "pmc_fetch", "FETCH_SIZE")
    write = load_pmc(d, "pmc_write", "WRITE_SIZE")
    cells = n ** 3
    field_bytes = cells * 8.0
    # calibration kernel of known traffic in our 8 B/lane pattern:
    #   stepper_kernel<3> (cache_previous_tendencies): reads 3 fields, writes 3;  or, when the host swaps the G buffers
    #   and that kernel never runs, pressure_correct_kernel: reads p, u, v, w (4 fields), writes u, v, w (3).
    med = lambda v: sorted(v)[len(v) // 2]
    read_factor = write_factor = None
    for cal, nr, nw in (("ocn::stepper_kernel<3>", 3, 3), ("ocn::pressure_correct_kernel", 4, 3)):
        if cal in fetch and cal in write:
            read_factor = nr * field_bytes / (med(fetch[cal]) * 1024)
            write_factor = nw * field_bytes / (med(write[cal]) * 1024)
            break
    out = {"tag": tag, "n": n, "read_calibration_factor": read_factor, "write_calibration_factor": write_factor, "kernels": []}
    for s in stats:
        k = dict(s)
        if s["name"] in fetch and s["name"] in write and read_factor:
            rd = med(fetch[s["name"]]) * 1024 * read_factor
            wr = med(write[s["name"]]) * 1024 * write_factor
            k.update(read_bytes=rd, write_bytes=wr, traffic_bytes=rd + wr, traffic_bytes_per_cell=(rd + wr) / cells,
                     hbm_GBps=(rd + wr) / (s["avg_us"] * 1e-6) / 1e9,
                     raw_fetch_KiB=med(fetch[s["name"]]), raw_write_KiB=med(write[s["name"]]))
        out["kernels"].append(k)
    os.makedirs("profiles", exist_ok=True)
    base = os.path.join("profiles", f"{tag}_{n}")
    json.dump(out, open(base + ".json", "w"), indent=1)
    with open(base + ".md", "w") as f:
        f.write(f"# rocprofv3 summary {tag}, bench.py --size {n}\n\n")
        f.write("Source: `rocprofv3 --kernel-trace --stats` and two separate `--pmc` passes (FETCH_SIZE, WRITE_SIZE); "
                "see tools/profile_gpu.sh / tools/summarize_profile.py.\n\n")
        f.write(f"PMC calibration on `{cal}` (known fields read / written): read x{read_factor}, write x{write_factor}\n\n")
        f.write("| kernel | calls | avg us | % GPU time | HBM traffic/launch (B/cell) | HBM GB/s |\n|---|---|---|---|---|---|\n")
        for k in out["kernels"][:25]:
            t = f"{k['traffic_bytes_per_cell']:.1f}" if "traffic_bytes" in k else "-"
            b = f"{k['hbm_GBps']:.0f}" if "hbm_GBps" in k else "-"
            f.write(f"| `{k['name']}` | {k['calls']} | {k['avg_us']:.1f} | {k['pct']:.2f} | {t} | {b} |\n")
    print(open(base + ".md").read())


if __name__ == "__main__":
    main()
