"""Per-kernel instruction counts of a gfx950 assembly listing (hipcc -S --cuda-device-only): fp64 VALU, all VALU, LDS, scratch (spill) accesses of
every tiled kernel of the fast-math namespace.  Usage: python tools/isa_report.py file.s [...]"""
import re, sys, subprocess
def report(sfile):
    txt = open(sfile).read()
    out = []
    for m in re.finditer(r"^(_ZN8ocn_fast\w+):.*?\n(.*?)^\.Lfunc_end", txt, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if "tiled" not in name: continue
        ins = [l.split()[0] for l in body.splitlines() if l.startswith("\t") and not l.strip().startswith(";") and not l.strip().startswith(".")]
        f64 = sum(1 for i in ins if i.startswith("v_") and "f64" in i)
        valu = sum(1 for i in ins if i.startswith("v_"))
        scr = sum(1 for i in ins if i.startswith("scratch_"))
        lds = sum(1 for i in ins if i.startswith("ds_"))
        short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.split("(")[0]
        out.append((short, f64, valu, lds, scr))
    return out
for f in sys.argv[1:]:
    print("==", f)
    for r in report(f): print("  %-78s f64 %4d valu %4d lds %3d scratch %3d" % r)
