"""Per-kernel resources of gfx950 assembly listings (hipcc -S --cuda-device-only): SGPRs, VGPRs, scratch bytes, occupancy, and the static
instruction mix -- all VALU, fp64 VALU, LDS, v_readlane / v_writelane (SGPR spills into VGPR lanes), scratch accesses.  Line-based (the
listings are tens of MB).  Usage: python tools/kernel_resources.py file.s [...] [--filter substring] [--loop]
--loop additionally counts the instructions of the blocks marked as belonging to a loop (the marching loop of the z-marching kernels)."""
import re
import subprocess
import sys


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return [o.split("(")[0] for o in out]


def parse(path):
    kernels, cur = [], None
    for line in open(path, errors="replace"):
        if cur is None:
            m = re.match(r"^(_Z\w+):", line)
            if m:
                cur = {"name": m.group(1), "ins": [], "loop": [], "inloop": False, "meta": {}}
            continue
        if line.startswith(".Lfunc_end"):
            cur["done"] = True
            continue
        if cur.get("done"):
            m = re.match(r"^; (NumSgprs|NumVgprs|ScratchSize|Occupancy|LDSByteSize): (\d+)", line)
            if m:
                cur["meta"][m.group(1)] = int(m.group(2))
            if line.startswith("; Occupancy") or line.startswith("\t.text") or line.startswith("\t.section"):
                if "Occupancy" in cur["meta"]:
                    kernels.append(cur)
                    cur = None
            continue
        if re.match(r"^\.LBB\d+_\d+:", line):
            cur["inloop"] = "Loop" in line
            continue
        if line.startswith("\t") and not line.strip().startswith((";", ".")):
            op = line.split()[0]
            cur["ins"].append(op)
            if cur["inloop"]:
                cur["loop"].append(op)
    return kernels


def mix(ins):
    return {"valu": sum(1 for i in ins if i.startswith("v_")), "f64": sum(1 for i in ins if i.startswith("v_") and "f64" in i),
            "lds": sum(1 for i in ins if i.startswith("ds_")), "readlane": sum(1 for i in ins if i.startswith("v_readlane")),
            "writelane": sum(1 for i in ins if i.startswith("v_writelane")), "scratch": sum(1 for i in ins if i.startswith("scratch_")),
            "vmem": sum(1 for i in ins if i.startswith(("global_", "buffer_", "flat_")))}


def main():
    args = sys.argv[1:]
    flt, loop, files = None, False, []
    while args:
        a = args.pop(0)
        if a == "--filter":
            flt = args.pop(0)
        elif a == "--loop":
            loop = True
        else:
            files.append(a)
    for f in files:
        ks = parse(f)
        names = demangle([k["name"] for k in ks])
        print("==", f)
        for k, n in zip(ks, names):
            if flt and flt not in n:
                continue
            m, meta = mix(k["ins"]), k["meta"]
            line = (f"  {n[:96]:98s} sgpr {meta.get('NumSgprs', -1):3d} vgpr {meta.get('NumVgprs', -1):3d} scratch {meta.get('ScratchSize', -1):4d} B "
                    f"occ {meta.get('Occupancy', -1)} | valu {m['valu']:4d} f64 {m['f64']:4d} lds {m['lds']:3d} vmem {m['vmem']:3d} "
                    f"readlane {m['readlane']:3d} writelane {m['writelane']:3d} scratch {m['scratch']:3d}")
            if loop and k["loop"]:
                l = mix(k["loop"])
                line += f" || loop: valu {l['valu']} f64 {l['f64']} lds {l['lds']} vmem {l['vmem']} readlane {l['readlane']} scratch {l['scratch']}"
            print(line)


if __name__ == "__main__":
    main()
