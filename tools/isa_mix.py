#!/usr/bin/env python3
"""Instruction mix of the main loop of a kernel in a hipcc -S listing: finds the kernel's largest backward branch (the plane loop) and
counts the instructions between its target label and the branch by class.
  hipcc --offload-arch=gfx950 -O3 ... --cuda-device-only -S file.hip -o file.s;  tools/isa_mix.py file.s <mangled-name-prefix>"""
import collections
import re
import sys


def main():
    path, prefix = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith(prefix) and l.rstrip().split(":")[0].startswith(prefix) and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end]
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    best = None
    for i, l in enumerate(body):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"\s+s_branch\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            span = i - labels[m.group(1)]
            if best is None or span > best[0]:
                best = (span, labels[m.group(1)], i)
    _, a, b = best
    cnt = collections.Counter()
    for l in body[a:b + 1]:
        t = l.strip()
        if not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        op = t.split()[0]
        if op.startswith("v_") and "f64" in op:
            cls = "valu_f64:" + re.sub(r"_e\d+$", "", op)
        elif op.startswith("v_cndmask"):
            cls = "valu_cndmask"
        elif op.startswith("v_"):
            cls = "valu_other"
        elif op.startswith("ds_"):
            cls = "lds"
        elif op.startswith(("global_", "flat_", "buffer_", "scratch_")):
            cls = "vmem"
        elif op.startswith("s_waitcnt"):
            cls = "s_waitcnt"
        elif op.startswith("s_barrier"):
            cls = "s_barrier"
        elif op.startswith("s_"):
            cls = "salu"
        else:
            cls = "other:" + op
        cnt[cls] += 1
    tot = sum(cnt.values())
    f64 = sum(v for k, v in cnt.items() if k.startswith("valu_f64"))
    valu = f64 + cnt["valu_cndmask"] + cnt["valu_other"]
    print(f"kernel {prefix[:70]}...: main loop {b - a + 1} lines, {tot} instructions; VALU {valu} (fp64 {f64}, cndmask {cnt['valu_cndmask']}, other {cnt['valu_other']}), "
          f"LDS {cnt['lds']}, VMEM {cnt['vmem']}, SALU {cnt['salu']}, waitcnt {cnt['s_waitcnt']}, barriers {cnt['s_barrier']}")
    for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]):
        if k.startswith("valu_f64"):
            print(f"    {k:40s} {v}")


if __name__ == "__main__":
    main()
