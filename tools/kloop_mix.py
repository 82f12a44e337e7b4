"""Instruction mix of the K loops of the MFMA kernels (static, from the gfx950 ISA hipcc emits): for every kernel whose name
matches, the innermost loops that contain MFMAs -- scalar / vector / LDS / memory / MFMA instruction counts of one trip and the
scalar + vector instructions per MFMA.  A wave issues one instruction at a time: with two waves per SIMD a trip whose non-MFMA
instructions need more issue cycles than its MFMAs need pipe cycles (16 each) cannot keep the matrix pipe busy.
  python tools/kloop_mix.py csrc-file.hip 'kernel-regex' [extra hipcc flags]   (counts are of the loop TEXT: both sides of a
  branch inside the loop are counted)"""
import collections
import os
import re
import subprocess
import sys

src, pat = sys.argv[1], re.compile(sys.argv[2])
out = "/tmp/kloop_mix.s"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-result", "-Wno-return-type", "-S",
                "--cuda-device-only", os.path.abspath(src), "-o", out] + sys.argv[3:], check=True, stderr=subprocess.DEVNULL, cwd=os.path.dirname(os.path.abspath(src)))
text = open(out).read()
kind = lambda x: ("mfma" if "mfma" in x else "valu" if x.startswith("v_") else "salu" if x.startswith("s_") else "lds" if x.startswith("ds_")
                  else "vmem" if x.startswith(("global", "scratch", "buffer")) else "other")
for f in re.split(r"\n\t\.globl\t", text):
    name = f.split("\n", 1)[0].split()[0]
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if not pat.search(dem):
        continue
    lines = [l.split(";")[0].rstrip() for l in f.split("\n")]
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"(\.LBB\d+_\d+):", l)] if m}
    loops = []
    for i, l in enumerate(lines):
        m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    inner = [lp for lp in loops if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] and (o[1] - o[0]) < (lp[1] - lp[0]) and
                                             any("mfma" in x for x in lines[o[0]:o[1]]) for o in loops)]
    rows = []
    for a, b in sorted(set(inner)):
        seg = [x.strip() for x in lines[a:b + 1] if x.strip() and not x.strip().startswith(".") and not x.strip().endswith(":")]
        c = collections.Counter(kind(x) for x in seg)
        if c["mfma"] >= 8:
            rows.append(c)
    print(dem.replace("(anonymous namespace)::", "")[:120])
    for c in rows:
        print(f"    trip: {c['mfma']:5d} MFMA {c['salu']:5d} scalar {c['valu']:5d} vector {c['lds']:4d} LDS {c['vmem']:4d} memory   "
              f"-> {(c['salu'] + c['valu']) / c['mfma']:5.2f} scalar+vector per MFMA")
