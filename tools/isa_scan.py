#!/usr/bin/env python
"""Static look at the MFMA loops of a kernel file's ISA: per kernel, how many v_mfma are directly behind an `s_waitcnt lgkmcnt(0)`
that follows an LDS read (the compiler's read - wait - MFMA serialisation: an exposed LDS latency per MFMA), by the source line of
that read.  usage: python tools/isa_scan.py rgb-no-more_amd/csrc/<file>.hip [-DFLAG ...]   (runs hipcc -S -gline-tables-only)"""
import re
import subprocess
import sys
from collections import Counter

src = sys.argv[1]
out = "/tmp/isa_scan.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-gline-tables-only", "-S",
                "--cuda-device-only", src, "-o", out] + sys.argv[2:], check=True, stderr=subprocess.DEVNULL)
s = open(out).read()
files = dict(re.findall(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', s))
for name in re.findall(r"\.amdhsa_kernel (\S+)", s):
    k = s.find("\n" + name + ":")
    j = s.find(".Lfunc_end", k)
    cur, lines = None, []
    for l in s[k:j].split("\n"):
        t = l.strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
        if m:
            cur = (files.get(m.group(1), m.group(1)).split("/")[-1], int(m.group(2)))
            continue
        t = t.split(";")[0].strip()
        if t and not t.startswith("."):
            lines.append((t, cur))
    n = sum(1 for l, _ in lines if l.startswith("v_mfma"))
    if not n:
        continue
    ser = Counter()
    for q, (l, _) in enumerate(lines):
        if l.startswith("v_mfma"):
            back = lines[max(0, q - 3):q]
            if any(b[0].startswith("s_waitcnt") and "lgkmcnt(0)" in b[0] for b in back):
                rd = [b for b in back if b[0].startswith("ds_read")]
                if rd:
                    ser[rd[-1][1]] += 1
    regs = re.search(re.escape(name) + r"[\s\S]*?; NumVgprs: (\d+)[\s\S]*?; ScratchSize: (\d+)", s[j:])
    print(f"{name[:72]:72s} mfma {n:4d}  serialized {sum(ser.values()):4d}  vgprs/scratch {regs.group(1) if regs else '?'}/{regs.group(2) if regs else '?'}  {dict(ser) if ser else ''}")
