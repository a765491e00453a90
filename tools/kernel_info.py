#!/usr/bin/env python3
"""Resource usage of every kernel in a hipcc -S listing (tools/disasm.sh): registers, spills, LDS, occupancy, instruction mix.
usage: tools/kernel_info.py /tmp/rt_disasm/head.s [substring]"""
import re, subprocess, sys
path = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
name = None; body = []; rows = []
def demangle(n):
    try: return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    except Exception: return n
info = {}
for line in open(path):
    m = re.match(r"^(_Z\w+):\s*; @", line)
    if m: name = m.group(1); body = []; info = {}; continue
    if name is None: continue
    s = line.strip()
    m = re.match(r"; (NumVgprs|TotalNumSgprs|ScratchSize|Occupancy|LDSByteSize|codeLenInByte) *[:=] *(\d+)", s)
    if m: info[m.group(1)] = int(m.group(2))
    m = re.match(r"; (sgpr_spill_count|vgpr_spill_count): *(\d+)", s)
    if m: info[m.group(1)] = int(m.group(2))
    if s and not s.startswith((";", ".")) and not s.endswith(":"): body.append(s.split()[0])
    if s.startswith(".end_amdhsa_kernel") or (s.startswith("; Occupancy")):
        pass
    if "Occupancy" in info and name:
        from collections import Counter
        c = Counter()
        for op in body:
            k = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") and not op.startswith(("s_load", "s_buffer", "s_waitcnt", "s_cbranch", "s_branch")) else \
                "smem" if op.startswith(("s_load", "s_buffer")) else "branch" if op.startswith(("s_cbranch", "s_branch")) else "wait" if op.startswith("s_waitcnt") else \
                "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "flat_", "buffer_", "scratch_")) else "other"
            c[k] += 1
        rows.append((name, dict(info), c)); name = None
for n, i, c in rows:
    d = demangle(n)
    if sub and sub not in d: continue
    print(f"{d[:150]}\n    vgpr {i.get('NumVgprs')} sgpr {i.get('TotalNumSgprs')} scratch {i.get('ScratchSize')} B lds {i.get('LDSByteSize')} occupancy {i.get('Occupancy')} code {i.get('codeLenInByte')} B | " + " ".join(f"{k} {v}" for k, v in sorted(c.items())))
