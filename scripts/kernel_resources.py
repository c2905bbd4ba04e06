"""Compiler view of the kernels' resources: parses a `hipcc -Rpass-analysis=kernel-resource-usage` log.
   usage: kernel_resources.py <log> [name filter]"""
import re, subprocess, sys
log = open(sys.argv[1]).read()
filt = sys.argv[2] if len(sys.argv) > 2 else ""
cur = None
rows = {}
for l in log.split("\n"):
    m = re.search(r"remark: (?:Function )?Name: (\S+)", l)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+(?:\[[^\]]*\])?): (\d+)", l)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for n, r in rows.items():
    dn = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r"\(.*", "", dn).replace("dsopp_hip::", "").replace("void ", "")
    if filt not in dn:
        continue
    print(f"{dn[:70]:70s} V {r.get('VGPRs', 0):4d} A {r.get('AGPRs', 0):3d} spill {r.get('VGPRs Spill', 0):3d} sgpr-spill {r.get('SGPRs Spill', 0):3d} "
          f"scratch {r.get('ScratchSize [bytes/lane]', 0):4d} occ {r.get('Occupancy [waves/SIMD]', 0)} lds {r.get('LDS Size [bytes/block]', 0)}")
