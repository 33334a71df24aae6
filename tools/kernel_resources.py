"""Development tool: registers / spills / scratch of every kernel of one source file.

    python tools/kernel_resources.py gnn_manip_amd/csrc/hmlp.hip [extra hipcc flags]
"""
import re
import subprocess
import sys

src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-fno-slp-vectorize", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"gm::\(anonymous namespace\)::", "", cur).split("(")[0]
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(\w[\w ]*?)(?: \[bytes/lane\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
    if "error" in line:
        print(line)
print(f"{'kernel':60s} VGPR AGPR SGPR spill scratch")
for k, r in rows.items():
    print(f"{k[:60]:60s} {r.get('VGPRs', 0):4d} {r.get('AGPRs', 0):4d} {r.get('TotalSGPRs', 0):4d} {r.get('VGPRs Spill', 0):5d} {r.get('ScratchSize', 0):7d}")
