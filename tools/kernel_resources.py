"""The compiler's resource table (VGPRs, AGPRs, scratch, occupancy) per kernel, from a `-Rpass-analysis=kernel-resource-usage` build log
or by compiling the library into a temporary directory.  usage: python tools/kernel_resources.py [name-substring] [--log FILE]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = next((a for a in sys.argv[1:] if not a.startswith("--")), "")
log = None
if "--log" in sys.argv:
    log = open(sys.argv[sys.argv.index("--log") + 1]).read()
else:
    with tempfile.TemporaryDirectory() as d:
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=on",
                            "-Rpass-analysis=kernel-resource-usage", os.path.join(ROOT, "quadruped_control_amd/csrc/qc_balance.hip"), "-o", os.path.join(d, "lib.so")]
                           + [a for a in sys.argv[1:] if a.startswith("-D")], capture_output=True, text=True)
        log = r.stderr
        if r.returncode:
            sys.exit(log[-3000:])
cur = None
rows = {}
for ln in log.splitlines():
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        cur = m.group(1); rows[cur] = {}
        continue
    m = re.search(r":\d+:\s+(?:remark:\s+)?(VGPRs Spill|SGPRs Spill|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", ln)
    if m and cur:
        key = m.group(1)
        rows[cur][key if key.endswith("Spill") else key.split(" ")[0]] = int(m.group(2))
for k, v in rows.items():
    if pat in k:
        short = re.sub(r"^_ZN2qc\d+", "", k)[:90]
        print(f"{short:92s} VGPR {v.get('VGPRs', 0):3d} AGPR {v.get('AGPRs', 0):3d} scratch {v.get('ScratchSize', 0):4d} B occ {v.get('Occupancy', 0)} sgpr-spill {v.get('SGPRs Spill', 0)} vgpr-spill {v.get('VGPRs Spill', 0)}")
