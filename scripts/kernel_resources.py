"""Per-kernel register / scratch / LDS / occupancy table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re
import subprocess
import sys

src = sys.argv[1]
flags = "-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-gpu-rdc -mno-unsafe-fp-atomics --offload-arch=gfx950".split()
out = subprocess.run(["/opt/rocm/bin/hipcc", *flags, "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                     capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
print(f"{'kernel':60s} {'VGPR':>5s} {'AGPR':>5s} {'scratch':>8s} {'LDS':>7s} {'occ':>4s} {'vspill':>6s}")
for k, v in rows.items():
    print(f"{k[:60]:60s} {v.get('VGPRs', 0):5d} {v.get('AGPRs', 0):5d} {v.get('ScratchSize', 0):8d} {v.get('LDS Size', 0):7d} "
          f"{v.get('Occupancy', 0):4d} {v.get('VGPRs Spill', 0):6d}")
