"""Per-kernel register / scratch / occupancy summary of one HIP source, from the compiler's resource remarks.
Usage: python scripts/kernel_resources.py warp_lds.hip [-DFOO=1 ...]"""
import os, re, subprocess, sys
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cds_mvsnet_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", *sys.argv[2:],
       "-c", sys.argv[1], "-o", "/tmp/_kr.o", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, cwd=csrc, capture_output=True, text=True).stderr
cur, vals = None, {}
for ln in out.splitlines():
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(anonymous namespace\)::", "", cur).split("(")[0].replace("void ", "")
        vals = {}
        continue
    m = re.search(r"remark:\s*([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", ln)
    if m and cur:
        vals[m.group(1).strip()] = int(m.group(2))
        if m.group(1).startswith("LDS"):
            print(f"{cur:64s} vgpr {vals.get('VGPRs', 0):3d} agpr {vals.get('AGPRs', 0):3d} scratch {vals.get('ScratchSize', 0):4d}"
                  f" occ {vals.get('Occupancy', 0)} vspill {vals.get('VGPRs Spill', 0):3d} sspill {vals.get('SGPRs Spill', 0):3d}")
