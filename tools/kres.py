#!/usr/bin/env python3
"""Register / scratch / LDS use of the kernels whose (demangled) name contains PATTERN:
   tools/kres.py PATTERN [file.hip]   (compiles the file for gfx950 with -Rpass-analysis=kernel-resource-usage)"""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = sys.argv[1]
src = sys.argv[2] if len(sys.argv) > 2 else "mgps_kernels.hip"
cs = os.path.join(root, "geometricmultigridpressuresolver_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{root}/include", "-I.", "-I/opt/rocm/include", "-c", src,
       "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, cwd=cs, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z][A-Za-z ]*?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
names = list(rows)
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
for n, d in zip(names, dem):
    if pat in d:
        r = rows[n]
        short = re.sub(r">\(.*", ">", d.replace("mgps::(anonymous namespace)::", "").replace("void ", ""))
        print(f"{short:70s} vgpr {r.get('VGPRs', -1):3d} agpr {r.get('AGPRs', -1):3d} sgpr {r.get('TotalSGPRs', -1):3d} scratch {r.get('ScratchSize', -1):4d} lds {r.get('LDS Size', -1):6d} occ {r.get('Occupancy', -1)}")
