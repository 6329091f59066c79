"""Per-kernel register / spill / occupancy / LDS table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kres.py dlmc-quant_amd/csrc/conv_i8.hip [-DDLMCQ_LAB]"""
import re
import subprocess
import sys

src = sys.argv[1]
flags = ("-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt "
         "-fno-gpu-flush-denormals-to-zero -I/root/repo/include -Rpass-analysis=kernel-resource-usage").split()
out = subprocess.run(["/opt/rocm/bin/hipcc", *flags, *sys.argv[2:], "-c", src, "-o", "/dev/null"], capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s+\[-Rpass", line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
for name, r in rows.items():
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(.*", "", dem).replace("dlmcq::", "").replace("void ", "")
    print(f"{dem:70s} vgpr {r.get('VGPRs','?'):>4} agpr {r.get('AGPRs','?'):>4} spill {r.get('VGPRs Spill','?'):>3} occ {r.get('Occupancy [waves/SIMD]','?')} lds {r.get('LDS Size [bytes/block]','?')}")
