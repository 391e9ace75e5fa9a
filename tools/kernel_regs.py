"""Registers, scratch and occupancy per kernel of one HIP source, from the compiler's
resource-usage remarks:  python tools/kernel_regs.py dsp_amd/csrc/fft_stockham.hip [filter] [flags...]"""
import re
import subprocess
import sys

src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
flags = sys.argv[3:]
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"] + flags
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in err.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    for key, rx in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                    ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(rx, line)
        if m and cur:
            rows[cur][key] = int(m.group(1))
for name, r in rows.items():
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(anonymous namespace\)::", "", dem).split("(")[0]
    if pat in dem:
        print("%-70s vgpr %3d agpr %3d scratch %4d occ %d" % (dem[:70], r.get("vgpr", -1), r.get("agpr", 0), r.get("scratch", 0), r.get("occ", 0)))
