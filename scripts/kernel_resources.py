"""Per-kernel register / LDS / occupancy table from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
usage: hipcc ... -Rpass-analysis=kernel-resource-usage ... 2> res.txt ; python scripts/kernel_resources.py res.txt [filter]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
blocks = re.split(r"remark: .*?Function Name: ", txt)[1:]
names = [b.split("\n")[0].strip() for b in blocks]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
for b, dn in zip(blocks, dem):
    if flt and flt not in dn:
        continue
    g = lambda k: (re.search(k + r": (\d+)", b) or [0, 0])[1]
    dn = re.sub(r"ilqr::", "", dn)
    print(dn[:120].ljust(120), "VGPR", g("VGPRs"), "AGPR", g("AGPRs"), "spill", g("VGPR Spill"), "scratch",
          g(r"ScratchSize \[bytes/lane\]"), "occ", g(r"Occupancy \[waves/SIMD\]"), "LDS", g(r"LDS Size \[bytes/block\]"))
