"""Register / spill table of one HIP source: python tools/kernel_regs.py mfcnet-tracker_amd/csrc/conv3x3_ring.hip [extra hipcc flags]"""
import re, subprocess, sys, os
src = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(root, "include"),
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + sys.argv[2:]
out = subprocess.run(cmd, stderr=subprocess.PIPE, text=True).stderr
cur = None
rows = []
for ln in out.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.*) \[-Rpass", ln) or re.search(r":\d+:\d+: remark:\s+(.*) \[-Rpass", ln)
    body = m.group(1) if m else ln
    m2 = re.search(r"Function Name: (\S+)", body) or re.search(r"Name: (\S+)", body)
    if m2:
        cur = {"name": m2.group(1)}; rows.append(cur); continue
    for key in ("TotalSGPRs", "VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs Spill", "VGPRs Spill", "LDS Size [bytes/block]"):
        m3 = re.search(re.escape(key) + r": (\d+)", body)
        if m3 and cur is not None and key not in cur:
            cur[key] = int(m3.group(1))
dem = subprocess.run(["c++filt"] + [r["name"] for r in rows], stdout=subprocess.PIPE, text=True).stdout.splitlines() if rows else []
for r, d in zip(rows, dem):
    print(f"{d[:90]:90s} VGPR {r.get('VGPRs', '?'):>4} AGPR {r.get('AGPRs', '?'):>3} SGPR {r.get('TotalSGPRs', '?'):>4} spillV {r.get('VGPRs Spill', '?'):>4} spillS {r.get('SGPRs Spill', '?'):>3} scratch {r.get('ScratchSize [bytes/lane]', '?'):>4} occ {r.get('Occupancy [waves/SIMD]', '?')}")
