"""Register / LDS / occupancy table of the kernels of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
usage: kernel_resources.py <file.hip> [name-substring]"""
import re, subprocess, sys
src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-c", src, "-o", "/dev/null"] + sys.argv[3:] + [
                      "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
name, d = None, {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name, d = m.group(1), {}
        continue
    m = re.search(r"remark:\s+(.+?):\s+(\S+)\s+\[-Rpass", line)
    if m and name:
        d[m.group(1).strip()] = m.group(2)
        if m.group(1).strip().startswith("LDS Size"):
            if pat in name:
                dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                dem = re.sub(r"\(.*", "", dem)
                print(f"{dem[:70]:70s} VGPR {d.get('VGPRs','?'):>4} AGPR {d.get('AGPRs','?'):>3} SGPR {d.get('TotalSGPRs','?'):>3} spill {d.get('VGPRs Spill','?'):>3} "
                      f"occ {d.get('Occupancy [waves/SIMD]','?'):>2} LDS {d.get('LDS Size [bytes/block]','?')}")
