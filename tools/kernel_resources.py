"""Per-kernel register / spill / occupancy table from hipcc -Rpass-analysis=kernel-resource-usage."""
import re, subprocess, sys, glob, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
files = sys.argv[1:] or sorted(glob.glob(os.path.join(root, "gotennet_amd/csrc/*.hip")))
keys = [("VGPRs", "vgpr"), ("AGPRs", "agpr"), ("SGPRs", "sgpr"), ("SGPRs Spill", "sspill"), ("VGPRs Spill", "vspill"),
        (r"Occupancy \[waves/SIMD\]", "occ"), (r"LDS Size \[bytes/block\]", "lds")]
for f in files:
    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                          "--cuda-device-only", "-c", "-Rpass-analysis=kernel-resource-usage", f, "-o", "/dev/null"],
                         capture_output=True, text=True).stderr
    rows, cur = [], None
    for l in out.splitlines():
        m = re.search(r"Function Name: (\S+)", l)
        if m:
            cur = {"name": m.group(1)}; rows.append(cur); continue
        for k, short in keys:
            m = re.search(r"remark:\s+" + k + r": (\d+)", l)
            if m and cur is not None:
                cur[short] = m.group(1)
    for r in rows:
        n = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
        n = re.sub(r"\(.*", "", n).replace("void ", "")[:62]
        print(f"{n:62s} " + " ".join(f"{s} {r.get(s, '-'):>5s}" for _, s in keys))
