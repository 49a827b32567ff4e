"""VGPRs / scratch / occupancy / LDS of every kernel of the library, from hipcc's resource remarks (no GPU needed):
python tools/kernel_resources.py > profiles/rNN_kernel_resources.txt"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "simdjson-go_amd", "csrc")


def kernels_of(src, flags=()):
    """[(kernel, VGPRs, scratch bytes per lane, waves per SIMD, LDS bytes per block)] of one .hip file"""
    out = subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-c",
                          "-Rpass-analysis=kernel-resource-usage", *flags, "-o", os.devnull, os.path.join(CS, src)],
                         capture_output=True, text=True).stderr
    rows, cur = [], {}
    for line in out.splitlines():
        m = re.search(r"remark: +(Function Name|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1).split(" ")[0], m.group(2)
        if k == "Function":
            cur = {"name": v}
        cur[k] = v
        if k == "LDS":
            name = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
            name = name.replace("(anonymous namespace)::", "").replace("sj::", "").replace("void ", "")
            name = re.sub(r"\((?!.*>).*$", "", name)  # drop the argument list (behind the last template bracket)
            rows.append((name, int(cur["VGPRs"]), int(cur["ScratchSize"]), int(cur["Occupancy"]), int(cur["LDS"])))
    return rows


if __name__ == "__main__":
    print("# %-14s %-86s %5s %8s %5s %7s" % ("file", "kernel", "VGPRs", "scratch", "occ", "LDS"))
    for f in sorted(f for f in os.listdir(CS) if f.endswith(".hip")):
        for name, vgprs, scratch, occ, lds in kernels_of(f):
            print("%-16s %-86s %5d %8d %5d %7d" % (f, name[:86], vgprs, scratch, occ, lds))
