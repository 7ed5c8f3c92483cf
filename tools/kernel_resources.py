"""Register / LDS / scratch budget of every kernel of libtgis_hip.so, from hipcc's -Rpass-analysis=kernel-resource-usage remarks
(cross-compiles: needs no GPU).   python tools/kernel_resources.py > profiles/rNN_kernel_resources.txt"""
import glob
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "text-generation-inference_amd", "csrc")
FIELDS = [("Function Name", "name"), ("TotalSGPRs", "sgpr"), ("VGPRs", "vgpr"), ("AGPRs", "agpr"),
          ("ScratchSize [bytes/lane]", "scratch"), ("Occupancy [waves/SIMD]", "occ"), ("LDS Size [bytes/block]", "lds")]


def main():
    print("file | kernel | SGPR | VGPR | AGPR | scratch B/lane | waves/SIMD | static LDS B/block")
    for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
        p = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/dev/null",
                            "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
        cur = {}
        rows = []
        for line in p.stderr.splitlines():
            m = re.search(r"remark: [^:]*:\d+:\d+: +(.*?): (\S+) \[-Rpass-analysis", line) or \
                re.search(r"remark: +(.*?): (\S+) \[-Rpass-analysis", line)
            if not m:
                continue
            for label, key in FIELDS:
                if m.group(1).strip() == label:
                    if key == "name" and cur:
                        rows.append(cur)
                        cur = {}
                    cur[key] = m.group(2)
        if cur:
            rows.append(cur)
        for r in rows:
            name = subprocess.run(["c++filt", r.get("name", "?")], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(anonymous namespace\)::", "", name)
            name = re.sub(r"\(.*\)$", "", name)[:110]
            print(f"{os.path.basename(src)} | {name} | {r.get('sgpr')} | {r.get('vgpr')} | {r.get('agpr')} | {r.get('scratch')} | "
                  f"{r.get('occ')} | {r.get('lds')}")


if __name__ == "__main__":
    main()
