"""Build experiments/lib/libtgis_experiments.so: every product source compiled with -DTGIS_EXPERIMENTS (which re-enables the
entry points that live behind that macro in csrc/gptq.hip and csrc/attention.hip) plus experiments/csrc/*.hip.
Not run by __graft_entry__.build(); `python experiments/build.py` when an experiment is wanted."""
import glob
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "text-generation-inference_amd", "csrc")


def build() -> str:
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip"))) + sorted(glob.glob(os.path.join(HERE, "csrc", "*.hip")))
    objdir = os.path.join(HERE, "lib", "obj")
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DTGIS_EXPERIMENTS", "-I", CSRC,
             "-I", os.path.join(HERE, "csrc")]

    def one(src):
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        subprocess.run(["hipcc", *flags, "-c", src, "-o", obj], check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 1)) as pool:
        objs = list(pool.map(one, srcs))
    out = os.path.join(HERE, "lib", "libtgis_experiments.so")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs], check=True)
    return out


if __name__ == "__main__":
    print(build())
