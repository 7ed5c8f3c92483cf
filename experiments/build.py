"""Build experiments/lib/libtgis_experiments.so: the product sources plus experiments/csrc/*.hip.  Two experiment units
(gptq_experiments.hip, attention_experiments.hip) INCLUDE the product's csrc/gptq.hip / csrc/attention.hip whole and add their
entry points behind them, so those two product files are not compiled a second time.  The product sources carry no
experiment switches (round 5).  Not run by __graft_entry__.build(); `python experiments/build.py` when an experiment is wanted."""
import glob
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "text-generation-inference_amd", "csrc")


def build() -> str:
    wrapped = {"gptq.hip", "attention.hip"}  # compiled through experiments/csrc/{gptq,attention}_experiments.hip
    srcs = sorted(p for p in glob.glob(os.path.join(CSRC, "*.hip")) if os.path.basename(p) not in wrapped)
    srcs += sorted(glob.glob(os.path.join(HERE, "csrc", "*.hip")))
    objdir = os.path.join(HERE, "lib", "obj")
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", CSRC,
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
