"""Development A/B builds: recompile a subset of csrc/*.hip with extra -D flags and link them with the regular objects of every
other translation unit into gabotorch_amd/libgabo_hip_<tag>.so (select it with GABO_HIP_LIB=...; never the product library).

    python tools/ab_build.py <tag> spd_pairwise.hip[,other.hip] -DGABO_QL_NO_LOOKAHEAD [...]
"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gabotorch_amd import _build  # noqa: E402


def main():
    tag, units = sys.argv[1], sys.argv[2].split(",")
    flags = sys.argv[3:]
    _build.build()                                     # the regular objects must exist
    outdir = os.path.join(_build.OBJ, "ab_" + tag)
    os.makedirs(outdir, exist_ok=True)
    objs = []
    for src in _build.sources():
        name = os.path.basename(src)
        if name in units:
            obj = os.path.join(outdir, os.path.splitext(name)[0] + ".o")
            cmd = [_build._hipcc()] + _build.FLAGS + flags + ["-c", src, "-o", obj]
            subprocess.run(cmd, check=True)
        else:
            obj = os.path.join(_build.OBJ, os.path.splitext(name)[0] + ".o")
        objs.append(obj)
    lib = os.path.join(_build.PKG, f"libgabo_hip_{tag}.so")
    subprocess.run([_build._hipcc(), "-shared", "-fPIC", f"--offload-arch={_build.ARCH}"] + objs + ["-o", lib], check=True)
    print(lib)


if __name__ == "__main__":
    main()
