"""Builds gabotorch_amd/libgabo_hip.so from csrc/*.hip (device code, hipcc for gfx950: cross-compiles without a GPU) and host/*.cpp (host-only C++).

In-tree on purpose: the .so travels to the GPU box with the repository snapshot.  `python -m gabotorch_amd._build`
rebuilds; `__graft_entry__.build()` calls `build()`.
"""
import concurrent.futures
import os
import shutil
import subprocess
import time
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")          # device code (*.hip) and its headers
HOST = os.path.join(PKG, "host")          # host-only C++ above the launches (*.cpp): compiled as plain C++ against the HIP runtime API
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(PKG, "libgabo_hip.so")
ARCH = "gfx950"
# -amdgpu-mfma-vgpr-form: MFMA accumulators allocated in VGPRs (gfx950 has one unified register file); without it hipcc keeps
# them in AGPRs and copies all of them to and from VGPRs around every loop iteration that carries an accumulator (sphere_pairwise.hip)
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffp-contract=off", "-Wall", "-Wno-unused-variable",
         "-mllvm", "-amdgpu-mfma-vgpr-form"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libgabo_hip.so (set HIPCC=/path/to/hipcc)")


# translation units that instantiate the big unrolled templates, longest first: they are started before everything else so
# that the parallel build does not end on one of them
_HEAVY = ["spd_tr_solve_duo.hip", "spd_tr_solve_le_hi.hip", "spd_tr_solve_le_mid.hip", "spd_tr_solve_le.hip", "spd_tr_solve_hi.hip", "spd_tr_le_hi.hip", "spd_tr_le.hip", "spd_tr_solve.hip",
          "spd_backward_duo.hip", "spd_pairwise_wide3.hip", "spd_pairwise_wide2.hip", "spd_tr_wide.hip", "spd_acq.hip", "spd_pairwise_wide.hip", "spd_backward.hip", "spd_tr.hip",
          "spd_pairwise.hip", "spd_tr_solve_frob.hip", "spd_tr_frob.hip", "spd_acq_le.hip"]


def sources():
    names = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    names.sort(key=lambda f: _HEAVY.index(f) if f in _HEAVY else len(_HEAVY))
    host = sorted(f for f in os.listdir(HOST) if f.endswith(".cpp")) if os.path.isdir(HOST) else []
    return [os.path.join(CSRC, f) for f in names] + [os.path.join(HOST, f) for f in host]


def _rocm_include():
    return os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(_hipcc()))), "include")


HOST_FLAGS = ["-x", "c++", "-O3", "-std=c++17", "-fPIC", "-pthread", "-ffp-contract=off", "-Wall", "-D__HIP_PLATFORM_AMD__"]


def _deps():
    hdr = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    hdr.append(os.path.join(os.path.dirname(PKG), "include", "gabo_hip.h"))
    hdr.append(os.path.abspath(__file__))
    return hdr


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _recorded_deps(depfile):
    """headers a translation unit actually included, as hipcc recorded them at its last compilation (-MD); None if unknown"""
    try:
        text = open(depfile).read().replace("\\\n", " ")
    except OSError:
        return None
    deps = [t for t in text.split(":", 1)[-1].split() if not t.startswith("/opt/") and not t.startswith("/usr/")]
    return deps if all(os.path.exists(d) for d in deps) else None


def _compile(src, extra):
    obj = os.path.join(OBJ, os.path.splitext(os.path.basename(src))[0] + ".o")
    depfile = obj[:-2] + ".d"
    known = _recorded_deps(depfile)
    deps = ([src, os.path.abspath(__file__)] + known) if known is not None else ([src] + _deps())
    if not _stale(obj, deps):
        return obj, False
    if src.endswith(".cpp"):       # host-only translation unit: the same clang, no offload pass
        cmd = [_hipcc()] + HOST_FLAGS + ["-I", _rocm_include()] + [f for f in extra if f.startswith("-D")] + ["-MD", "-MF", depfile, "-c", src, "-o", obj]
    else:
        cmd = [_hipcc()] + FLAGS + extra + ["-MD", "-MF", depfile, "-c", src, "-o", obj]
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if os.environ.get("GABO_BUILD_TIMES"):
        print(f"{time.time() - t0:7.1f} s  {os.path.basename(src)}", flush=True)
    return obj, True


def build(force=False, extra_flags=(), verbose=False):
    """Compile every csrc/*.hip for gfx950 and every host/*.cpp, link libgabo_hip.so.  Returns the library path."""
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            path = os.path.join(OBJ, f)
            if os.path.isdir(path):                    # (tools/ab_build.py keeps its variant objects in sub-directories)
                shutil.rmtree(path)
            else:
                os.remove(path)
    srcs = sources()
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(os.cpu_count() or 4, 8, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, list(extra_flags)), srcs))
    objs = [o for o, _ in res]
    if any(changed for _, changed in res) or _stale(LIB, objs):
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}"] + objs + ["-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {LIB} ({os.path.getsize(LIB)} bytes) from {len(srcs)} sources")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, extra_flags=[a for a in sys.argv[1:] if a.startswith("-D")], verbose=True)
