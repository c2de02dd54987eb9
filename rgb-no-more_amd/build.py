"""Build librgbnm.so (HIP kernels + C ABI, gfx950 only) and librgbnm_reader.so (host libjpeg reader) in-tree."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librgbnm.so")
READER_LIB = os.path.join(HERE, "librgbnm_reader.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-result"]
FLAGS += os.environ.get("RGBNM_HIPCC_FLAGS", "").split()      # experiments only (e.g. -DWRES_EXP)


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=False):
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "rgbnm.h"))
    objs = []
    jobs = []
    for s in srcs:
        o = s[:-4] + ".o"
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        extra = []                      # per-file switches: a `// hipcc-flags: ...` line in the first 40 lines of the source
        with open(s) as fh:
            for _, line in zip(range(40), fh):
                if line.startswith("// hipcc-flags:"):
                    extra += line.split(":", 1)[1].split()
        cmd = [HIPCC] + FLAGS + extra + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stderr}")
        return r.stderr

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for warn in ex.map(cc, jobs):
                if verbose and warn:
                    print(warn)
    if force or jobs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
    return LIB


def build_reader(force=False):
    src = os.path.join(CSRC, "reader.c")
    if not os.path.exists(src):
        return None
    if force or _stale(READER_LIB, [src]):
        cmd = ["gcc", "-O2", "-fPIC", "-shared", "-I/opt/conda/include", src, "-o", READER_LIB,
               "-L/opt/conda/lib", "-ljpeg", "-lpthread", "-Wl,-rpath,/opt/conda/lib"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"gcc failed for reader.c:\n{r.stderr}")
    return READER_LIB


if __name__ == "__main__":
    print(build_lib(force="-f" in sys.argv, verbose=True))
    print(build_reader(force="-f" in sys.argv))
