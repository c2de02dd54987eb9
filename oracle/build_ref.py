"""Recipe: compile the REFERENCE's own host reader into oracle/_ref/ (test infrastructure only).

The reference's `dct_manip/dct_manip.cpp` (pybind11 + libtorch + IJG libjpeg) is compiled
*from where it lies* under /root/reference -- no reference source is copied into this repo.
Output: oracle/_ref/dct_manip_ref*.so (git-ignored; travels to the GPU box with the snapshot).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the result.
If /root/reference is absent (GPU box) this is a no-op: the prebuilt .so is used as-is.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
SRC = "/root/reference/dct_manip/dct_manip.cpp"
NAME = "dct_manip_ref"


def built_path():
    if not os.path.isdir(OUT):
        return None
    for f in os.listdir(OUT):
        if f.startswith(NAME) and f.endswith(".so"):
            return os.path.join(OUT, f)
    return None


def build(verbose=False):
    if not os.path.exists(SRC):
        return built_path()
    p = built_path()
    if p and os.path.getmtime(p) >= os.path.getmtime(SRC):
        return p
    os.makedirs(OUT, exist_ok=True)
    from torch.utils.cpp_extension import load
    load(name=NAME, sources=[SRC], extra_cflags=["-std=c++17", "-O2"],
         extra_include_paths=["/opt/conda/include"],
         extra_ldflags=["-L/opt/conda/lib", "-ljpeg", "-Wl,-rpath,/opt/conda/lib"],
         build_directory=OUT, verbose=verbose, is_python_module=True)
    return built_path()


def load_ref():
    """Import the compiled reference reader as a python module (or None if not built)."""
    p = built_path()
    if p is None:
        return None
    import importlib.util
    import torch  # noqa: F401  (libtorch symbols must be loaded first)
    # the pybind module inside is named 'dct_manip' (PYBIND11_MODULE(TORCH_EXTENSION_NAME...))
    spec = importlib.util.spec_from_file_location(NAME, p)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build(verbose="-v" in sys.argv))
