"""ORACLE build recipe (test infrastructure, not product code).

`build_oracle()`   gcc-compiles oracle/roi_ops.c + oracle/csc_ops.c -> oracle/_build/liboracle_roi.so
`build_ref()`      when /root/reference is present, compiles the reference's own
                   detectron2/layers/csrc/ROIAlign/ROIAlign_cpu.cpp *where it lies* (no copy)
                   together with oracle/ref_binding.cpp (a 20-line pybind shim of ours) into
                   oracle/_ref/d2_roialign_ref*.so.  Outputs only go to oracle/_ref/ (git-ignored,
                   NOT gpurun-ignored, so the built .so travels to the GPU box).
`build_ref_pcl()`  same for projects/WSL/wsl/layers/csrc/pcl_loss/pcl_loss_cpu.cpp + oracle/ref_pcl_binding.cpp
                   -> oracle/_ref/wsl_pcl_ref.so (the reference's PCL loss, which it always runs on the CPU).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD_DIR = os.path.join(HERE, "_build")
REF_DIR = os.path.join(HERE, "_ref")
ORACLE_SO = os.path.join(BUILD_DIR, "liboracle_roi.so")
ORACLE_SOURCES = ("roi_ops.c", "csc_ops.c")
REF_SRC = "/root/reference/detectron2/layers/csrc/ROIAlign/ROIAlign_cpu.cpp"
REF_INC = "/root/reference/detectron2/layers/csrc"


PCL_SRC = "/root/reference/projects/WSL/wsl/layers/csrc/pcl_loss/pcl_loss_cpu.cpp"
PCL_INC = "/root/reference/projects/WSL/wsl/layers/csrc"


def _newer(src, dst):
    return (not os.path.exists(dst)) or os.path.getmtime(src) > os.path.getmtime(dst)


def build_oracle(force=False):
    srcs = [os.path.join(HERE, f) for f in ORACLE_SOURCES]
    os.makedirs(BUILD_DIR, exist_ok=True)
    if force or any(_newer(src, ORACLE_SO) for src in srcs):
        # no -march / -ffast-math: keep plain IEEE fp32, no FMA contraction
        cmd = ["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off"] + srcs + [
            "-o", ORACLE_SO, "-lm"]
        subprocess.check_call(cmd)
    return ORACLE_SO


def ref_so_path():
    if not os.path.isdir(REF_DIR):
        return None
    for f in sorted(os.listdir(REF_DIR)):
        if f.startswith("d2_roialign_ref") and f.endswith(".so"):
            return os.path.join(REF_DIR, f)
    return None


def _compile_torch_ext(name, src, shim, inc_dir, out):
    import torch
    from torch.utils import cpp_extension as ce
    import sysconfig

    inc = ce.include_paths() + [sysconfig.get_paths()["include"], inc_dir]
    libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", "-DTORCH_EXTENSION_NAME=" + name,
           "-DTORCH_API_INCLUDE_EXTENSION_H", "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    for i in inc:
        cmd += ["-I", i]
    cmd += [src, shim, "-o", out, "-L", libdir, "-Wl,-rpath," + libdir,
            "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_python"]
    subprocess.check_call(cmd)
    return out


def build_ref(force=False):
    """Compile the reference ROIAlign CPU source in place -> oracle/_ref/. Returns path or None."""
    if not os.path.exists(REF_SRC):
        return ref_so_path()  # GPU box: use the prebuilt file if it travelled
    existing = ref_so_path()
    shim = os.path.join(HERE, "ref_binding.cpp")
    if existing and not force and not _newer(shim, existing):
        return existing
    os.makedirs(REF_DIR, exist_ok=True)
    return _compile_torch_ext("d2_roialign_ref", REF_SRC, shim, REF_INC, os.path.join(REF_DIR, "d2_roialign_ref.so"))


def build_ref_pcl(force=False):
    """Compile the reference PCL loss (CPU) in place -> oracle/_ref/wsl_pcl_ref.so. Returns path or None."""
    out = os.path.join(REF_DIR, "wsl_pcl_ref.so")
    if not os.path.exists(PCL_SRC):
        return out if os.path.exists(out) else None
    shim = os.path.join(HERE, "ref_pcl_binding.cpp")
    if os.path.exists(out) and not force and not _newer(shim, out):
        return out
    os.makedirs(REF_DIR, exist_ok=True)
    return _compile_torch_ext("wsl_pcl_ref", PCL_SRC, shim, PCL_INC, out)


if __name__ == "__main__":
    print(build_oracle(force="--force" in sys.argv))
    print(build_ref(force="--force" in sys.argv))
    print(build_ref_pcl(force="--force" in sys.argv))
