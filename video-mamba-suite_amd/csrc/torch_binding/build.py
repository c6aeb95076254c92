"""Builds video-mamba-suite_amd/_vms_torch.so, the compiled PyTorch binding of libvms_hip.so (vms_torch.cpp), in-tree
with one g++ command (no JIT cache: the file travels with the tree).  Skips the build when the output is newer than its
inputs.  python video-mamba-suite_amd/csrc/torch_binding/build.py [--force]"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(PKG, "_vms_torch.so")
SRC = os.path.join(HERE, "vms_torch.cpp")
HDR = os.path.join(os.path.dirname(PKG), "include", "vms_hip.h")
LIB = os.path.join(PKG, "vms_hip", "libvms_hip.so")


STAMP = OUT + ".stamp"


def stamp():
    """What a built binding depends on besides its sources: the torch build and the Python ABI it was compiled against
    (a binding from another torch / interpreter fails to import and would silently leave every call on ctypes)."""
    import torch
    return f"torch {torch.__version__} | python {sysconfig.get_config_var('SOABI')} | hip {getattr(torch.version, 'hip', None)}"


def up_to_date():
    if not os.path.exists(OUT) or not os.path.exists(STAMP):
        return False
    if not all(os.path.getmtime(OUT) >= os.path.getmtime(f) for f in (SRC, HDR, __file__)):
        return False
    with open(STAMP) as f:
        return f.read().strip() == stamp()


def main(force=False):
    if not force and up_to_date():
        return OUT
    import torch
    from torch.utils import cpp_extension as ce
    rocm = os.environ.get("ROCM_HOME", "/opt/rocm")
    inc = ce.include_paths() + [os.path.join(rocm, "include"), sysconfig.get_paths()["include"]]
    libdirs = ce.library_paths() + [os.path.join(rocm, "lib"), os.path.dirname(LIB)]
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    cmd = (["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
            "-DTORCH_EXTENSION_NAME=_vms_torch", "-DTORCH_API_INCLUDE_EXTENSION_H", f"-D_GLIBCXX_USE_CXX11_ABI={abi}"]
           + [f"-I{p}" for p in inc] + [SRC, "-o", OUT] + [f"-L{p}" for p in libdirs]
           + ["-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-ltorch_python", "-lamdhip64", "-lvms_hip",
              "-Wl,-rpath,$ORIGIN/vms_hip"] + [f"-Wl,-rpath,{p}" for p in ce.library_paths() + [os.path.join(rocm, "lib")]])
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(stamp() + "\n")
    return OUT


if __name__ == "__main__":
    print(main(force="--force" in sys.argv))
