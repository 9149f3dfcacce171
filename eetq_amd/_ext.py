"""Builder / loader of the compiled operator module ``EETQ`` (eetq_amd/csrc/torch_ext.cpp).

The module is the reference's pybind boundary (csrc/eetpy.cpp:7-19) re-hosted on libeetq_amd.so: host C++ only, compiled
with g++ against the installed torch headers and linked to the in-tree C-ABI library.  It is kept in-tree at the repo root
(``EETQ.cpython-*.so``) so that ``import EETQ`` works exactly as with the reference, and so that it travels with the tree.
"""
import importlib
import os
import shutil
import subprocess
import sys
import sysconfig

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
SRC = os.path.join(_HERE, "csrc", "torch_ext.cpp")
HDR = os.path.join(ROOT, "include", "eetq_amd.h")
SO_PATH = os.path.join(ROOT, "EETQ" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def _stale():
    if not os.path.exists(SO_PATH):
        return True
    t = os.path.getmtime(SO_PATH)
    return any(os.path.exists(p) and os.path.getmtime(p) > t for p in (SRC, HDR))


def build(force=False, verbose=False):
    """g++ -shared torch_ext.cpp -> <repo>/EETQ.cpython-*.so (about a minute: torch/extension.h is a large header)."""
    from . import _lib
    _lib.build()
    if not (force or _stale()):
        return SO_PATH
    import torch
    from torch.utils import cpp_extension as ce
    cxx = shutil.which("g++") or shutil.which("c++")
    if cxx is None:
        raise RuntimeError("eetq_amd: no C++ compiler found; cannot build the EETQ extension module")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", SRC, "-o", SO_PATH + ".tmp",
           "-DTORCH_EXTENSION_NAME=EETQ", "-DTORCH_API_INCLUDE_EXTENSION_H", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI), "-Wno-attributes"]
    for inc in ce.include_paths() + [sysconfig.get_paths()["include"], os.path.join(rocm, "include")]:
        cmd += ["-isystem", inc]
    cmd += ["-L" + tlib, "-ltorch", "-ltorch_cpu", "-ltorch_python", "-lc10", "-lc10_hip", "-L" + _HERE, "-leetq_amd",
            "-Wl,-rpath," + tlib, "-Wl,-rpath,$ORIGIN/eetq_amd"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(" ".join(cmd))
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("eetq_amd: building the EETQ extension module failed:\n" + res.stdout[-4000:])
    os.replace(SO_PATH + ".tmp", SO_PATH)
    return SO_PATH


def load():
    """Import (building first when stale and a compiler is available) and return the compiled ``EETQ`` module."""
    if _stale() and shutil.which("g++"):
        import fcntl
        with open(SO_PATH + ".lock", "w") as lock:  # several ranks may get here together
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                if _stale():
                    build()
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)
    import torch  # noqa: F401  (libtorch must be mapped before the extension)
    from . import _lib
    _lib.lib()    # maps libeetq_amd.so after torch's HIP runtime (see _lib._share_hip_runtime_with_torch)
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    mod = importlib.import_module("EETQ")
    if not hasattr(mod, "__eetq_amd_version__"):
        raise ImportError("a module named EETQ is importable but it is not this library's (%s)" % getattr(mod, "__file__", "?"))
    return mod
