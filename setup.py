"""Build / install of focnerf_amd with the reference's extension-module API (north_star: "keeps the existing ... setup.py extension API").

The reference ships four setup.py files, one per CUDA extension, each installing a top-level torch-extension module — `_raymarching`,
`_gridencoder`, `_freqencoder`, `_ffmlp` (raymarching/setup.py, gridencoder/setup.py, ...) — that its Python wrappers import first
(`try: import _raymarching as _backend`, raymarching/raymarching.py:9-12). This file installs modules of exactly those names: host-only
pybind11 shims (focnerf_amd/csrc/ext/*.cpp) over the C ABI of libfocnerf_hip.so, which holds the hand-written gfx950 kernels.

    python setup.py build_ext --inplace   # libfocnerf_hip.so (hipcc, gfx950) + the four modules, in-tree (repo root / focnerf_amd/ext)
    pip install --no-build-isolation .    # the same, installed: the reference's wrappers then run on the HIP kernels unedited

`build_ext` first runs focnerf_amd/csrc/Makefile (hipcc --offload-arch=gfx950); the extension modules themselves contain no device code
and compile with the host compiler. There is no CUDA path and no CPU fallback.
"""
import os
import subprocess

from setuptools import Extension, find_packages, setup
from setuptools.command.build_ext import build_ext

HERE = os.path.dirname(os.path.abspath(__file__))
ROCM = os.environ.get("ROCM_HOME", "/opt/rocm")


class BuildHip(build_ext):
    def run(self):
        subprocess.check_call(["make", "-C", os.path.join(HERE, "focnerf_amd", "csrc"), "-j4", f"HIPCC={ROCM}/bin/hipcc", "ARCH=gfx950"])
        super().run()
        if self.inplace:                      # the in-tree layout the tests and INTEGRATION.md use: focnerf_amd/ext on PYTHONPATH
            subprocess.check_call(["make", "-C", os.path.join(HERE, "focnerf_amd", "csrc", "ext"), "-j4"])


def shim(name):
    import torch
    tdir = os.path.dirname(torch.__file__)
    return Extension(
        name=f"_{name}",
        sources=[os.path.join("focnerf_amd", "csrc", "ext", f"{name}_ext.cpp")],
        include_dirs=[os.path.join(tdir, "include"), os.path.join(tdir, "include", "torch", "csrc", "api", "include"), os.path.join(ROCM, "include")],
        define_macros=[("__HIP_PLATFORM_AMD__", "1"), ("USE_ROCM", "1"), ("TORCH_API_INCLUDE_EXTENSION_H", None), ("TORCH_EXTENSION_NAME", f"_{name}"),
                       ("_GLIBCXX_USE_CXX11_ABI", str(int(torch._C._GLIBCXX_USE_CXX11_ABI)))],
        extra_compile_args=["-O2", "-std=c++17", "-fvisibility=hidden", "-Wno-deprecated-declarations"],
        library_dirs=[os.path.join(tdir, "lib"), os.path.join(HERE, "focnerf_amd")],
        libraries=["c10", "c10_hip", "torch_cpu", "torch", "torch_python"],
        extra_link_args=["-l:libfocnerf_hip.so", "-Wl,-rpath,$ORIGIN/focnerf_amd", "-Wl,-rpath,$ORIGIN/..", f"-Wl,-rpath,{os.path.join(tdir, 'lib')}"],
        language="c++",
    )


if __name__ == "__main__":
    setup(
        name="focnerf_amd",
        version="0.2.0",
        description="FOCNeRF volume-rendering hot path on MI355X (gfx950): hand-written HIP kernels behind the reference's extension API",
        packages=find_packages(include=["focnerf_amd", "focnerf_amd.*"]),
        package_data={"focnerf_amd": ["libfocnerf_hip.so"]},
        ext_modules=[shim(n) for n in ("raymarching", "gridencoder", "freqencoder", "ffmlp")],
        cmdclass={"build_ext": BuildHip},
        zip_safe=False,
    )
