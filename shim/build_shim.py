#!/usr/bin/env python3
"""Builds the libtorch shim (xllm::kernel::mi355::* + AttentionImpl) in-tree as shim/xllm_mi355_shim*.so.
Pure host C++ (g++): the kernels live in libxllm_mi355.so, which this links against."""
import os
import subprocess
import sys
import sysconfig

import torch
from torch.utils import cpp_extension as ce

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def main():
    name = "xllm_mi355_shim"
    out = os.path.join(HERE, name + sysconfig.get_config_var("EXT_SUFFIX"))
    srcs = [os.path.join(HERE, f) for f in ("mi355_ops_api.cpp", "mi355_attention.cpp", "mi355_process_group.cpp", "pybind.cpp",
                                               os.path.join("stub", "kernels", "dcu", "attention_runner_stub.cpp"))]
    stub = os.path.join(HERE, "stub")   # stand-ins for the two reference headers mi355_attention.h includes
    deps = srcs + [os.path.join(HERE, "mi355_ops_api.h"), os.path.join(HERE, "mi355_attention.h"),
                   os.path.join(HERE, "mi355_process_group.h"),
                   os.path.join(ROOT, "include", "xllm_mi355.h"),
                   os.path.join(stub, "layers", "common", "attention_metadata.h"),
                   os.path.join(stub, "framework", "kv_cache", "kv_cache.h"),
                   os.path.join(stub, "kernels", "dcu", "attention_runner.h")]
    if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    inc = ce.include_paths("cuda") if hasattr(ce, "include_paths") else []
    inc += ["/opt/rocm/include", sysconfig.get_paths()["include"], stub]
    libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
    kern = os.path.join(ROOT, "xllm_amd", "lib")
    flags = ["-O2", "-std=c++17", "-fPIC", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
             f"-DTORCH_EXTENSION_NAME={name}", "-D_GLIBCXX_USE_CXX11_ABI=" + str(int(torch._C._GLIBCXX_USE_CXX11_ABI)),
             "-Wno-deprecated-declarations"] + [f"-I{i}" for i in inc]
    # one object per source, compiled in parallel and only when the source or a header changed (a libtorch translation unit
    # takes about a minute of g++)
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [d for d in deps if d.endswith(".h")]
    jobs, objs = [], []
    for src in srcs:
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if not os.path.exists(obj) or any(os.path.getmtime(obj) < os.path.getmtime(d) for d in [src] + headers):
            jobs.append(subprocess.Popen(["g++"] + flags + ["-c", src, "-o", obj]))
    for j in jobs:
        if j.wait() != 0:
            raise subprocess.CalledProcessError(j.returncode, j.args)
    cmd = ["g++", "-shared", "-fPIC"] + objs
    cmd += [f"-L{libdir}", f"-L{kern}", "-lxllm_mi355", "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-lc10", "-lc10_hip",
            "-ltorch_python", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,$ORIGIN/../xllm_amd/lib", "-o", out]
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(main())
