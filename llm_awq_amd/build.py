"""In-tree build of the native pieces (gfx950 only):

  llm_awq_amd/lib/libawq_cdna4.so          hipcc  --offload-arch=gfx950   kernels + the C ABI (include/awq_cdna4.h)
  llm_awq_amd/ext/awq_inference_engine*.so g++    pybind/torch shim over the C ABI (no device code, no hipify)

Both are git-ignored but travel to the GPU box with the repo snapshot.  `python -m llm_awq_amd.build`
rebuilds whatever is older than its sources; hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import subprocess
import sys
import sysconfig
import time

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
EXT_DIR = os.path.join(PKG, "ext")
LIB_PATH = os.path.join(LIB_DIR, "libawq_cdna4.so")
EXT_SUFFIX = sysconfig.get_config_var("EXT_SUFFIX") or ".so"
EXT_PATH = os.path.join(EXT_DIR, "awq_inference_engine" + EXT_SUFFIX)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

HIP_SOURCES = ["awq_gemv.hip", "awq_gemv_cdna4.hip", "awq_gemv_dma.hip", "awq_v2_kernels.hip", "awq_gemm.hip", "awq_gemm_plan.hip", "awq_gemm_v4n.hip", "awq_gemm_v6.hip", "awq_skinny_cdna4.hip", "awq_midm_cdna4.hip", "awq_util.hip", "awq_oneshot.hip", "awq_capi.hip"]
# (round 4: the experiment kernels that AWQ_PROBES=1 builds once compiled -- v5, v6w -- are history: tools/EXPERIMENTS.md names the commits)
PROBE_SOURCES = []
HIP_DEPS = ["awq_device.hpp", "awq_dma.hpp", "awq_kernels.hpp", os.path.join(ROOT, "include", "awq_cdna4.h")]


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd, what):
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout[-4000:] + "\n" + r.stderr[-8000:] + "\n")
        raise RuntimeError(f"building {what} failed (exit {r.returncode}): {' '.join(cmd[:6])} ...")
    return time.time() - t0


def build_lib(force: bool = False, verbose: bool = True) -> str:
    """Every .hip source is compiled to its own object (in parallel, only when older than its source or the shared headers),
    then linked: an edit to one kernel file rebuilds one object."""
    from concurrent.futures import ThreadPoolExecutor

    deps = [d if os.path.isabs(d) else os.path.join(CSRC, d) for d in HIP_DEPS]
    obj_dir = os.path.join(LIB_DIR, "obj")
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
             # keep MFMA results in VGPRs: the matrix-core dequant feeds v_cvt_pk_bf16_f32 directly
             # (the AGPR form costs one v_accvgpr_read per value)
             "-mllvm", "-amdgpu-mfma-vgpr-form",
             # AWQ_PROBES=1: compile the timing-only probes (linear-read / null kernels, GEMM v4 no-DMA / no-epilogue)
             # behind the gemv_probe / gemm_v6 (tens digit) knobs; a default build has no knob that changes results
             *(["-DAWQ_ENABLE_PROBES"] if os.environ.get("AWQ_PROBES") == "1" else []),
             # the command processor preloads the first 16 kernel-argument dwords into SGPRs at dispatch (gfx950): the waves of a
             # decode launch issue their first loads without waiting for an s_load of the arguments -- +2.5 % decode tok/s
             # (profiles/r02_kernarg_preload_ab.txt); AWQ_KERNARG_PRELOAD=0 builds without it
             *([] if os.environ.get("AWQ_KERNARG_PRELOAD") == "0" else ["-mllvm", "-amdgpu-kernarg-preload-count=16"])]
    stamp = os.path.join(obj_dir, "flags.txt")
    same_flags = os.path.exists(stamp) and open(stamp).read() == " ".join(flags)
    if not force and not _newer(LIB_PATH, [os.path.join(CSRC, n) for n in HIP_SOURCES] + deps) and (same_flags or not os.path.exists(stamp)):
        return LIB_PATH  # up to date (the GPU box receives the linked library without the per-file objects / the stamp)
    os.makedirs(obj_dir, exist_ok=True)
    if not same_flags:
        force = True
    sources = HIP_SOURCES + (PROBE_SOURCES if os.environ.get("AWQ_PROBES") == "1" else [])
    jobs = []
    for name in sources:
        src, obj = os.path.join(CSRC, name), os.path.join(obj_dir, name + ".o")
        if force or _newer(obj, [src] + deps):
            jobs.append((name, [HIPCC, *flags, "-c", src, "-o", obj]))
    t0 = time.time()
    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(lambda j: _run(j[1], j[0]), jobs))
    objs = [os.path.join(obj_dir, n + ".o") for n in sources]
    if jobs or _newer(LIB_PATH, objs):
        _run([HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", LIB_PATH], "libawq_cdna4.so")
        open(stamp, "w").write(" ".join(flags))
        if verbose:
            print(f"[llm_awq_amd.build] libawq_cdna4.so built in {time.time() - t0:.1f}s ({len(jobs)} of {len(HIP_SOURCES)} objects recompiled)")
    return LIB_PATH


def build_ext(force: bool = False, verbose: bool = True) -> str:
    import torch
    from torch.utils import cpp_extension as cpp

    src = os.path.join(CSRC, "torch_binding.cpp")
    deps = [src, os.path.join(ROOT, "include", "awq_cdna4.h")]
    if force or _newer(EXT_PATH, deps):
        build_lib(verbose=verbose)
        os.makedirs(EXT_DIR, exist_ok=True)
        tlib = cpp.library_paths()[0]
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
               "-DHIPBLAS_V2", "-DTORCH_EXTENSION_NAME=awq_inference_engine", "-DTORCH_API_INCLUDE_EXTENSION_H",
               f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-Wno-deprecated-declarations"]
        for inc in cpp.include_paths() + ["/opt/rocm/include", sysconfig.get_paths()["include"]]:
            cmd += ["-isystem", inc]
        cmd += [src, "-o", EXT_PATH, "-L" + tlib, "-L" + LIB_DIR, "-lawq_cdna4", "-lc10", "-lc10_hip", "-ltorch_cpu",
                "-ltorch_hip", "-ltorch", "-ltorch_python", "-L/opt/rocm/lib", "-lamdhip64",
                "-Wl,-rpath,$ORIGIN/../lib", "-Wl,-rpath," + tlib]
        dt = _run(cmd, "awq_inference_engine extension")
        if verbose:
            print(f"[llm_awq_amd.build] awq_inference_engine extension built in {dt:.1f}s")
    return EXT_PATH


def build_all(force: bool = False, verbose: bool = True):
    return build_lib(force, verbose), build_ext(force, verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
