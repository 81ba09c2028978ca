"""Build libu2mkd_hip.so for gfx950 with hipcc (in-tree; cross-compiles without a GPU).

    python -m u2mkd_amd.build [--force]

The .so lands in u2mkd_amd/lib/ (git-ignored, but it travels to the GPU box
with the repo snapshot).  One object per .hip file, built in parallel.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
OBJDIR = os.path.join(os.path.dirname(HERE), 'build', 'obj')
LIB = os.path.join(LIBDIR, 'libu2mkd_hip.so')
ARCH = 'gfx950'
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['-O3', f'--offload-arch={ARCH}', '-fPIC', '-std=c++17', '-Wall', '-Wno-unused-function']


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    hdrs.append(os.path.join(os.path.dirname(HERE), 'include', 'u2mkd_hip.h'))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src, force):
    obj = os.path.join(OBJDIR, os.path.basename(src)[:-4] + '.o')
    newest = max(os.path.getmtime(src), _deps_mtime())
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= newest:
        return obj, False
    cmd = [HIPCC, *FLAGS, '-c', src, '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'hipcc failed for {src}:\n{r.stdout}\n{r.stderr}')
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(lambda s: _compile(s, force), srcs))
    objs = [o for o, _ in results]
    rebuilt = any(c for _, c in results)
    if rebuilt or force or not os.path.exists(LIB):
        cmd = [HIPCC, f'--offload-arch={ARCH}', '-shared', '-fPIC', '-o', LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
        if verbose:
            print(f'[u2mkd_amd.build] linked {LIB}')
    elif verbose:
        print(f'[u2mkd_amd.build] up to date: {LIB}')
    build_host(force=force or rebuilt, verbose=verbose)
    return LIB


HOST_SRC = os.path.join(HERE, 'csrc_host', 'host_ops.cpp')
HOST_LIB = os.path.join(LIBDIR, '_u2mkd_host.so')


def build_host(force: bool = False, verbose: bool = True) -> str:
    """lib/_u2mkd_host.so: the C++ host side of the hottest operators (csrc_host/host_ops.cpp), a CPython extension over
    libtorch and libu2mkd_hip.so.  Host code only: g++ (no device code in it)."""
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    newest = max(os.path.getmtime(HOST_SRC), os.path.getmtime(os.path.join(os.path.dirname(HERE), 'include', 'u2mkd_hip.h')))
    if not force and os.path.exists(HOST_LIB) and os.path.getmtime(HOST_LIB) >= newest:
        if verbose:
            print(f'[u2mkd_amd.build] up to date: {HOST_LIB}')
        return HOST_LIB
    abi = int(bool(getattr(torch._C, '_GLIBCXX_USE_CXX11_ABI', True)))
    cmd = [os.environ.get('CXX', 'g++'), '-O2', '-std=c++17', '-fPIC', '-shared', '-D__HIP_PLATFORM_AMD__=1', '-DUSE_ROCM=1',
           '-DHIPBLAS_V2', '-DTORCH_EXTENSION_NAME=_u2mkd_host', f'-D_GLIBCXX_USE_CXX11_ABI={abi}', '-DTORCH_API_INCLUDE_EXTENSION_H',
           *[f'-I{p}' for p in ce.include_paths()], '-I/opt/rocm/include', '-I' + sysconfig.get_paths()['include'],
           HOST_SRC, '-o', HOST_LIB, *[f'-L{p}' for p in ce.library_paths()], '-lc10', '-lc10_hip', '-ltorch', '-ltorch_cpu',
           '-ltorch_hip', '-ltorch_python', f'-L{LIBDIR}', '-lu2mkd_hip', '-Wl,-rpath,$ORIGIN']
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'host extension build failed:\n{r.stdout}\n{r.stderr}')
    if verbose:
        print(f'[u2mkd_amd.build] built {HOST_LIB}')
    return HOST_LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
