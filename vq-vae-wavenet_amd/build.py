"""Build libvqwave.so (HIP, gfx950 only) in-tree with hipcc.

    python vq-vae-wavenet_amd/build.py [--force]

The library lands in vq-vae-wavenet_amd/lib/libvqwave.so (git-ignored, but it travels to the
GPU box with the gpurun snapshot).  hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, 'build' + os.environ.get('VQW_BUILD_SUFFIX', ''))
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, os.environ.get('VQW_LIB_NAME', 'libvqwave.so'))
SOURCES = ['common', 'conv_gemm', 'wgrad_gemm', 'pointwise', 'cond_proj', 'vq', 'wrappers', 'ar_decode', 'ar_persist', 'gate_f16x3']
# -pragma-unroll-threshold: the fully unrolled epilogues over 256 accumulator registers are longer than LLVM's default bound for
# `#pragma unroll` (16 k instructions); a loop it leaves rolled indexes the accumulators dynamically, i.e. puts them in scratch
FLAGS = ['-O3', '--offload-arch=gfx950', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function', '-mllvm', '-pragma-unroll-threshold=65536'] + \
    os.environ.get('VQW_EXTRA_FLAGS', '').split()


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    hdrs.append(os.path.join(HERE, '..', 'include', 'vqwave.h'))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(name, force, hdr_m):
    src = os.path.join(CSRC, name + '.hip')
    obj = os.path.join(OBJ, name + '.o')
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(src)
            and os.path.getmtime(obj) >= hdr_m):
        return obj
    cmd = [_hipcc()] + FLAGS + ['-c', src, '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed for %s:\n%s' % (name, r.stderr[-4000:]))
    return obj


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    hdr_m = _deps_mtime()
    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(lambda n: _compile(n, force, hdr_m), SOURCES))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n' + r.stderr[-4000:])
        if verbose:
            print('built', LIB)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
