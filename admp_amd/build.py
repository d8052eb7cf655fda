"""Build libadmp_hip.so (HIP kernels + C ABI) in-tree for gfx950.

    python -m admp_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting admp_amd/lib/libadmp_hip.so is what the
package loads through ctypes (admp_amd/_lib.py).  No other artefact is produced.
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(CSRC, '_obj')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libadmp_hip.so')
SOURCES = ['engine.hip', 'pair_kernels.hip', 'recip_kernels.hip', 'atom_kernels.hip', 'nbr_kernels.hip', 'cell_kernels.hip', 'dft_kernels.hip', 'pfa_kernels.hip', 'fftx_kernels.hip', 'slab_kernels.hip', 'disp_kernels.hip', 'rccl_comm.hip', 'md_kernels.hip']
ARCH = 'gfx950'
FLAGS = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-munsafe-fp-atomics', '-Wno-unused-result']
# -fno-slp-vectorize: hipcc's SLP vectoriser packs scalar f32 arithmetic into v_pk_*_f32 pairs, whose operands must sit in
# adjacent registers.  In the pair kernel 296 of 1076 VALU instructions became v_mov and 205 VGPRs were needed
# (2 waves/SIMD); without it: 928 VALU, 127 VGPRs (4 waves/SIMD), 0.874 -> 0.496 ms on 1M atoms.  The gather and the
# closing kernel gain as well (152 -> 103, 136 -> 100 VGPRs).
FLAGS.append('-fno-slp-vectorize')
FLAGS += os.environ.get('ADMP_EXTRA_FLAGS', '').split()      # experiments: e.g. -DADMP_SITE_ALIGN=128
EXTRA_FLAGS = {}

def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    hs.append(os.path.join(os.path.dirname(HERE), 'include', 'admp_hip.h'))
    return hs


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src):
    obj = os.path.join(OBJ, src.replace('.hip', '.o'))
    path = os.path.join(CSRC, src)
    if _stale(obj, [path, os.path.abspath(__file__)] + _headers()):
        cmd = ['hipcc'] + FLAGS + EXTRA_FLAGS.get(src, []) + ['-c', path, '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed for %s:\n%s' % (src, r.stderr[-4000:]))
    return obj


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(_compile, SOURCES))
    if force or _stale(LIB, objs):
        rocm = os.environ.get('ROCM_PATH', '/opt/rocm')
        cmd = ['hipcc', '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB] + objs + \
              ['-L' + os.path.join(rocm, 'lib'), '-lrocfft', '-lhiprtc', '-ldl', '-Wl,-rpath,' + os.path.join(rocm, 'lib')]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n%s' % r.stderr[-4000:])
    if verbose:
        print('built', LIB)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv, verbose=True)
