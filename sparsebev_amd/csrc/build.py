"""Build libsbev_hip.so (gfx950 only) in-tree with hipcc.  No torch extension, no pybind, no cmake:
plain `hipcc -c` per translation unit + one `hipcc -shared` link, so the .so travels with the repo
snapshot to the GPU box.  Usage: python -m sparsebev_amd.csrc.build [--force]"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, 'libsbev_hip.so')
OBJ_DIR = os.path.join(HERE, 'build')
ARCH = 'gfx950'
COMMON = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']
# translation unit -> extra flags
UNITS = {
    'capi_common.hip': [],
    'msmv_sampling.hip': [],
    # hardware float atomics (global_atomic_add_f32) for the grad_value scatter instead of a CAS loop
    'msmv_sampling_bwd.hip': ['-munsafe-fp-atomics'],
    'gemm.hip': [],
    # VGPR-form MFMAs: the kernel fits 256 VGPRs, AGPR-form costs 144 accumulator copies per loop iteration
    'gemm_regtile.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form=1'],
    'gemm_bf16x3.hip': [],
    'gemm_any.hip': [],                  # layout-generic GEMM: grad_x / grad_W of every Linear (training)
    'mixing_bwd.hip': [],
    'attention_bwd.hip': [],
    'attention_bwd_mfma.hip': [],
    'backward_ops.hip': ['-ffp-contract=off'],   # re-runs project.hip's individually rounded projection to re-select the camera
    # no implicit FMA contraction: the fused gather + mixing instantiations and the plain mixing kernel must round identically
    # (contraction decisions are made per instantiation by the backend; explicit fmaf / MFMA are unaffected)
    'mixing.hip': ['-ffp-contract=off'],
    'attention.hip': [],
    'layout.hip': [],
    'head.hip': ['-ffp-contract=off'],   # __fmul_rn / __fadd_rn are plain * and + in HIP: keep them unfused
    'decoder.hip': [],
    'row_chain.hip': [],                 # the row-local op chains of a layer (4x4x1 MFMA with A broadcast)
    # bit-exact projection: no FMA contraction anywhere in this file
    'project.hip': ['-ffp-contract=off'],
}


def _hipcc():
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found (need ROCm >= 7.0 to build libsbev_hip.so for gfx950)')
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = _hipcc()
    os.makedirs(OBJ_DIR, exist_ok=True)
    headers = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith('.hpp')]
    headers.append(os.path.join(os.path.dirname(os.path.dirname(HERE)), 'include', 'sbev_hip.h'))
    headers.append(os.path.abspath(__file__))
    objs = []
    for src, extra in UNITS.items():
        s = os.path.join(HERE, src)
        o = os.path.join(OBJ_DIR, src.replace('.hip', '.o'))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + COMMON + extra + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd))
            subprocess.check_call(cmd)
    if force or _stale(LIB, objs):
        cmd = [hipcc, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
