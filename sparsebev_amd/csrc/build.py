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
    'gemm_bf16s.hip': [],                # split-bf16 (bf16x6 / bf16x3) Linears, 8-wave workgroups on v_mfma_f32_32x32x16_bf16
    'gemm_any.hip': [],                  # layout-generic GEMM: grad_x / grad_W of every Linear (training)
    'gemm_tn_f16s.hip': [],              # grad_W of AdaptiveMixing's two big Linears: fp16 hi + lo split on the way into LDS
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


def header_deps():
    """Everything a translation unit can textually include: *.hpp and the *.inc fragments of this directory (msmv_chunk.inc is
    included by both msmv_sampling.hip and mixing.hip), the public header and this build recipe."""
    deps = [os.path.join(HERE, f) for f in sorted(os.listdir(HERE)) if f.endswith(('.hpp', '.inc'))]
    deps.append(os.path.join(os.path.dirname(os.path.dirname(HERE)), 'include', 'sbev_hip.h'))
    deps.append(os.path.abspath(__file__))
    return deps


def object_path(src):
    return os.path.join(OBJ_DIR, src.replace('.hip', '.o'))


def stale_units():
    """Translation units whose object is missing or older than its source or any header / include fragment."""
    headers = header_deps()
    return [src for src in UNITS if _stale(object_path(src), [os.path.join(HERE, src)] + headers)]


def build(force=False, verbose=False, jobs=None):
    hipcc = _hipcc()
    os.makedirs(OBJ_DIR, exist_ok=True)
    todo = list(UNITS) if force else stale_units()
    cmds = []
    for src in todo:
        cmd = [hipcc] + COMMON + UNITS[src] + ['-c', os.path.join(HERE, src), '-o', object_path(src)]
        if verbose:
            print(' '.join(cmd))
        cmds.append(cmd)
    if cmds:      # the units are independent: compile them side by side
        from concurrent.futures import ThreadPoolExecutor
        n = jobs or min(len(cmds), os.cpu_count() or 1, 16)
        with ThreadPoolExecutor(max_workers=n) as pool:
            list(pool.map(subprocess.check_call, cmds))
    objs = [object_path(src) for src in UNITS]
    if force or _stale(LIB, objs):
        cmd = [hipcc, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
