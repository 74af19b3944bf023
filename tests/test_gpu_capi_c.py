"""The C ABI from plain C (tests/capi/capi_demo.c: gcc + HIP runtime API + libsbev_hip.so, C oracle as the checker):
no Python and no torch on the calling side.  The binary is built by __graft_entry__.build() / tests/capi/Makefile."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, 'tests', 'capi', 'build', 'capi_demo')


def _ensure_built():
    if not os.path.exists(EXE):
        subprocess.run(['make', '-C', os.path.join(ROOT, 'tests', 'capi')], check=True, capture_output=True)
    return EXE


def test_c_client_links_against_the_in_tree_library():
    exe = _ensure_built()
    out = subprocess.run(['ldd', exe], capture_output=True, text=True, check=True).stdout
    lib = [l for l in out.splitlines() if 'libsbev_hip.so' in l]
    assert lib and 'not found' not in lib[0]
    assert os.path.realpath(lib[0].split('=>')[1].split('(')[0].strip()) == os.path.realpath(
        os.path.join(ROOT, 'sparsebev_amd', 'csrc', 'libsbev_hip.so'))
    assert 'libtorch' not in out and 'libpython' not in out        # the boundary carries no torch / Python dependency


@pytest.mark.gpu
def test_sampler_called_from_c_matches_the_c_oracle():
    r = subprocess.run([_ensure_built()], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'max |hip - oracle|' in r.stdout and 'L = 6 rejected' in r.stdout and 'P = 33 rejected' in r.stdout
    assert '0 differ from the scalar nan_to_num' in r.stdout and 'pair fault word: clean' in r.stdout          # round 5's entry points
