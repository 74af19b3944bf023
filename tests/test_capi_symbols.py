"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol include/sbev_hip.h
declares (and nothing it does not), the ctypes table matches, and argument validation returns the
documented status codes without ever touching a GPU."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import ROOT
from sparsebev_amd import _lib

HEADER = os.path.join(ROOT, 'include', 'sbev_hip.h')


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(sbev_[a-z0-9_]+)\s*\(', text)))


@pytest.fixture(scope='module')
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        from sparsebev_amd.csrc import build
        build.build()
    return _lib.load()


def test_header_symbols_all_exported(lib):
    syms = declared_symbols()
    assert 'sbev_msmv_fwd' in syms and 'sbev_project_select' in syms
    for s in syms:
        assert hasattr(lib, s), 'libsbev_hip.so does not export %s' % s
    assert sorted(_lib.SIGNATURES) == syms, 'ctypes table and include/sbev_hip.h disagree'


def test_exports_are_plain_c(lib):
    out = subprocess.check_output(['nm', '-D', '--defined-only', _lib.LIB_PATH]).decode()
    exported = sorted(l.split()[-1] for l in out.splitlines() if ' T ' in l and l.split()[-1].startswith('sbev_'))
    assert exported == declared_symbols()          # extern "C": unmangled, exactly the header's set


def test_abi_version_and_device_query(lib):
    assert lib.sbev_abi_version() == 1
    assert lib.sbev_device_count() >= 0


def test_argument_validation_without_gpu(lib):
    L = 4
    feats = (ctypes.c_void_p * L)(1, 1, 1, 1)
    hw = (ctypes.c_int32 * (2 * L))(*([4, 4] * L))
    s = (ctypes.c_int64 * L)(*([16] * L))
    one = ctypes.c_void_p(16)

    def call(L_=L, P=4, C=8, layout=0, Bp=2):
        return lib.sbev_msmv_fwd(feats, hw, L_, 0, Bp, 6, C, 3, P, 1, s, 0, s, C, one, one, one, layout, 1, 1, None)

    assert call(P=33) == -1 and b'num_point exceed limits' in lib.sbev_last_error()   # msmv_sampling.cpp:125
    assert call(L_=6) == -1
    assert call(C=6) == -1
    assert call(layout=7) == -1
    assert call(Bp=0) == 0                                                             # empty input: no launch, OK
    assert lib.sbev_project_select(None, None, 1, 4, 1, 6, 4, 4, 1.0, 1.0, 1e-5, None, None, None, None, None) == -1
    assert lib.sbev_project_select(None, None, 0, 4, 1, 6, 4, 4, 1.0, 1.0, 1e-5, None, None, None, None, None) == 0


def test_product_has_no_cpu_fallback():
    import torch
    from sparsebev_amd import ops
    f = [torch.zeros(1, 6, 2, 2, 4)]
    with pytest.raises(RuntimeError, match='device tensors'):
        ops.msmv_sampling(f, torch.zeros(1, 1, 1, 3), torch.zeros(1, 1, 1, 1))
    # and the product package never imports (or even mentions) the oracle
    for root, _, files in os.walk(os.path.join(ROOT, 'sparsebev_amd')):
        for fn in files:
            if fn.endswith('.py'):
                text = open(os.path.join(root, fn)).read()
                assert 'import oracle' not in text and 'from oracle' not in text and 'oracle.' not in text, fn


def test_splitk_plan_is_a_pure_host_function(lib):
    """sbev_linear_splitk_plan: K splits that fill the GPU once.  Config 2's out-projection (900 x 256 x 32768) runs as
    19 row groups x 2 column groups x 26 splits = 988 wave tasks on the 1024 one-wave-per-SIMD slots."""
    plan = lib.sbev_linear_splitk_plan
    assert plan(900, 256, 32768) == 26
    assert plan(3600, 256, 32768) == 6 and plan(3200, 256, 32768) == 7          # configs 4 and 3 (75 / 67 row groups)
    assert plan(36, 256, 32768) == 64                                           # capped: at least 512 of K per split
    assert plan(900, 250, 32768) >= 1 and plan(900, 256, 1000) == 1             # shapes of the generic tile kernel / tiny K
    assert plan(0, 256, 32768) == 1 and plan(900, 0, 32768) == 1
    assert lib.sbev_linear_splitk_workspace(900, 256, 26) == 900 * 256 * 26 * 4


def test_sampler_refuses_a_slab_beyond_the_32bit_in_slab_offset():
    """Argument validation only (returns before any HIP call, so it runs without a GPU): the forward sampler keeps a
    tap's offset inside one sample-batch slab in 32 bits; a level whose slab does not fit must be refused, not mis-read."""
    import ctypes
    from sparsebev_amd import _lib
    lib = _lib.load()
    fake = ctypes.c_void_p(0x1000)
    feats = (ctypes.c_void_p * 1)(fake)
    for H, W, ok_expected in ((1024, 1024, True), (8192, 8192, False)):
        C = 64
        hw = (ctypes.c_int32 * 2)(H, W)
        sbo = (ctypes.c_int64 * 1)(6 * H * W * C)
        sv = (ctypes.c_int64 * 1)(H * W * C)
        # Bp = 0: a valid call returns OK before touching the device; an invalid one must fail in validation first
        st = lib.sbev_msmv_fwd(feats, hw, 1, 0, 0, 6, C, 4, 4, 1, sbo, 0, sv, C, fake, fake, fake, 0, 1, 1, None)
        if ok_expected:
            assert st == 0, lib.sbev_last_error()
        else:
            assert st != 0 and b'32-bit' in lib.sbev_last_error()


def test_transformer_refuses_other_code_sizes():
    import pytest
    from sparsebev_amd.transformer import SparseBEVTransformer
    from sparsebev_amd import synthetic as S
    with pytest.raises(ValueError):
        SparseBEVTransformer(256, num_frames=2, num_levels=4, code_size=11, pc_range=S.PC_RANGE)


def test_round2_entry_points_validate_without_gpu(lib):
    """Argument validation of the training / fused entry points returns the documented status before any HIP call."""
    one = ctypes.c_void_p(16)
    pc = (ctypes.c_double * 6)(-51.2, -51.2, -5, 51.2, 51.2, 3)
    # shape coverage table of the fused gather + mixing launch
    assert lib.sbev_sample_mix_supported(4, 64, 4, 8, 4, 4) == 1 and lib.sbev_sample_mix_supported(5, 64, 4, 16, 4, 4) == 1
    assert lib.sbev_sample_mix_supported(4, 64, 4, 2, 4, 4) == 1          # T*P = 8: padded row tile (round 3)
    assert lib.sbev_sample_mix_supported(4, 64, 8, 8, 4, 4) == 1 and lib.sbev_sample_mix_supported(5, 64, 8, 15, 4, 4) == 1     # P = 8; the 15 x 8 shape
    assert lib.sbev_sample_mix_supported(4, 64, 8, 12, 4, 4) == 0 and lib.sbev_sample_mix_supported(4, 64, 6, 4, 4, 4) == 0    # 96 in_points; P = 6
    assert lib.sbev_sample_mix_supported(4, 32, 4, 8, 4, 4) == 0          # C = 32
    assert lib.sbev_sample_mix_supported(4, 64, 4, 8, 1, 4) == 0          # reference layout (gdiv != G)
    # generic GEMM: empty problems are fine, bad leading dimensions are refused
    assert lib.sbev_gemm_f32(None, 0, 8, None, 1, 8, None, 8, 0, 8, 8, 0, None, None) == 0
    assert lib.sbev_gemm_f32(one, 0, 4, one, 1, 8, one, 8, 4, 8, 8, 0, None, None) == -1 and b'leading dimension' in lib.sbev_last_error()
    assert lib.sbev_gemm_f32_workspace(900, 256, 32768) > 0 and lib.sbev_gemm_f32_workspace(900, 32768, 256) == 0
    assert lib.sbev_gemm_f32_workspace(-1, 4, 4) == -1
    # workspaces
    assert lib.sbev_colsum_workspace(900, 256) == 29 * 256 * 4 and lib.sbev_layer_norm_bwd_workspace(900, 256) == (1800 + 2 * 29 * 256) * 4
    # mixing backward: same shape contract as the forward
    assert lib.sbev_adaptive_mixing_bwd_f32(one, one, one, one, one, 4, 4, 32, 32, 128, 1e-5, None) == -1      # C != 64
    assert lib.sbev_adaptive_mixing_bwd_f32(one, one, one, one, one, 4, 4, 30, 64, 128, 1e-5, None) == -1      # Pin % 4
    assert lib.sbev_adaptive_mixing_bwd_f32(None, None, None, None, None, 0, 4, 32, 64, 128, 1e-5, None) == 0  # empty
    # attention backward: head_dim and dropout range
    assert lib.sbev_sasa_bwd_f32(one, 776, one, pc, None, one, one, one, one, 1, 8, 8, 64, 0.0, 0, None) == -1
    assert lib.sbev_sasa_bwd_f32(one, 776, one, pc, None, one, one, one, one, 1, 8, 8, 32, 1.0, 0, None) == -1
    assert lib.sbev_sasa_bwd_f32(None, 776, None, pc, None, None, None, None, None, 0, 8, 8, 32, 0.1, 0, None) == 0
    # LayerNorm backward: width contract
    assert lib.sbev_layer_norm_bwd(one, one, one, one, 1e-5, 0, one, one, one, one, 4, 6, None) == -1
    # sampler backward: mixing-layout gradient needs B' = B*T*G
    feats = (ctypes.c_void_p * 1)(16)
    hw = (ctypes.c_int32 * 2)(4, 4)
    s64 = (ctypes.c_int64 * 1)(6 * 16 * 64)
    sv = (ctypes.c_int64 * 1)(16 * 64)
    assert lib.sbev_msmv_bwd_ex(feats, None, hw, 1, 6, 6, 64, 3, 4, 1, s64, 0, sv, 64, one, one, one, 1, 4, 4, one, one, None) == -1
    assert lib.sbev_dropout_f32(one, one, 16, 1, 1.0, None) == -1 and lib.sbev_dropout_f32(None, None, 0, 1, 0.5, None) == 0
    assert lib.sbev_copy_widen_f32(one, 3, one, 4, None) == -1


def test_query_order_entry_points_validate_without_gpu(lib):
    """round 4: the launch-order sort and switch (sbev_query_order, sbev_decoder_query_order) check their arguments before any HIP call"""
    one = ctypes.c_void_p(16)
    pc = (ctypes.c_double * 6)(-51.2, -51.2, -5, 51.2, 51.2, 3)
    assert lib.sbev_query_order_max() == 4096
    assert lib.sbev_query_order(None, 10, pc, 0, 900, None, None) == 0 and lib.sbev_query_order(None, 10, pc, 2, 0, None, None) == 0      # empty
    assert lib.sbev_query_order(one, 10, pc, 1, 4097, one, None) == -1 and b'at most 4096' in lib.sbev_last_error()
    assert lib.sbev_query_order(one, 1, pc, 1, 8, one, None) == -1          # a row holds at least the two centre columns
    assert lib.sbev_query_order(None, 10, pc, 1, 8, one, None) == -1 and lib.sbev_query_order(one, 10, None, 1, 8, one, None) == -1
    prev = lib.sbev_decoder_query_order(1)
    assert lib.sbev_decoder_query_order(prev) == 1                           # returns the previous setting
    assert lib.sbev_decoder_query_order(prev) == prev


def test_two_byte_relayout_validates_without_gpu(lib):
    """round 4: sbev_nchw_to_nhwc_b16 / _indirect (bf16 / fp16 storage) check their arguments before any HIP call"""
    one, two = ctypes.c_void_p(16), ctypes.c_void_p(4096)
    assert lib.sbev_nchw_to_nhwc_b16(None, None, 0, 256, 64, None) == 0                       # empty
    assert lib.sbev_nchw_to_nhwc_b16(None, two, 1, 256, 64, None) == -1 and lib.sbev_nchw_to_nhwc_b16(one, None, 1, 256, 64, None) == -1
    assert lib.sbev_nchw_to_nhwc_b16(one, one, 1, 256, 64, None) == -1                        # aliased
    assert lib.sbev_nchw_to_nhwc_b16(one, two, 1, 0, 64, None) == -1 and lib.sbev_nchw_to_nhwc_b16(one, two, 70000, 8, 4, None) == -1
    assert lib.sbev_nchw_to_nhwc_b16_indirect(None, 0, two, 1, 256, 64, None) == -1
    assert lib.sbev_nchw_to_nhwc_b16_indirect(ctypes.c_void_p(12), 0, two, 1, 256, 64, None) == -1      # table not 8-byte aligned
    assert lib.sbev_nchw_to_nhwc_b16_indirect(one, -1, two, 1, 256, 64, None) == -1


def test_round3_training_entry_points_validate_without_gpu(lib):
    """The grouped parameter-gradient launches and the fp16 hi + lo training GEMMs refuse what they do not cover before any HIP call."""
    one = ctypes.c_void_p(16)
    VP = ctypes.c_void_p
    # grad_W on the fp16 kernel: shape coverage (>= 256 tiles of 128 x 128, K >= 32, multiples of 4), 32-bit buffer offsets
    assert lib.sbev_gemm_tn_f16s_ok(256, 32768, 900) == 1 and lib.sbev_gemm_tn_f16s_ok(32768, 256, 900) == 1
    assert lib.sbev_gemm_tn_f16s_ok(256, 256, 900) == 0 and lib.sbev_gemm_tn_f16s_ok(256, 32768, 16) == 0 and lib.sbev_gemm_tn_f16s_ok(258, 32768, 900) == 0
    assert lib.sbev_gemm_tn_f16s(one, 256, one, one, 32768, one, one, 32768, 256, 256, 900, 0, None) == -1               # too few tiles
    assert lib.sbev_gemm_tn_f16s(one, 256, None, one, 32768, one, one, 32768, 256, 32768, 900, 0, None) == -1            # null scale
    assert lib.sbev_gemm_tn_f16s(one, 128, one, one, 32768, one, one, 32768, 256, 32768, 900, 0, None) == -1             # lda < M
    assert lib.sbev_gemm_tn_f16s(one, 256, one, one, 1 << 22, one, one, 32768, 256, 32768, 900, 0, None) == -1 and b'2 GiB' in lib.sbev_last_error()
    assert lib.sbev_f16s_tensor_scale(one, 6, 4, 6, one, None) == -1                                                    # K % 4
    # split-K Linear with the operand scale in device memory: same shape contract as sbev_linear_splitk_f16s, fp32 X only
    assert lib.sbev_linear_splitk_f16s_xdev(one, one, one, one, None, None, None, None, 1e-5, one, 4, 128, 256, 256, 0, 3, one, None) == -1   # N != 256
    assert lib.sbev_linear_splitk_f16s_xdev(one, None, one, one, None, None, None, None, 1e-5, one, 4, 256, 256, 256, 0, 3, one, None) == -1  # null scale
    assert lib.sbev_linear_splitk_f16s_xdev(None, None, None, None, None, None, None, None, 1e-5, None, 0, 256, 256, 256, 0, 3, None, None) == 0
    # mixing backward that also writes the partial maxima
    assert lib.sbev_adaptive_mixing_bwd_max_f32(one, one, one, one, one, None, 4, 4, 32, 64, 128, 1e-5, None) == -1     # null item_max
    assert lib.sbev_adaptive_mixing_bwd_max_f32(None, None, None, None, None, None, 0, 4, 32, 64, 128, 1e-5, None) == 0  # empty
    # grouped reductions: group / segment limits
    i32 = lambda *v: (ctypes.c_int32 * len(v))(*v)
    ptrs = lambda n: (VP * n)(*([16] * n))
    assert lib.sbev_colsum_group(ptrs(17 * 8), ptrs(17), i32(*[4] * 17), i32(*[1] * 17), i32(*[0] * 17), 17, 8, None) == -1      # > 16 groups
    assert lib.sbev_colsum_group(ptrs(8), ptrs(1), i32(4), i32(9), i32(0), 1, 8, None) == -1                                      # > 8 segments
    assert lib.sbev_layer_norm_param_group(ptrs(72), ptrs(72), ptrs(72), ptrs(9), ptrs(9), ptrs(9), ptrs(9), i32(*[4] * 9), i32(*[1] * 9),
                                           i32(*[0] * 9), i32(*[0] * 9), 9, 8, None) == -1                                        # > 8 groups
    assert lib.sbev_layer_norm_bwd_rows(one, one, one, one, 1e-5, 0, one, one, 4, 6, None) == -1                                  # N % 4
    assert lib.sbev_gemm_f32_multi_workspace(256, 256, 900, 6) > 0 and lib.sbev_gemm_f32_multi_workspace(256, 256, 900, 9) == -1


def test_chain_pack_size_is_a_pure_host_function(lib):
    """sbev_decoder_chain_pack_floats: the size of the lane-ordered weight image of the row-chain kernels for the
    reference's layer shape, 0 for shapes the kernels do not cover; the pack call validates its arguments on the host."""
    from sparsebev_amd.runtime import DecoderConfig, DecoderWeights
    cfg = DecoderConfig()
    cfg.B = cfg.Q = 1
    cfg.T, cfg.N, cfg.G, cfg.P, cfg.L = 8, 6, 4, 4, 4
    cfg.D, cfg.H, cfg.ffn, cfg.num_classes, cfg.code_size, cfg.attn_in_rows = 256, 8, 512, 10, 10, 776
    n = lib.sbev_decoder_chain_pack_floats(ctypes.byref(cfg))
    mats = (512 * 256 + 256 * 512 + 5 * 256 * 256 + 2 * 64 * 256      # ffn, cls0 / reg0 / cls3 / reg2 / pe3, cls6 / reg4 (64-column groups)
            + 832 * 256 + 256 * 256 + 128 * 256)                        # attn_in (776 -> 13 groups), attn_out, sampling (112 -> 2 groups)
    assert n > mats and (n - mats) % 64 == 0 and n - mats < 16384         # + the small vectors
    cfg.ffn = 1024
    assert lib.sbev_decoder_chain_pack_floats(ctypes.byref(cfg)) == 0
    cfg.ffn, cfg.code_size = 512, 8
    assert lib.sbev_decoder_chain_pack_floats(ctypes.byref(cfg)) == 0
    w = DecoderWeights()
    assert lib.sbev_decoder_chain_pack(ctypes.byref(cfg), ctypes.byref(w), None, None) == -1                  # SBEV_EINVAL
    assert b'sbev_decoder_chain_pack' in lib.sbev_last_error()
    assert lib.sbev_decoder_row_chain(1) == 0


def test_build_tracks_textual_include_fragments(monkeypatch):
    """msmv_chunk.inc is #included by msmv_sampling.hip AND mixing.hip (the fused gather + mixing kernel): editing it must make
    both objects stale -- the .so ships prebuilt to the GPU box, so a stale object would go unnoticed.  Also
    every #include "..." of every translation unit has to be in the dependency list at all."""
    from sparsebev_amd.csrc import build
    deps = {os.path.basename(d) for d in build.header_deps()}
    assert 'msmv_chunk.inc' in deps and 'sbev_common.hpp' in deps and 'sbev_hip.h' in deps
    for src in build.UNITS:
        text = open(os.path.join(build.HERE, src)).read()
        for inc in re.findall(r'#include\s+"([^"]+)"', text):
            assert os.path.basename(inc) in deps, '%s includes %s, which the build does not track' % (src, inc)
    for hdr in [d for d in build.header_deps() if d.endswith(('.hpp', '.inc'))]:       # headers including headers
        for inc in re.findall(r'#include\s+"([^"]+)"', open(hdr).read()):
            assert os.path.basename(inc) in deps, '%s includes %s, which the build does not track' % (hdr, inc)
    # touch msmv_chunk.inc (virtually: a patched mtime) -> every unit that can include it is stale, so in particular its two users
    real = os.path.getmtime
    chunk = os.path.join(build.HERE, 'msmv_chunk.inc')
    newest = max(real(build.object_path(s)) for s in build.UNITS if os.path.exists(build.object_path(s)))
    monkeypatch.setattr(build.os.path, 'getmtime', lambda f: newest + 10.0 if os.path.abspath(f) == chunk else real(f))
    stale = build.stale_units()
    assert 'msmv_sampling.hip' in stale and 'mixing.hip' in stale
    monkeypatch.undo()
    users = [s for s in build.UNITS if re.search(r'#include\s+"msmv_chunk\.inc"', open(os.path.join(build.HERE, s)).read())]
    assert sorted(users) == ['mixing.hip', 'msmv_sampling.hip']
