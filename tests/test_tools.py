"""The profile-summary tools are part of the evidence chain (profiles/*.json come out of them): their kernel-name matching is tested."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))


def test_pmc_summary_names_every_row_chain_instantiation():
    """Round 4's summaries lost all three row chains: the pattern expected `row_chain_kernel<d>` / `<d, d>` and the pair tail is
    `row_chain_kernel<0, 2, true>`.  Any template arity, demangled or mangled."""
    import pmc_summary as P
    ns = 'void (anonymous namespace)::'
    assert P.short(ns + 'row_chain_kernel<0, 2, true>((anonymous namespace)::ChainArgs)') == 'row_chain_kernel_tail'
    assert P.short(ns + 'row_chain_kernel<2, 1, false>((anonymous namespace)::ChainArgs)') == 'row_chain_kernel_attention'
    assert P.short(ns + 'row_chain_kernel<1, 1, false>((anonymous namespace)::ChainArgs)') == 'row_chain_kernel_front'
    assert P.short(ns + 'row_chain_kernel<0, 4>(x)') == 'row_chain_kernel_tail'
    assert P.short(ns + 'row_chain_kernel<1>(x)') == 'row_chain_kernel_front'
    assert P.short('_ZN12_GLOBAL__N_116row_chain_kernelILi0ELi2ELb1EEEvNS_9ChainArgsE') == 'row_chain_kernel_tail'
    assert P.short(ns + 'adaptive_mixing_kernel<2, true, 4, float, false>(x)') == 'adaptive_mixing_kernel'
    assert P.short(ns + 'adaptive_mixing_kernel<2, true, 0, float, false>(x)') == 'adaptive_mixing_kernel_plain'
    assert P.short(ns + 'sasa_kernel<false>(x)') == 'sasa_kernel'
    # round 5's two new kernels dropped out of profiles/r5_pmc_*.json ('transpose_tiles_kernel' is not a substring of
    # 'transpose_tiles_multi_kernel'); round 6's on-demand relayout kernels
    assert P.short('(anonymous namespace)::transpose_tiles_multi_kernel((anonymous namespace)::TrMultiArgs)') == 'transpose_tiles_multi_kernel'
    assert P.short(ns + 'transpose_tiles_kernel<true>((anonymous namespace)::TrArgs)') == 'transpose_tiles_kernel'
    assert P.short('(anonymous namespace)::finish_outputs_kernel((anonymous namespace)::FinishArgs)') == 'finish_outputs_kernel'
    assert P.short(ns + 'lazy_tiles_kernel<float>((anonymous namespace)::LazyArgs)') == 'lazy_tiles_kernel'
    assert P.short(ns + 'lazy_scan_kernel<unsigned short>((anonymous namespace)::LazyArgs)') == 'lazy_scan_kernel'
    assert P.short('void at::native::vectorized_elementwise_kernel<4>(x)') is None
