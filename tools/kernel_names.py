"""Names of the step's kernels as rocprofv3 prints them -> the short names bench.py and profiles/ use.

Template arguments are matched BY POSITION (`chain_kernel<W, Pol, DEG, MODE, RES>`: MODE is the fourth), not by a suffix such
as ', 1>': round 4 added the fifth parameter and the suffix match silently dropped both chain kernels from the counter
summaries (VERDICT r4 weak #5a)."""
import re

FWD_NAME, CHAIN_NAME, FUSED_NAME, INFER_NAME = 'chain_kernel<MODE_FWD_TRAIN>', 'chain_kernel<MODE_CHAIN>', 'bwd128_kernel', 'fused_fwd_kernel (inference)'
_CHAIN = re.compile(r'chain_kernel<\s*(\d+)\s*,\s*([A-Za-z0-9_]+)\s*,\s*(\d+)\s*,\s*(\d+)')
_PLAIN = ('bwd128_kernel', 'dout128_kernel', 'reduce128_stage1', 'reduce128_kernel', 'dw_kernel', 'reduce_kernel', 'rt_kernel', 'adam_dev_kernel',
          'adam_kernel', 'chi2_image_kernel', 'loss_sum_kernel', 'pack_weights_kernel', 'eht_vis_kernel', 'eht_loss_kernel', 'eht_bwd_kernel')


def short_name(kernel_name):
    """Short name of a demangled kernel name, or None for kernels that are not the library's."""
    m = _CHAIN.search(kernel_name)
    if m:
        return {1: FWD_NAME, 2: CHAIN_NAME}.get(int(m.group(4)))
    if 'fused_fwd_kernel' in kernel_name or 'fused_fwd_wide_kernel' in kernel_name:
        return INFER_NAME
    for n in _PLAIN:
        if re.search(r'\b%s\b' % n, kernel_name):
            return n
    return None


def expected_kernels(width):
    """The MLP kernels one training step + one inference render of bench.py must show in every profiler pass."""
    if width == 128:
        return [FWD_NAME, FUSED_NAME, INFER_NAME]
    return [FWD_NAME, CHAIN_NAME, 'dw_kernel', INFER_NAME]


if __name__ == '__main__':
    assert short_name('void chain_kernel<256, PolBF16, 3, 1, false>(BwdArgs)') == FWD_NAME
    assert short_name('void chain_kernel<256, PolBF16, 3, 2, false>(BwdArgs)') == CHAIN_NAME
    assert short_name('void chain_kernel<128, PolBF16, 3, 1, true>(BwdArgs) [clone .kd]') == FWD_NAME
    assert short_name('void chain_kernel<256, PolBF16T8, 3, 2>(BwdArgs)') == CHAIN_NAME
    assert short_name('void dw_kernel<256, PolBF16>(BwdArgs)') == 'dw_kernel'
    assert short_name('void (anonymous namespace)::bwd128_kernel<4, 3>(BwdArgs)') == FUSED_NAME
    assert short_name('void (anonymous namespace)::reduce128_kernel<4>(BwdArgs, int, int)') == 'reduce128_kernel'
    assert short_name('void fused_fwd_kernel<256, PolBF16, 3, true, false, false>(FusedArgs)') == INFER_NAME
    assert short_name('void at::native::vectorized_elementwise_kernel<4, ...>') is None
    print('ok')
