"""Register / scratch budget of the compiled gfx950 kernels, read from the code object inside libfbbev_hip.so
(tools/kernel_resources.py; no GPU needed).  Guards what the measurements were taken with: no kernel spills to scratch,
and the hot kernels keep the register budget their occupancy was tuned for -- a regression here changes the speed of a
kernel without changing a single result."""
import os
import re
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import kernel_resources as KR  # noqa: E402

pytestmark = pytest.mark.skipif(not (os.path.exists(os.path.join(KR.LLVM, 'llvm-readelf')) and shutil.which('objcopy')),
                                reason='llvm-readelf / objcopy not available')


@pytest.fixture(scope='module')
def res():
    from fb_bev_amd import build
    return KR.kernel_resources(build.build())


def _regs(r):
    return r['vgpr'] + r.get('agpr', 0)


def test_no_kernel_uses_scratch_or_spills(res):
    assert len(res) > 150
    bad = {k: v for k, v in res.items() if v.get('scratch', 0) or v.get('vgpr_spills', 0)}
    assert not bad, list(bad)[:5]
    # scalar registers may overflow into lanes of a vector register (v_writelane: no memory traffic) -- a handful at most.
    # the split DA backward kernels (~30 kernel arguments each) park more of their loop-invariant scalars there: one VGPR's worth
    lim = lambda k: 64 if 'k_da_cross_attn_bwd_' in k else 24  # noqa: E731
    over = {k: v.get('sgpr_spills', 0) for k, v in res.items() if v.get('sgpr_spills', 0) > lim(k)}
    assert not over, over


# pattern -> most registers (VGPR + AGPR) a matching kernel may use; 512 / budget = waves per SIMD the tuning assumed
BUDGETS = {
    r'k_pool_fwd_dense2': 80,            # dense bev_pool_v2: 6 waves / SIMD (profiles/r01_occupancy_experiment.txt)
    r'k_pool_zmean': 72,
    r'k_pool_bwd_pixel': 96,
    r'k_pool_bwd_rows': 32,
    r'k_sort_scatterILi4E': 64,           # thin chunks: 8 waves / SIMD
    r'k_sort_scatterILi16E': 128,         # 1024-thread workgroups (4 waves / SIMD is one workgroup: 128 registers each)
    r'k_sort_hist': 64,
    r'k_interval_write': 128,                  # 1024-thread workgroups: 4 waves / SIMD
    r'k_keys_hist_geom': 128,
    r'k_da_cross_attn_fwd_unitILi10E': 168,    # the shipped head dim: 3 waves / SIMD (12 corner loads of a sample in flight)
    r'k_da_cross_attn_bwdILi': 128,            # the global-atomic backward: 4 waves / SIMD
    r'k_da_cross_attn_bwd_scatter': 96,        # value-gradient scatter: 5 waves / SIMD
    r'k_da_cross_attn_bwd_unitILi10E': 224,    # unit-owned gradients at the shipped head dim: 2 waves / SIMD (48 corner registers in flight)
    r'k_history_warp': 168,
    r'k_history_conv_tILi5ELi5E': 512,         # register-resident weights: one wave per SIMD by design (the whole file)
    r'k_history_warp_vm': 128,                 # voxel-major ring: 4+ waves / SIMD (16 sixteen-byte taps in flight per thread)
    r'k_history_conv_bf16ILi5ELi5E': 256,      # bf16-MFMA variant: two waves per SIMD (the next frame's loads need a partner)
    r'k_conv3d_ndhwc': 256,                    # two waves / SIMD: the ping-pong buffers need a partner wave
    r'k_conv3d_wgrad_ndhwc': 256,
    r'k_conv3d_k3_tile_bf16': 256,             # 8 waves per workgroup (4 MFMA + 4 loader): two per SIMD
    r'k_msda_fwd_unitILi10E': 136,
}


@pytest.mark.parametrize('pattern', sorted(BUDGETS))
def test_hot_kernels_keep_their_register_budget(res, pattern):
    hits = {k: _regs(v) for k, v in res.items() if re.search(pattern, k)}
    assert hits, pattern
    over = {k: n for k, n in hits.items() if n > BUDGETS[pattern]}
    assert not over, over
