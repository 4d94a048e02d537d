"""The data-layout contract of csrc/conv_wino2s.hip without a GPU: tools/wino2s_emulate.py restates every index expression of
the kernel (DMA slot decode, raw planes, transform task -> V bytes, fragment addresses, accumulator lanes, staging tile, output
pass, weight fragment order) on numpy arrays and must reproduce a float64 convolution; the plane geometry the host picks must
make the transform's ds_read_b128 groups conflict-free for the block shapes of the generator's layers."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import wino2s_emulate as em  # noqa: E402


@pytest.mark.parametrize("N,H,W,cin,cout,res,act,block", [
    (2, 6, 6, 16, 64, False, "relu", None),          # one chunk (odd count: a zero chunk follows), 3x3 tiles
    (3, 5, 7, 32, 64, True, "leaky", None),          # odd extents: ragged tiles, masked stores
    (1, 16, 16, 16, 128, False, "none", (8, 8, 1)),  # the 96x96 layers' block, two cout tiles
    (5, 8, 8, 16, 64, True, "relu", (4, 4, 4)),      # the 24x24 layers' block, an image group running past the batch
    (2, 4, 16, 48, 64, False, "relu", (2, 8, 4)),    # three chunks
    (9, 3, 3, 16, 64, False, "relu", (2, 2, 12)),
    (3, 2, 9, 16, 64, False, "relu", (1, 4, 12)),
])
def test_emulated_kernel_reproduces_the_convolution(N, H, W, cin, cout, res, act, block):
    r = np.random.default_rng(N * 100 + H)
    x = r.standard_normal((N, H, W, cin)).astype(np.float32)
    w = (r.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
    scale, shift = r.uniform(0.5, 1.5, cout).astype(np.float32), (r.standard_normal(cout) * 0.1).astype(np.float32)
    rs = r.standard_normal((N, H, W, cout)).astype(np.float32) if res else None
    got = em.emulate(x, w, scale, shift, rs, act, block)
    ref = em.reference(x, w, scale, shift, rs, act)
    assert not np.isnan(got).any(), "an output pixel was never written"
    assert np.abs(got - ref).max() <= 5e-6, np.abs(got - ref).max()     # fp32 roundings of U and of the transform only


def test_plane_geometry_is_conflict_free_for_the_generator_blocks():
    """96x96 / 48x48 layers run 8x8x1 blocks, 24x24 layers 4x4x4: every 16-lane group of a transform read must hit 16 distinct
    16-byte slots (cost = padding only, < 64 = the price of one collision)"""
    for N, TH, TW, want in ((128, 48, 48, (8, 8, 1)), (128, 24, 24, (8, 8, 1)), (128, 12, 12, (4, 4, 4))):
        b = em.pick_block(N, TH, TW)
        assert b == want, (b, want)
        p, is_ = em.plane_geom(*b)
        assert em.conflict_cost(b[0], b[1], b[2], p, is_) == 0, (b, p, is_)
        assert b[2] * is_ <= em.CELLS and p >= b[1] + 1 and is_ >= (2 * b[0] + 2) * p
    for b in em.BLOCKS:                                   # every candidate that fits has a geometry inside the plane
        if em.block_fits(b):
            p, is_ = em.plane_geom(*b)
            assert b[2] * is_ <= em.CELLS, b


@pytest.mark.parametrize("args", [(2, 5, 6, 32, 64, 4, 6, 5), (3, 3, 3, 16, 128, 3, 3, 14), (1, 9, 7, 48, 64, 8, 8, 2),
                                  (5, 1, 1, 16, 64, 1, 1, 64)])
def test_emulated_fused_phase_split_kernel_reproduces_the_transposed_convolution(args):
    """conv_tp2s.hip's index arithmetic (tools/tp2s_emulate.py): raw slots -> LDS planes, shifted A fragments, the weight-fragment
    order of tp2_pack -> tp2s_pack, tap sequence / phases / shifts, accumulators -> output pixels; ragged blocks, image groups
    running past the batch, three 16-channel chunks"""
    import tp2s_emulate
    err, scale = tp2s_emulate.run(*args)
    assert err <= 1e-12 * scale


@pytest.mark.parametrize("args", [(2, 9, 7, 32, 4, 8, 6), (1, 17, 16, 16, 16, 16, 1), (3, 1, 1, 16, 1, 1, 42)])
def test_emulated_direct_3x3_split_kernel_reproduces_the_convolution(args):
    """conv_k3s.hip's index arithmetic (tools/tp2s_emulate.run_k3s): three raw slots per thread -> planes, nine taps = nine pixel
    shifts under the one-pixel halo, 256 rows over four waves, the weight-fragment order; ragged blocks, image groups past the batch"""
    import tp2s_emulate
    err, scale = tp2s_emulate.run_k3s(*args)
    assert err <= 1e-12 * scale


@pytest.mark.parametrize("args", [(2, 20, 35, 6), (1, 5, 3, 8), (1, 16, 16, 5)])
def test_emulated_first_layer_split_kernel_reproduces_the_convolution(args):
    """conv_stem7s.hip's index arithmetic (tools/tp2s_emulate.run_stem7s): the 22 x 22 region slots, chunks of four taps on the four
    k-groups of the 16x16x32 MFMA (tap / 7 by multiply-shift, taps 49 .. 51 with zero weights), block rows per wave, the weight order"""
    import tp2s_emulate
    err, scale = tp2s_emulate.run_stem7s(*args)
    assert err <= 1e-12 * scale
