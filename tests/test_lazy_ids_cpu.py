"""CPU check of the arithmetic behind the LAZY ids of round 1 of the tile relaxation (pli_slam_amd/csrc/lsd_tile.hip, `lazyBins` /
`freeMask`; device_prims.hpp `tx_unclaimed_norm_word`): a region decides "which gradient bin is this unclaimed pixel in, relative to
mine" from the pixel's gradient norm in 2^-22 fixed point F = floor(norm * 2^22) and the bin boundaries T(b) = (b * kfix) >> 8 with
kfix = round(maxGrad / (nBins - 1) * 2^30), and only asks the double plane when F lies within `margin` = 8 units of a boundary.
This restates the two formulas in numpy and checks the claim they rest on: OUTSIDE the margin the fixed-point decision equals the
reference's bin = min((int)(norm * (nBins - 1) / maxGrad), nBins - 1) — for every bin, for norms placed adversarially around every
boundary, for several largest norms.  (The kernels themselves are held to the oracle bit for bit by the GPU suite; the test switch
PLI_TX_LAZY_MARGIN sends every pixel through the exact path there.)"""
import numpy as np

MARGIN = 8


def thresholds(max_grad, nbins):
    kfix = np.uint64(int(max_grad / (nbins - 1) * 1073741824.0 + 0.5))
    b = np.arange(nbins + 1, dtype=np.uint64)
    # lazyT: two 24-bit multiplies instead of a 64-bit one — the same value as (b * kfix) >> 8
    t = ((b * (kfix >> np.uint64(12))) << np.uint64(4)) + ((b * (kfix & np.uint64(0xFFF))) >> np.uint64(8))
    assert np.array_equal(t, (b * kfix) >> np.uint64(8))
    assert int(t[-1]) < 2 ** 31 and int(kfix >> np.uint64(12)) < 2 ** 24
    return t.astype(np.int64)


def exact_bin(norm, max_grad, nbins):
    coef = np.float64(nbins - 1) / np.float64(max_grad)
    return np.minimum((norm * coef).astype(np.int64), nbins - 1)


def test_outside_the_margin_the_fixed_point_orders_the_bins_like_the_reference():
    rng = np.random.default_rng(5)
    for nbins in (1024, 256, 129):
        for max_grad in (5.3, 17.0, 99.99, 255.0 * np.sqrt(2.0) / 2, 360.6, 499.0, rng.uniform(6, 400)):
            T = thresholds(max_grad, nbins)
            # norms: a dense cloud around every boundary b / coef (within +-40 fixed-point units), plus uniform ones
            centre = np.arange(1, nbins, dtype=np.float64) * (max_grad / (nbins - 1))
            off = rng.integers(-40, 41, size=(centre.size, 64)).astype(np.float64) / 4194304.0
            norms = np.concatenate([(centre[:, None] + off).ravel(), rng.uniform(0.0, max_grad, 200000), [max_grad, np.nextafter(max_grad, 0)]])
            norms = norms[(norms > 0) & (norms <= max_grad)]
            F = np.minimum(norms * 4194304.0, 2147418112.0).astype(np.int64)          # tx_unclaimed_norm_word (floor)
            bq = exact_bin(norms, max_grad, nbins)
            # every region bin b the pixel could be compared with: take its own, the one below and the one above
            for d in (-1, 0, 1):
                b = np.clip(bq + d, 0, nbins - 1)
                t0 = T[b]
                t1 = np.where(b + 1 > nbins - 1, 0x7FFFFFEF - MARGIN, T[np.minimum(b + 1, nbins)])
                below = F < t0 - MARGIN                      # a lower bin than the region's: the pixel's id is above the region's
                same = (F - (t0 + MARGIN + 1) >= 0) & (F - (t0 + MARGIN + 1) < np.maximum(t1 - MARGIN - (t0 + MARGIN + 1), 0))
                above = F > t1 + MARGIN
                assert not np.any(below & (bq >= b)), (nbins, max_grad, d)
                assert not np.any(same & (bq != b)), (nbins, max_grad, d)
                assert not np.any(above & (bq <= b)), (nbins, max_grad, d)
                ambiguous = ~(below | same | above)
                # ... and the share left to the double plane is what DESIGN.md says: ~(2 * margin + 2) units of a bin's width
                width = max_grad / (nbins - 1) * 4194304.0
                uniform = slice(centre.size * 64, centre.size * 64 + 200000)
                assert ambiguous[uniform].mean() < 4.0 * (2 * MARGIN + 2) / width + 1e-4, (nbins, max_grad, ambiguous[uniform].mean())


def test_the_ranges_the_kernel_and_the_host_refuse():
    """lsd_tile.hip trusts the fixed point only for maxGrad < 500 (no 8-bit image has a larger 2x2 gradient norm: 500 * 2^22 < 2^31) and a bin
    width maxGrad / (nBins - 1) < 3.99 (kfix = width * 2^30 must fit 32 bits); pli_capi.hip picks the lazy form only above 128 bins, which
    implies the second for every admissible norm — found by this file: 17 bins overflowed kfix."""
    assert 500.0 * 4194304.0 < 2 ** 31
    assert 3.99 * 1073741824.0 + 0.5 < 2 ** 32
    assert 500.0 / (129 - 1) < 3.99
    assert (500.0 / 16) * 1073741824.0 > 2 ** 32            # what 17 bins would have needed
