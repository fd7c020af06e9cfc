"""Second opinions that do not come from the oracle's author reading OpenCV: definitional / library
implementations of the published algorithms, compared with the oracle's restatements of the OpenCV 3.3.1 primitives.
OpenCV itself is absent from this image, so these are the only non-self evidence available for the OpenCV-backed stages;
they pin the ALGORITHM (FAST-9/16 criterion and score, 3x3 Sobel, bilinear resampling geometry, Gaussian weights), not
OpenCV's bit-level conventions (fixed-point widths, rounding), which tools/pin/run_pin.sh checks where OpenCV exists."""
import numpy as np
import pytest
from scipy import ndimage

# Bresenham circle of radius 3, clockwise from 12 o'clock (Rosten & Drummond 2006, fig. 1)
RING = [(0, -3), (1, -3), (2, -2), (3, -1), (3, 0), (3, 1), (2, 2), (1, 3), (0, 3), (-1, 3), (-2, 2), (-3, 1), (-3, 0), (-3, -1),
        (-2, -2), (-1, -3)]


def fast9_score_map(img, floor=-1):
    """Definition: p is a corner at threshold t iff 9 contiguous ring pixels are all > p + t or all < p - t.  The score of p is the
    largest t for which that holds (-1: not even at t = 0), i.e. max over the 16 arcs of the arc's minimum |difference|, minus 1."""
    I = img.astype(np.int32)
    h, w = I.shape
    c = I[3:h - 3, 3:w - 3]
    d = np.stack([I[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] - c for dx, dy in RING])     # (16, h-6, w-6)
    best = np.full(c.shape, floor, np.int32)
    for sgn in (1, -1):
        dd = sgn * d
        for s in range(16):
            arc = np.minimum.reduce([dd[(s + k) % 16] for k in range(9)])
            best = np.maximum(best, arc - 1)
    out = np.full((h, w), -1, np.int32)
    out[3:h - 3, 3:w - 3] = best
    return out


def fast9_nms(img, th):
    """cv::FAST(img, th, nonmaxSuppression=true) by definition: corners are the pixels with score >= th; one is kept iff its
    score is greater than the score of each of its 8 neighbours (a non-corner counts as 0)."""
    S = fast9_score_map(img)
    C = np.where(S >= th, S, 0)
    P = np.pad(C, 1)
    keep = C > 0
    h, w = C.shape
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            if dx or dy:
                keep &= C > P[1 + dy:1 + dy + h, 1 + dx:1 + dx + w]
    ys, xs = np.nonzero(keep)
    return np.stack([xs, ys, C[ys, xs]], axis=1).astype(np.float32)      # raster order


@pytest.mark.parametrize("seed,th", [(0, 20), (1, 7), (2, 20), (3, 7), (4, 1), (5, 40)])
def test_fast_detector_equals_the_definition(oracle, seed, th):
    from pli_slam_amd import synth
    rng = np.random.default_rng(seed)
    if seed % 2 == 0:
        img = synth.make_stereo_pair(seed, 376, 240)[0]
    else:       # noise + blocks: dense corners, many ties in the non-maximum suppression
        img = rng.integers(0, 256, (97, 131), dtype=np.uint8)
        img[20:60, 30:90] = (img[20:60, 30:90] // 64) * 64
    got = oracle.fast_image(img, th)
    want = fast9_nms(img, th)
    assert len(got) == len(want) > 20
    assert np.array_equal(got, want)


def test_fast_arc_value_equals_the_definition_everywhere(oracle):
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (40, 40), dtype=np.uint8)
    S = fast9_score_map(img, floor=-1000)
    for y in range(3, 37):
        for x in range(3, 37):
            v = oracle.fast_arc(img, x, y)                 # the oracle's arc value is the best arc minimum itself: score + 1 (negative: no arc)
            assert v == S[y, x] + 1, (x, y, v, S[y, x])


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_sobel_equals_scipy(oracle, seed):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (57, 83), dtype=np.uint8)
    dx, dy = oracle.sobel(img)
    I = img.astype(np.int32)
    # cv::Sobel(ksize 3, BORDER_DEFAULT = REFLECT_101) = scipy's 'mirror'
    assert np.array_equal(dx, ndimage.sobel(I, axis=1, mode="mirror").astype(np.int16))
    assert np.array_equal(dy, ndimage.sobel(I, axis=0, mode="mirror").astype(np.int16))


def test_gaussian_blur_is_the_separable_filter_with_8_bit_weights(oracle):
    """The u8 Gaussian: cv::getGaussianKernel's weights exp(-x^2 / 2 sigma^2) / sum, each rounded to 8 fractional bits ON ITS OWN (the
    filter engine's fixed point: the integer kernel may sum to 257), REFLECT_101 borders, one rounding at the end.  scipy's separable
    correlation with those weights agrees to the final rounding; against the exact weights the result is off by that gain only."""
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (64, 80), dtype=np.uint8)
    for n, sigma in ((7, 2.0), (5, 1.0)):
        x = np.arange(n) - (n - 1) / 2
        k = np.exp(-x * x / (2 * sigma * sigma))
        k /= k.sum()
        k8 = np.rint(k * 256) / 256
        got = oracle.gaussian_blur(img, n, sigma).astype(np.float64)
        ref8 = ndimage.correlate1d(ndimage.correlate1d(img.astype(np.float64), k8, axis=1, mode="mirror"), k8, axis=0, mode="mirror")
        assert np.abs(got - ref8).max() <= 0.5 + 1e-9
        ref = ndimage.correlate1d(ndimage.correlate1d(img.astype(np.float64), k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
        assert np.abs(got - ref * k8.sum() ** 2).max() <= 1.5


def test_resize_is_pixel_centre_aligned_bilinear(oracle):
    """cv::resize INTER_LINEAR: source coordinate (d + 0.5) * scale - 0.5, clamped; the 11-bit fixed-point form stays within one
    grey level of the real-valued interpolation."""
    rng = np.random.default_rng(6)
    img = ndimage.gaussian_filter(rng.random((60, 90)) * 255, 1.5)
    img8 = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    for dw, dh in ((75, 50), (108, 72)):
        sx, sy = 90 / dw, 60 / dh
        ys = np.clip((np.arange(dh) + 0.5) * sy - 0.5, 0, 59)
        xs = np.clip((np.arange(dw) + 0.5) * sx - 0.5, 0, 89)
        ref = ndimage.map_coordinates(img8.astype(np.float64), np.meshgrid(ys, xs, indexing="ij"), order=1, mode="nearest")
        got = oracle.resize(img8, dw, dh, sx, sy).astype(np.float64)
        assert np.abs(got - ref).max() <= 1.0


def test_fast_atan2_within_its_published_precision(oracle):
    """cv::fastAtan2: degrees in [0, 360), accuracy about 0.3 degrees (OpenCV documentation)."""
    rng = np.random.default_rng(8)
    v = rng.normal(size=(4000, 2)).astype(np.float32)
    for y, x in v[:600]:
        a = oracle.fast_atan2(float(y), float(x))
        ref = np.degrees(np.arctan2(float(y), float(x))) % 360.0
        d = abs(a - ref)
        assert min(d, 360 - d) <= 0.3
