"""Second opinion on the three main-repo functions the CPU TV-L1 class calls and this tree can only RESTATE (their source,
modules/imgproc, is not in /root/reference): cv::resize INTER_LINEAR, cv::remap INTER_CUBIC (1/32-px table, a = -0.75, constant 0
border) and cv::medianBlur -- called at optflow/src/tvl1flow.cpp:479-480,522-524,1372-1374,1382-1383.  The reference-class pin
(tests/test_ref_pin.py) forwards exactly these three to oracle/imgproc_ref.c, i.e. it is circular there (missing item 2).
torch and scipy are INDEPENDENT implementations of the same published definitions and are in the image:

  cv::medianBlur 3 / 5 (BORDER_REPLICATE)   == scipy.ndimage.median_filter(mode="nearest")                      exactly
  cv::resize INTER_LINEAR (half-pixel)      == torch.nn.functional.interpolate(bilinear, align_corners=False)   <= 1e-6 relative
                                               (pyramid steps; the zoom back <= 4e-5: float source coordinates, see the test)
  cv::remap INTER_CUBIC, table phases       == torch grid_sample(bicubic, zeros), whose kernel is Keys a = -0.75  <= 1e-5
                                               on maps that sit ON the 1/32-px lattice (the quantisation then changes nothing)
  the 32-entry table                        == the Keys a = -0.75 formula evaluated in float64, <= 1 float ulp, rows sum to 1

What this does NOT prove is bit-level equality with OpenCV's binaries (rounding order inside cv::resize / cv::remap); for the flows
that matters at the 1e-7 level only.  DESIGN.md section 2 records the outcome."""
import numpy as np
import pytest

from oracle import oracle as O

torch = pytest.importorskip("torch")
ndimage = pytest.importorskip("scipy.ndimage")


def rnd(h, w, seed, lo=0.0, hi=1.0):
    return np.random.default_rng(seed).uniform(lo, hi, (h, w)).astype(np.float32)


@pytest.mark.parametrize("ksize", [3, 5])
@pytest.mark.parametrize("shape", [(7, 9), (48, 64), (135, 240), (1, 12), (12, 1)])
def test_median_blur_equals_scipy_median_filter(ksize, shape):
    """optflow/src/tvl1flow.cpp:1382-1383 (medianBlur of u1, u2 before an outer iteration): window median with replicated borders."""
    a = rnd(*shape, seed=ksize * 100 + shape[0])
    a[::3, ::2] = np.round(a[::3, ::2], 1)            # ties
    got = O.median_blur(a, ksize)
    want = ndimage.median_filter(a, size=ksize, mode="nearest")
    assert np.array_equal(got, want)


def test_resize_restatement_equals_torch_bilinear_on_the_five_pyramid_steps():
    """optflow/src/tvl1flow.cpp:479-480 (pyramid, factor 0.8) and :522-524 (flow zoom to the finer scale): INTER_LINEAR with
    half-pixel centres and clamped taps is torch's bilinear interpolate with align_corners=False given the SAME scale."""
    F = torch.nn.functional
    h, w = 270, 480
    a = rnd(h, w, seed=5)
    for step in range(5):
        dh, dw = O.scaled_dim(h, 0.8), O.scaled_dim(w, 0.8)
        got = O.resize_linear_cv(a, fx=0.8, fy=0.8)
        assert got.shape == (dh, dw)
        # cv::resize called with fx, fy uses scale = 1 / fx (not src / dst); tell torch the same scale
        # cv::resize called with fx, fy: destination size = round(src * f) and source coordinate (dx + 0.5) / fx - 0.5 (scale = 1 / fx, not
        # src / dst -- F.interpolate cannot be told both a size and a scale, so the second opinion is torch's bilinear SAMPLER,
        # grid_sample, on those coordinates; padding_mode="border" = cv::resize's clamped taps)
        ys = (np.arange(dh) + 0.5) / 0.8 - 0.5
        xs = (np.arange(dw) + 0.5) / 0.8 - 0.5
        gx = 2.0 * np.clip(xs, 0, w - 1) / (w - 1) - 1.0
        gy = 2.0 * np.clip(ys, 0, h - 1) / (h - 1) - 1.0
        grid = torch.from_numpy(np.stack(np.broadcast_arrays(gx[None, :], gy[:, None]), -1))[None]
        want = F.grid_sample(torch.from_numpy(a)[None, None].double(), grid, mode="bilinear", padding_mode="border", align_corners=True)[0, 0].numpy()
        assert want.shape == got.shape
        err = np.abs(got.astype(np.float64) - want).max() / max(1e-12, np.abs(want).max())
        assert err <= 1e-6, (step, err)
        # ... and the zoom back (dsize given: scale = src / dst), as the flow up-sampling does
        up = O.resize_linear_cv(got, dsize=(w, h))
        wantu = F.interpolate(torch.from_numpy(got)[None, None].double(), size=(h, w), mode="bilinear", align_corners=False)[0, 0].numpy()
        erru = np.abs(up.astype(np.float64) - wantu).max() / max(1e-12, np.abs(wantu).max())
        # cv::resize forms the source coordinate in FLOAT ((float)((dx + 0.5) * scale - 0.5), then floor and fraction in float): at
        # x ~ 480 the fraction carries 2^-24 x 480 = 3e-5 px of rounding, which a white-noise image (unit gradient per pixel) turns
        # into the same relative error; torch interpolates in double here.  The down-scaling steps above happen to be exact to 1e-6.
        assert erru <= 4e-5, (step, erru)
        a, h, w = got, dh, dw


def test_cubic_table_is_keys_minus_three_quarters():
    """The 32 x 4 phase table of cv::remap's INTER_CUBIC (INTER_BITS = 5): Keys' cubic convolution kernel with a = -0.75 at
    distances 1 + f, f, 1 - f, 2 - f."""
    tab = O.cubic_table().reshape(32, 4).astype(np.float64)
    A = -0.75

    def keys(x):
        x = abs(x)
        if x <= 1:
            return (A + 2) * x ** 3 - (A + 3) * x ** 2 + 1
        if x < 2:
            return A * x ** 3 - 5 * A * x ** 2 + 8 * A * x - 4 * A
        return 0.0
    for i in range(32):
        f = i / 32.0
        want = np.array([keys(1 + f), keys(f), keys(1 - f), keys(2 - f)])
        assert np.abs(tab[i] - want).max() <= 2.5e-7, (i, tab[i], want)          # float evaluation, a few ulp of 1
        assert abs(tab[i].sum() - 1.0) <= 1.2e-7
    assert np.array_equal(tab[0], [0.0, 1.0, 0.0, 0.0])


@pytest.mark.parametrize("seed", [1, 2])
def test_remap_restatement_equals_torch_bicubic_grid_sample_on_lattice_maps(seed):
    """optflow/src/tvl1flow.cpp:1372-1374 (I1w, I1wx, I1wy = remap(..., INTER_CUBIC)): on maps whose coordinates are multiples of
    1/32 px the table lookup is the exact kernel, so the restatement must agree with an independent bicubic sampler -- torch's
    grid_sample(mode="bicubic") uses the same a = -0.75 kernel; padding_mode="zeros" is cv::remap's constant-0 border."""
    F = torch.nn.functional
    rng = np.random.default_rng(seed)
    h, w = 60, 90
    src = rnd(h, w, seed=10 + seed)
    # destination coordinates on the 1/32 lattice, some of them outside the image (border path)
    mx = (rng.integers(-3 * 32, (w + 2) * 32, (h, w)) / 32.0).astype(np.float32)
    my = (rng.integers(-3 * 32, (h + 2) * 32, (h, w)) / 32.0).astype(np.float32)
    got = O.remap_cubic_cv(src, mx, my).astype(np.float64)
    # grid_sample, align_corners=True: x_pixel = (g + 1) / 2 * (W - 1)
    gx = 2.0 * mx.astype(np.float64) / (w - 1) - 1.0
    gy = 2.0 * my.astype(np.float64) / (h - 1) - 1.0
    grid = torch.from_numpy(np.stack([gx, gy], -1))[None]
    want = F.grid_sample(torch.from_numpy(src)[None, None].double(), grid, mode="bicubic", padding_mode="zeros", align_corners=True)[0, 0].numpy()
    # cv::remap switches to "all taps skipped => 0" only when the whole 4 x 4 window is outside; torch sums zero-padded taps: same value
    err = np.abs(got - want).max()
    assert err <= 1e-5, err


def test_remap_quantises_the_map_to_one_thirty_second_of_a_pixel():
    """The one property the flows feel: the map is rounded to 1/32 px (round half to even on map * 32) BEFORE sampling, so
    two maps inside the same bin give identical samples, and the sample equals that of the bin's lattice point."""
    h, w = 40, 50
    src = rnd(h, w, seed=3)
    base_x = np.full((h, w), 20.0 + 5 / 32.0, np.float32)
    base_y = np.full((h, w), 17.0 + 9 / 32.0, np.float32)
    a = O.remap_cubic_cv(src, base_x, base_y)
    b = O.remap_cubic_cv(src, base_x + np.float32(0.4 / 32), base_y - np.float32(0.4 / 32))
    c = O.remap_cubic_cv(src, base_x + np.float32(0.6 / 32), base_y)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
