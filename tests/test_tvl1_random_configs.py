"""Randomised parity sweep of cv::cuda::OpticalFlowDual_TVL1 (HIP through the C-ABI) against the oracle: image sizes that are not
multiples of anything (strip / band / tile edges of the blocked iteration kernel, the 6 x 6 windows of the fused warp, the 8-row
strips of the resize), random pyramid depth, warps, iteration counts on both sides of the kernels' block sizes, scale steps,
both semantics, u8 and f32 input, fixed work and the device-decided stop.  Seeded: the same 40 configurations every run."""
import os

import numpy as np
import pytest

from opencv_contrib_amd import synth

pytestmark = pytest.mark.gpu


def _configs():
    rng = np.random.default_rng(int(os.environ.get("MIFLOW_SWEEP_SEED", "20260923")))
    out = []
    for k in range(int(os.environ.get("MIFLOW_SWEEP_N", "40"))):
        h, w = int(rng.integers(17, 260)), int(rng.integers(17, 330))
        out.append(dict(shape=(h, w), seed=int(rng.integers(1, 10 ** 6)), dtype=("u8", "f32")[int(rng.integers(2))],
                        nscales=int(rng.integers(1, 6)), warps=int(rng.integers(1, 6)), iterations=int(rng.integers(1, 34)),
                        epsilon=float((0.0, 0.0, 0.02, 0.05)[int(rng.integers(4))]), scale_step=float((0.8, 0.5, 0.9)[int(rng.integers(3))]),
                        semantics=int(rng.integers(2)), flow_scale=float(rng.uniform(0.2, 1.5))))
    return out


@pytest.mark.parametrize("cfg", _configs(), ids=lambda c: f"{c['shape'][0]}x{c['shape'][1]}-{c['dtype']}-s{c['nscales']}w{c['warps']}i{c['iterations']}"
                                                          f"-e{c['epsilon']}-sem{c['semantics']}")
def test_random_configuration_matches_oracle(gpu, oracle, cfg):
    import torch
    from opencv_contrib_amd import cuda
    I0, I1, _ = synth.flow_pair(*cfg["shape"], seed=cfg["seed"], dtype=cfg["dtype"], flow_scale=cfg["flow_scale"])
    p = oracle.tvl1_params(iterations=cfg["iterations"], epsilon=cfg["epsilon"], nscales=cfg["nscales"], warps=cfg["warps"],
                           scale_step=cfg["scale_step"], semantics=cfg["semantics"])
    ref, st = oracle.tvl1_calc(I0, I1, p, return_stats=True)
    alg = cuda.OpticalFlowDual_TVL1.create(nscales=cfg["nscales"], warps=cfg["warps"], epsilon=cfg["epsilon"], iterations=cfg["iterations"],
                                           scaleStep=cfg["scale_step"], semantics=cfg["semantics"])
    t0, t1 = torch.from_numpy(I0).to(gpu), torch.from_numpy(I1).to(gpu)
    flow = alg.calc(t0, t1).cpu().numpy()
    assert np.isfinite(flow).all()
    assert alg.getNumScales() == st["nscales"]                       # levels dropped by the 16-px rule, tvl1flow.cpp:243-247
    d = np.sqrt(((flow - ref) ** 2).sum(-1))
    if cfg["epsilon"] == 0.0:
        # fast math against the exact restatement: the stated bound of the default path at the reference test's 10 iterations; the
        # rounding differences of v_rcp / v_sqrt / fma grow with the iteration count (a 300-configuration sweep found 5.8e-3 px at
        # 32 iterations x 5 warps x 4 scales on a 240 x 88 image), so the bound scales with it -- the reference's own acceptance
        # of its CUDA class against the CPU class, |1 - CCORR| <= 4e-3 (test_optflow.cpp:465), is asserted as well
        assert d.mean() <= 5e-3 * max(1.0, cfg["iterations"] / 10.0), d.mean()
        assert synth.ccorr_dissimilarity(flow, ref) <= 4e-3
    else:
        it = np.array(alg.lastIterations())
        rit = np.array(st["iters"])[:it.shape[0], :it.shape[1]]
        assert it.max() <= cfg["iterations"] and it.min() >= 1
        assert np.abs(it - rit).max() <= 2, (it.tolist(), rit.tolist())
        assert d.mean() <= 5e-2, d.mean()
    # a second object, exact math: bit-comparable arithmetic, tight tolerance
    ex = cuda.OpticalFlowDual_TVL1.create(nscales=cfg["nscales"], warps=cfg["warps"], epsilon=0.0, iterations=min(cfg["iterations"], 8),
                                          scaleStep=cfg["scale_step"], semantics=cfg["semantics"], exactMath=True)
    p.outer_iterations = min(cfg["iterations"], 8); p.epsilon = 0.0
    ref2 = oracle.tvl1_calc(I0, I1, p)
    d2 = np.sqrt(((ex.calc(t0, t1).cpu().numpy() - ref2) ** 2).sum(-1))
    assert d2.mean() <= 2e-4, d2.mean()


def _batch_configs():
    rng = np.random.default_rng(77)
    return [dict(shape=(int(rng.integers(20, 200)), int(rng.integers(20, 300))), n=int(rng.integers(2, 10)), lanes=int(rng.integers(0, 5)),
                 iterations=int(rng.integers(1, 25)), epsilon=float((0.0, 0.03)[int(rng.integers(2))]), semantics=int(rng.integers(2)),
                 seed=int(rng.integers(1, 10 ** 6))) for _ in range(10)]


@pytest.mark.parametrize("cfg", _batch_configs(), ids=lambda c: f"{c['shape'][0]}x{c['shape'][1]}-n{c['n']}-l{c['lanes']}-i{c['iterations']}-e{c['epsilon']}")
def test_random_batch_equals_single_calcs(gpu, cfg):
    """calc_batch over random batch sizes and lane counts (sub-batches of unequal size, more lanes than pairs) is bit-identical to
    the single calcs, iteration counts included."""
    import torch
    from opencv_contrib_amd import cuda
    pairs = [synth.flow_pair(*cfg["shape"], seed=cfg["seed"] + k)[:2] for k in range(cfg["n"])]
    I0s, I1s = [torch.from_numpy(p[0]).to(gpu) for p in pairs], [torch.from_numpy(p[1]).to(gpu) for p in pairs]
    kw = dict(iterations=cfg["iterations"], epsilon=cfg["epsilon"], semantics=cfg["semantics"])
    batch = cuda.OpticalFlowDual_TVL1.create(lanes=cfg["lanes"], **kw)
    single = cuda.OpticalFlowDual_TVL1.create(**kw)
    out = batch.calc_batch(I0s, I1s)
    for k in range(cfg["n"]):
        assert torch.equal(out[k], single.calc(I0s[k], I1s[k])), k
        assert batch.lastIterations(k) == single.lastIterations(0), k


@pytest.mark.parametrize("shape", [(16, 16), (37, 101), (64, 9), (5, 300), (150, 211), (240, 427)])
@pytest.mark.parametrize("iters", [1, 2, 3, 4, 5, 6, 7, 10, 13])
def test_exact_math_fused_blocks_bit_identical_to_single_iterations(gpu, shape, iters):
    """exact_math = 1 with fixed work runs blocks of up to five fused iterations (k_iterate_tbr MODE 2: the exact kernel's
    expressions on the rotating-register pipeline); timeBlock = 1 forces one launch per iteration.  Bit-identical flows for both
    semantics, on sizes narrower than a strip, shorter than a band, and with the border columns / rows inside halo lanes."""
    import torch
    from opencv_contrib_amd import cuda
    I0, I1, _ = synth.flow_pair(*shape, seed=shape[1] + iters)
    t0, t1 = torch.from_numpy(I0).to(gpu), torch.from_numpy(I1).to(gpu)
    for sem in (0, 1):
        a = cuda.OpticalFlowDual_TVL1.create(iterations=iters, epsilon=0.0, exactMath=True, semantics=sem)
        b = cuda.OpticalFlowDual_TVL1.create(iterations=iters, epsilon=0.0, exactMath=True, semantics=sem, timeBlock=1)
        fa, fb = a.calc(t0, t1), b.calc(t0, t1)
        assert torch.equal(fa, fb), (sem, float((fa - fb).abs().max()))
