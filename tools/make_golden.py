"""Generates the committed golden fixtures under tests/golden/ from the CPU oracle.

The reference's own goldens (opencv_extra: RubberWhale, tvl1_flow.flo, aloe-disp*.png) are not
available in this environment and the reference cannot be built (no OpenCV core), so these
fixtures pin the ORACLE (regression) and give the GPU tests a file-based target; they are not
outputs of the reference binary.  Re-run: python tools/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from opencv_contrib_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def tvl1():
    cases = {
        # name: (h, w, dtype, params)
        "tvl1_f32_96x128_it10": (96, 128, "f32", dict(iterations=10, epsilon=0.0)),
        "tvl1_u8_96x128_it10": (96, 128, "u8", dict(iterations=10, epsilon=0.0)),
        "tvl1_f32_77x101_eps": (77, 101, "f32", dict(iterations=300, epsilon=0.01)),
        "tvl1_f32_96x128_it10_cuda": (96, 128, "f32", dict(iterations=10, epsilon=0.0, semantics=1)),
    }
    for name, (h, w, dt, kw) in cases.items():
        I0, I1, gt = synth.flow_pair(h, w, seed=1234, dtype=dt)
        flow, st = O.tvl1_calc(I0, I1, O.tvl1_params(**kw), return_stats=True)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), I0=I0, I1=I1, flow=flow.astype(np.float32),
                            iters=np.array(st["iters"], np.int32), params=np.array(json.dumps(kw)))
        print(name, "EPE vs analytic truth", synth.epe(flow, gt), st["iters"][0])
        if name == "tvl1_f32_96x128_it10":   # the same field in the reference's on-disk golden format (optflow/test/test_tvl1optflow.cpp:49-107)
            from opencv_contrib_amd import flowio
            flowio.writeOpticalFlow(os.path.join(OUT, name + ".flo"), flow.astype(np.float32))


def stereobm():
    left, right, _ = synth.stereo_pair(96, 224, seed=42, max_disp=30)
    for name, kw in {"sbm_96x224_nd64_bs15": dict(num_disparities=64, block_size=15),
                     "sbm_96x224_nd64_bs9_xsobel_uniq": dict(num_disparities=64, block_size=9, prefilter_type=1, uniqueness_ratio=10),
                     "sbm_96x224_nd32_bs11_norm_noedge": dict(num_disparities=32, block_size=11, prefilter_type=0, emulate_edge=0,
                                                              texture_threshold=0.0)}.items():
        disp = O.sbm_compute(left, right, O.sbm_params(**kw))
        np.savez_compressed(os.path.join(OUT, name + ".npz"), left=left, right=right, disp=disp, params=np.array(json.dumps(kw)))
        print(name, "nonzero", float((disp > 0).mean()))


def farneback():
    I0, I1, gt = synth.flow_pair(120, 160, seed=1234, dtype="u8")
    for name, kw in {"fb_u8_120x160_defaults": dict(), "fb_u8_120x160_gauss_poly7": dict(flags=256, poly_n=7, poly_sigma=1.5),
                     "fb_u8_120x160_fastpyr": dict(fast_pyramids=1, num_levels=3)}.items():
        flow = O.fb_calc(I0, I1, O.fb_params(**kw))
        np.savez_compressed(os.path.join(OUT, name + ".npz"), I0=I0, I1=I1, flow=flow.astype(np.float32), params=np.array(json.dumps(kw)))
        print(name, "EPE vs analytic truth", synth.epe(flow[20:-20, 20:-20], gt[20:-20, 20:-20]))


def surf():
    img = synth.blob_image(160, 200, seed=7)
    for name, kw in {"surf_160x200_thr300": dict(hessian_threshold=300.0, n_octaves=3, keypoints_ratio=0.05),
                     "surf_160x200_thr300_ext_upright": dict(hessian_threshold=300.0, n_octaves=3, keypoints_ratio=0.05, extended=1, upright=1)}.items():
        r = O.surf_detect_describe(img, O.surf_params(**kw))
        # these two fixtures ARE outputs of the reference: the reference's own host class over its own surf.cu, executed here
        # (oracle/_ref/libref_cu.so, oracle/Makefile.ref), gives the same keypoints and descriptors bit for bit (as a set: it appends
        # through atomicInc); the file keeps the oracle's deterministic scan order
        from oracle import refcu
        if refcu.available():
            ref = refcu.cuda_class_surf(img, **{k: (bool(v) if k in ("extended", "upright") else v) for k, v in kw.items()})
            order = np.lexsort((r["size"], r["x"], r["y"], r["octave"]))
            assert ref["n"] == r["n"]
            for k in ("x", "y", "laplacian", "octave", "size", "angle", "hessian", "descriptors"):
                assert np.array_equal(r[k][order], ref[k]), k
            print(name, "== the reference CUDA class run on the host")
        np.savez_compressed(os.path.join(OUT, name + ".npz"), img=img, x=r["x"], y=r["y"], laplacian=r["laplacian"], octave=r["octave"],
                            size=r["size"], angle=r["angle"], hessian=r["hessian"], descriptors=r["descriptors"].astype(np.float32),
                            params=np.array(json.dumps(kw)))
        print(name, "features", r["n"])


def stereo_next():
    """StereoSGM and DisparityBilateralFilter goldens on the StereoBM pair."""
    left, right, _ = synth.stereo_pair(96, 224, seed=42, max_disp=30)
    for name, kw in {"sgm_96x224_nd64_hh4": dict(num_disparities=64, mode=3),
                     "sgm_96x224_nd128_hh_min2_clean": dict(num_disparities=128, mode=1, min_disparity=2, emulate_quirks=0)}.items():
        disp = O.sgm_compute(left, right, O.sgm_params(**kw))
        np.savez_compressed(os.path.join(OUT, name + ".npz"), left=left, right=right, disp=disp, params=np.array(json.dumps(kw)))
        print(name, "valid", float((disp >= 0).mean()))
    d = O.sbm_compute(left, right, O.sbm_params(num_disparities=64, block_size=15))
    kw = dict(ndisp=64, radius=4, iters=2)
    out = O.dbf_apply(d, left, O.dbf_params(**kw))
    np.savez_compressed(os.path.join(OUT, "dbf_96x224_nd64_r4_it2.npz"), disp=d, img=left, out=out, params=np.array(json.dumps(kw)))
    print("dbf refined", int((out != d).sum()))


def bfmatch():
    rng = np.random.default_rng(99)
    q = rng.standard_normal((60, 64)).astype(np.float32); t = rng.standard_normal((90, 64)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True); t /= np.linalg.norm(t, axis=1, keepdims=True)
    idx, dist = O.bf_knn_match2(q, t)
    np.savez_compressed(os.path.join(OUT, "bf_60x90x64.npz"), query=q, train=t, idx=idx, dist=dist, params=np.array(json.dumps({})))


def late_additions():
    """CPU-class SURF, sparse PyrLK, integer-descriptor matcher (written at the end of round 1)."""
    img = synth.blob_image(160, 200, seed=7)
    kp = O.surfcpu_detect(img, 300.0, 3, 2)
    k2, d2 = O.surfcpu_compute(img, kp, True, False)
    # round 4: keypoints and descriptors of this fixture are the output of the REFERENCE'S OWN class (oracle/_ref/libref_surfcpu.so =
    # xfeatures2d/src/surf.cpp compiled verbatim), which the restatement equals bit for bit
    from oracle import refocl
    rk, rd = refocl.surfcpu_detect_and_compute(img, 300.0, 3, 2, extended=True, upright=False)
    assert np.array_equal(rk, k2[k2[:, 2] > 0]) and np.array_equal(rd, d2[k2[:, 2] > 0])
    np.savez_compressed(os.path.join(OUT, "surfcpu_160x200_thr300_ext.npz"), img=img, detected=kp, keypoints=k2, descriptors=d2,
                        params=np.array(json.dumps(dict(hessian_threshold=300.0, n_octaves=3, n_octave_layers=2, extended=True, upright=False))))
    I0, I1, _ = synth.flow_pair(120, 160, seed=1234, dtype="u8")
    pts = np.stack([np.random.default_rng(5).uniform(-5, 165, 200), np.random.default_rng(6).uniform(-5, 125, 200)], 1).astype(np.float32)
    nxt, st, err = O.pyrlk_sparse(I0, I1, pts, (21, 21), 3, 30)
    np.savez_compressed(os.path.join(OUT, "sparselk_u8_120x160_w21_l3.npz"), I0=I0, I1=I1, prev_pts=pts, next_pts=nxt, status=st, err=err,
                        params=np.array(json.dumps(dict(win_size=[21, 21], max_level=3, iters=30))))
    rng = np.random.default_rng(77)
    q = rng.integers(0, 256, (70, 32)).astype(np.uint8)
    trains = [rng.integers(0, 256, (n, 32)).astype(np.uint8) for n in (90, 40)]
    idx, img_, dist = O.bf_knn_match(q, trains, 3, O.NORM_HAMMING)
    np.savez_compressed(os.path.join(OUT, "bfint_u8_hamming_70x130x32.npz"), query=q, train0=trains[0], train1=trains[1], idx=idx, img=img_,
                        dist=dist, params=np.array(json.dumps(dict(norm=6, k=3))))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if "--late" in sys.argv:
        late_additions()
        sys.exit(0)
    stereo_next()
    bfmatch()
    tvl1()
    stereobm()
    farneback()
    surf()
