// Source-compatibility smoke test of the C++ drop-in headers: the code below is written the way a caller of
// the reference writes it (cudaoptflow/samples/optical_flow.cpp:184-185, cudastereo/test/test_stereo.cpp:76-80).
// Usage: shim_smoke <in.bin> <out.bin>   (in: int32 h, w; u8 I0[h*w], I1[h*w]; out: f32 flow[h*w*2], u8 disp[h*w])
// Without a GPU it must fail with cv::Exception (no CPU fallback) and return 3.
#include <algorithm>
#include <cstdio>
#include <vector>
#include "opencv2/cudaoptflow.hpp"
#include "opencv2/superres/optical_flow.hpp"
#include "opencv2/cudastereo.hpp"
#include "opencv2/cudafeatures2d.hpp"
#include "opencv2/xfeatures2d/cuda.hpp"

int main(int argc, char **argv)
{
    using namespace cv;
    try {
        Ptr<cuda::OpticalFlowDual_TVL1> tvl1 = cuda::OpticalFlowDual_TVL1::create();
        if (tvl1->getNumIterations() != 300 || tvl1->getDefaultName() != "DenseOpticalFlow.OpticalFlowDual_TVL1") return 2;
        tvl1->setNumIterations(10);
        tvl1->setEpsilon(0.0);
        Ptr<cuda::StereoBM> bm = cuda::createStereoBM(32, 9);
        Ptr<cuda::FarnebackOpticalFlow> fb = cuda::FarnebackOpticalFlow::create();
        if (fb->getWinSize() != 13 || fb->getDefaultName() != "DenseOpticalFlow.FarnebackOpticalFlow") return 2;
        fb->setNumLevels(3);
        cuda::SURF_CUDA surf(300, 3, 2, false, 0.05f);
        if (surf.descriptorSize() != 64 || surf.defaultNorm() != NORM_L2 || cuda::SURF_CUDA().descriptorSize() != 128) return 2;
        if (argc < 3) return 0;
        FILE *f = fopen(argv[1], "rb");
        if (!f) return 4;
        int hw[2];
        if (fread(hw, 4, 2, f) != 2) return 4;
        const int h = hw[0], w = hw[1];
        std::vector<unsigned char> a((size_t)h * w), b((size_t)h * w);
        if (fread(a.data(), 1, a.size(), f) != a.size() || fread(b.data(), 1, b.size(), f) != b.size()) return 4;
        fclose(f);
        cuda::GpuMat d0(h, w, CV_8UC1), d1(h, w, CV_8UC1), flow, disp;
        d0.upload(a.data(), (size_t)w);
        d1.upload(b.data(), (size_t)w);
        cuda::Stream stream;
        tvl1->calc(d0, d1, flow, stream);
        bm->compute(d0, d1, disp, stream);
        cuda::GpuMat fbflow;
        fb->calc(d0, d1, fbflow, stream);
        stream.waitForCompletion();
        if (fbflow.type() != CV_32FC2 || fbflow.size() != d0.size()) return 5;
        if (flow.type() != CV_32FC2 || flow.size() != d0.size() || disp.type() != CV_8UC1) return 5;
        std::vector<float> hf((size_t)h * w * 2);
        std::vector<unsigned char> hd((size_t)h * w);
        flow.download(hf.data(), (size_t)w * 8);
        disp.download(hd.data(), (size_t)w);
        FILE *o = fopen(argv[2], "wb");
        fwrite(hf.data(), 4, hf.size(), o);
        fwrite(hd.data(), 1, hd.size(), o);
        fbflow.download(hf.data(), (size_t)w * 8);
        fwrite(hf.data(), 4, hf.size(), o);
        std::vector<KeyPoint> kps;
        std::vector<float> desc;
        surf(d0, cuda::GpuMat(), kps, desc);
        const int nk = (int)kps.size();
        fwrite(&nk, 4, 1, o);
        for (const KeyPoint &k : kps) { float r[5] = {k.pt.x, k.pt.y, k.size, k.angle, k.response}; fwrite(r, 4, 5, o); }
        fwrite(desc.data(), 4, desc.size(), o);
        if (desc.size() != (size_t)nk * 64) return 7;
        fclose(o);
        {   // frame after frame through the GpuMat overload the buffers are REUSED (a header cut to the feature count must not
            // look too small: that was a hipFree + hipMalloc of up to 33 MB per frame), and a frame without keypoints leaves the caller's
            // descriptors untouched, as the reference's computeDescriptors does (surf.cuda.cpp:227-236)
            cuda::GpuMat kg, dg;
            surf(d0, cuda::GpuMat(), kg, dg);
            if (kg.cols != nk || dg.rows != nk || dg.cols != 64) return 10;
            const void *kp0 = kg.data, *dp0 = dg.data;
            surf(d0, cuda::GpuMat(), kg, dg);
            if (kg.data != kp0 || dg.data != dp0 || kg.cols != nk || dg.rows != nk) return 10;
            std::vector<float> d2((size_t)nk * 64);
            if (nk > 0) { dg.download(d2.data(), 64 * sizeof(float)); if (d2 != desc) return 10; }
            cuda::SURF_CUDA none(1e12, 3, 2, false, 0.05f);   // a threshold nothing passes
            none(d0, cuda::GpuMat(), kg, dg);
            if (kg.cols != 0 || dg.data != dp0 || dg.rows != nk) return 10;
            surf(d0, cuda::GpuMat(), kg);                       // detect only: the same keypoint buffer again
            if (kg.data != kp0 || kg.cols != nk) return 10;
        }
        // superres adapters (the in-tree caller of the flow classes): planar flow == split of the class's own result
        if (argc > 3) {
            Ptr<superres::DualTVL1OpticalFlow> sr = superres::createOptFlow_DualTVL1_CUDA();
            if (sr->getIterations() != 300 || sr->getScalesNumber() != 5) return 8;
            sr->setIterations(10);
            sr->setEpsilon(0.0);
            cuda::GpuMat u, v;
            sr->calc(d0, d1, u, &v);
            if (u.type() != CV_32FC1 || v.type() != CV_32FC1 || u.size() != d0.size()) return 8;
            std::vector<float> hu((size_t)h * w), hv((size_t)h * w);
            u.download(hu.data(), (size_t)w * 4);
            v.download(hv.data(), (size_t)w * 4);
            FILE *o2 = fopen(argv[3], "wb");
            fwrite(hu.data(), 4, hu.size(), o2);
            fwrite(hv.data(), 4, hv.size(), o2);
            Ptr<superres::FarnebackOpticalFlow> sf = superres::createOptFlow_Farneback_CUDA();
            if (sf->getWindowSize() != 13) return 8;
            sf->setLevelsNumber(3);
            cuda::GpuMat merged;
            sf->calc(d0, d1, merged);
            if (merged.type() != CV_32FC2) return 8;
            merged.download(hf.data(), (size_t)w * 8);
            fwrite(hf.data(), 4, hf.size(), o2);
            fclose(o2);
            sr->collectGarbage();
        }
        // descriptor matching + disparity post-filter (the steps after the hot path's SURF and StereoBM)
        if (nk > 0) {
            cuda::GpuMat ddesc(nk, 64, CV_32FC1);
            ddesc.upload(desc.data(), 64 * sizeof(float));
            Ptr<cuda::DescriptorMatcher> bf = cuda::DescriptorMatcher::createBFMatcher(NORM_L2);
            std::vector<DMatch> mm;
            bf->match(ddesc, ddesc, mm);
            if ((int)mm.size() != nk) return 9;
            for (int k = 0; k < nk; ++k) if (mm[k].trainIdx != k || mm[k].distance != 0.f) return 9;
            std::vector<std::vector<DMatch> > knn;
            bf->knnMatch(ddesc, ddesc, knn, 2);
            if ((int)knn.size() != nk || (nk > 1 && knn[0].size() != 2)) return 9;
            // any k, NORM_L1, radius and the train collection (add() twice, like the reference's *_Collection tests)
            Ptr<cuda::DescriptorMatcher> b1 = cuda::DescriptorMatcher::createBFMatcher(NORM_L1);
            b1->knnMatch(ddesc, ddesc, knn, 3);
            if ((int)knn.size() != nk || (int)knn[0].size() != std::min(3, nk) || knn[0][0].trainIdx != 0 || knn[0][0].distance != 0.f) return 10;
            for (size_t j = 1; j < knn[0].size(); ++j) if (knn[0][j].distance < knn[0][j - 1].distance) return 10;
            const int h0 = nk / 2;
            if (h0 > 0) {
                bf->add(std::vector<cuda::GpuMat>(1, ddesc(Rect(0, 0, 64, h0))));
                bf->add(std::vector<cuda::GpuMat>(1, ddesc(Rect(0, h0, 64, nk - h0))));
                if (bf->empty() || bf->getTrainDescriptors().size() != 2) return 10;
                bf->match(ddesc, mm);
                if ((int)mm.size() != nk) return 10;
                for (int k = 0; k < nk; ++k)
                    if (mm[k].distance != 0.f || mm[k].imgIdx != (k < h0 ? 0 : 1) || mm[k].trainIdx != (k < h0 ? k : k - h0)) return 10;
                bf->knnMatch(ddesc, knn, 3);
                if ((int)knn.size() != nk || knn[nk - 1][0].imgIdx != 1 || knn[nk - 1][0].trainIdx != nk - 1 - h0) return 10;
                std::vector<std::vector<DMatch> > rad;
                bf->radiusMatch(ddesc, rad, 1e-6f);            // only the descriptor itself (and exact duplicates) is that close
                if ((int)rad.size() != nk || rad[0].empty() || rad[0][0].distance != 0.f) return 10;
                bf->clear();
                if (!bf->empty()) return 10;
            }
            std::vector<std::vector<DMatch> > rad;
            bf->radiusMatch(ddesc, ddesc, rad, 10.f);          // unit-norm descriptors: everything is within 10
            if ((int)rad.size() != nk || (int)rad[0].size() != nk || rad[0][0].trainIdx != 0) return 10;
            for (size_t j = 1; j < rad[0].size(); ++j) if (rad[0][j].distance < rad[0][j - 1].distance) return 10;
        }
        Ptr<cuda::DensePyrLKOpticalFlow> lk = cuda::DensePyrLKOpticalFlow::create();
        if (lk->getWinSize().width != 13 || lk->getMaxLevel() != 3 || lk->getNumIters() != 30 ||
            lk->getDefaultName() != "DenseOpticalFlow.DensePyrLKOpticalFlow") return 9;
        cuda::GpuMat lkflow;
        lk->calc(d0, d1, lkflow, stream);
        stream.waitForCompletion();
        if (lkflow.type() != CV_32FC2 || lkflow.size() != d0.size()) return 9;
        Ptr<cuda::StereoSGM> sgm = cuda::createStereoSGM(0, 64);
        if (sgm->getP1() != 10 || sgm->getP2() != 120 || sgm->getMode() != cuda::StereoSGM::MODE_HH4 || sgm->getBlockSize() != -1) return 9;
        cuda::GpuMat sdisp;
        sgm->compute(d0, d1, sdisp);
        if (sdisp.depth() != 3 /* CV_16S */ || sdisp.size() != d0.size()) return 9;
        Ptr<cuda::DisparityBilateralFilter> dbf = cuda::createDisparityBilateralFilter(32, 3, 1);
        if (dbf->getRadius() != 3 || dbf->getSigmaRange() != 10.0) return 9;
        cuda::GpuMat refined;
        dbf->apply(disp, d0, refined);
        if (refined.type() != CV_8UC1 || refined.size() != disp.size()) return 9;
        // miflow extensions (everything the reference classes do not have lives in cv::cuda::miflow): the batched-frames mode
        // returns, for every pair, the bytes of the single call; the numerics switches are accepted
        {
            std::vector<cuda::GpuMat> A(3, d0), B(3, d1), F, FB, D;
            cuda::miflow::calcBatch(tvl1, A, B, F, stream);
            cuda::miflow::calcBatch(fb, A, B, FB, stream);
            cuda::miflow::computeBatch(bm, A, B, D, stream);
            stream.waitForCompletion();
            if (F.size() != 3 || FB.size() != 3 || D.size() != 3) return 11;
            std::vector<float> g0((size_t)h * w * 2), g1((size_t)h * w * 2);
            flow.download(g0.data(), (size_t)w * 8);
            F[2].download(g1.data(), (size_t)w * 8);
            if (g0 != g1) return 11;
            fbflow.download(g0.data(), (size_t)w * 8);
            FB[1].download(g1.data(), (size_t)w * 8);
            if (g0 != g1) return 11;
            std::vector<unsigned char> e0((size_t)h * w), e1((size_t)h * w);
            disp.download(e0.data(), (size_t)w);
            D[0].download(e1.data(), (size_t)w);
            if (e0 != e1) return 11;
            Ptr<cuda::OpticalFlowDual_TVL1> t2 = cuda::OpticalFlowDual_TVL1::create(0.25, 0.15, 0.3, 3, 2, 0.0, 4);
            cuda::miflow::setSemantics(t2, MI_SEM_CUDA_COMPAT);
            cuda::miflow::setExactMath(t2, true);
            cuda::GpuMat f2;
            t2->calc(d0, d1, f2);
            if (f2.type() != CV_32FC2) return 11;
            // class defaults (300 iterations, epsilon 0.01: the device-decided stop) with and without the stop-slack extension
            Ptr<cuda::OpticalFlowDual_TVL1> t3 = cuda::OpticalFlowDual_TVL1::create();
            cuda::GpuMat f3, f4;
            t3->calc(d0, d1, f3);
            cuda::miflow::setStopSlack(t3, 1);
            t3->calc(d0, d1, f4);
            bool rejected = false;
            try { cuda::miflow::setStopSlack(t3, 99); } catch (const cv::Exception &) { rejected = true; }
            if (!rejected || f3.size() != f4.size()) return 12;
        }
        {   // Setters of the stereo classes validate at compute() / apply(), like the reference's (stereobm.cpp:143-146): an invalid value
            // as the FIRST call after construction is stored, the compute throws, and every other parameter still holds the
            // constructor's value (the rollback copy used to be taken before the constructor body had filled them).
            Ptr<cuda::StereoBM> b2 = cuda::createStereoBM(32, 9);
            b2->setNumDisparities(7);
            if (b2->getNumDisparities() != 7 || b2->getBlockSize() != 9 || b2->getPreFilterCap() != 31 || b2->getTextureThreshold() != 3) return 13;
            bool rej = false;
            cuda::GpuMat dd;
            try { b2->compute(d0, d1, dd); } catch (const cv::Exception &) { rej = true; }
            if (!rej) return 13;
            b2->setNumDisparities(32);
            b2->compute(d0, d1, dd);
            std::vector<unsigned char> e0((size_t)h * w), e1((size_t)h * w);
            disp.download(e0.data(), (size_t)w);
            dd.download(e1.data(), (size_t)w);
            if (e0 != e1) return 13;
            Ptr<cuda::StereoSGM> s2 = cuda::createStereoSGM(0, 128, 10, 120, 5, cuda::StereoSGM::MODE_HH4);
            s2->setNumDisparities(100);
            if (s2->getNumDisparities() != 100 || s2->getP1() != 10 || s2->getP2() != 120 || s2->getUniquenessRatio() != 5 ||
                s2->getMode() != cuda::StereoSGM::MODE_HH4) return 14;
            Ptr<cuda::DisparityBilateralFilter> f2 = cuda::createDisparityBilateralFilter(64, 3, 1);
            f2->setRadius(-1);
            if (f2->getNumDisparities() != 64 || f2->getRadius() != -1 || f2->getNumIters() != 1) return 15;
            rej = false;
            try { cuda::GpuMat o2; f2->apply(disp, d0, o2, stream); } catch (const cv::Exception &) { rej = true; }
            if (!rej) return 15;
        }
        // error mapping: CV_Assert-style failures surface as cv::Exception
        bool threw = false;
        try { cuda::GpuMat bad(h, w, CV_32FC1); bm->compute(bad, bad, disp); } catch (const cv::Exception &) { threw = true; }
        return threw ? 0 : 6;
    } catch (const cv::Exception &e) {
        fprintf(stderr, "cv::Exception %d: %s\n", e.code, e.what());
        return 3;
    }
}
