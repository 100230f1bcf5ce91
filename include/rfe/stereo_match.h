// rfe/stereo_match.h -- drop-in body for Frame::ComputeStereoMatches (reference src/Frame.cc:1159-1446,
// called from the stereo Frame constructor at src/Frame.cc:171) on top of rfe_stereo_match.
// Works on any Frame-like type with the reference's member names: N, mvKeys, mvKeysRight, mDescriptors,
// mDescriptorsRight, imgLeft, imgRight, mb, mbf, mvuRight, mvDepth.  nLevels must be 1 (as for the extractor).
// `ctx` can be the session of either extractor: mpSPextractorLeft->featureExtractor->ExtractorSession.
#pragma once
#include <vector>
#include "../rover_fe.h"
#include "cv_compat.h"

namespace ORB_SLAM3 {

template <class FrameT>
int ComputeStereoMatches_rfe(rfe_ctx* ctx, FrameT& F) {
    const int N = (int)F.mvKeys.size(), Nr = (int)F.mvKeysRight.size();
    F.mvuRight = std::vector<float>(N, -1.0f);
    F.mvDepth = std::vector<float>(N, -1.0f);
    if (N == 0) return 0;
    std::vector<float> kl((size_t)N * 2), kr((size_t)(Nr > 0 ? Nr : 1) * 2), dl((size_t)N * 256), dr((size_t)(Nr > 0 ? Nr : 1) * 256);
    for (int i = 0; i < N; ++i) {
        kl[2 * i] = F.mvKeys[i].pt.x; kl[2 * i + 1] = F.mvKeys[i].pt.y;
        const float* s = F.mDescriptors.template ptr<float>(i);
        std::copy(s, s + 256, dl.begin() + (size_t)i * 256);
    }
    for (int i = 0; i < Nr; ++i) {
        kr[2 * i] = F.mvKeysRight[i].pt.x; kr[2 * i + 1] = F.mvKeysRight[i].pt.y;
        const float* s = F.mDescriptorsRight.template ptr<float>(i);
        std::copy(s, s + 256, dr.begin() + (size_t)i * 256);
    }
    return rfe_stereo_match(ctx, F.imgLeft.template ptr<unsigned char>(0), F.imgRight.template ptr<unsigned char>(0), F.imgLeft.rows,
                            F.imgLeft.cols, (int)F.imgLeft.step, kl.data(), N, kr.data(), Nr, dl.data(), dr.data(), F.mb, F.mbf,
                            F.mvuRight.data(), F.mvDepth.data());
}

}  // namespace ORB_SLAM3
