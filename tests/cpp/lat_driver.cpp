// lat_driver.cpp -- the reference's OWN call pattern on the clock: batch 1, host pointers in and out, through the drop-in C++
// classes, exactly as a Rover-SLAM Tracking thread would drive them (bench.py's `latency.dropin`).
//   c2  BASELINE configs[1]: one 640x480 frame through SPextractor::operator()            (src/Frame.cc:544-559 -> SPextractor.cc:516-617)
//   c3  BASELINE configs[2]: a 640x480 pair: two extractions + SPmatcher::MatchingPoints_onnx(Frame&, Frame&)
//                                                                                          (src/Matchers/SPmatcher.cc:457-542)
//   c5  BASELINE configs[4]: a 752x480 stereo frame: left and right extraction on TWO THREADS with two extractors
//       (src/Frame.cc:142-147), ComputeStereoMatches (:1159-1446) and the temporal match against the previous left view
//       (SearchBySP, src/Matchers/SPmatcher.cc:1050-1054)
// Every call copies its image / keypoints / descriptors from pageable host memory and its results back: this is what the reference's
// interface forces, and what the device-resident entry points (bench.py `latency.resident`) avoid.
// usage: lat_driver <pair.u8: 2 x 480x640> <stereo.u8: T x 2 x 480x752> T steps warmup <out.bin>
// weights via $RFE_SP_WEIGHTS / $RFE_LG_WEIGHTS; prints ONE JSON line; out.bin holds the last results for bench.py's oracle check.
#include <sched.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "Extractors/SPextractor.h"
#include "Matchers/SPmatcher.h"
#include "rfe/stereo_match.h"

struct MockFrame {                      // the members SPmatcher::MatchingPoints_onnx(Frame&, Frame&) and ComputeStereoMatches read
    std::vector<cv::KeyPoint> mvKeys, mvKeysRight;
    cv::Mat mDescriptors, mDescriptorsRight, imgLeft, imgRight;
    float mb = 0.11f, mbf = 0.11f * 435.0f;
    std::vector<float> mvuRight, mvDepth;
};

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// per-call samples -> {"mean": .., "p50": .., "p95": .., "p99": .., "min": .., "max": .., "first": ..} (ms): a Tracking thread sees single calls, not a mean
struct Samples {
    std::vector<double> v;
    double t = 0;
    void start() { t = now_ms(); }
    void stop() { v.push_back(now_ms() - t); }
    std::string json() const {
        std::vector<double> s(v);
        std::sort(s.begin(), s.end());
        auto q = [&](double p) { return s.empty() ? 0.0 : s[std::min(s.size() - 1, (size_t)(p * (s.size() - 1) + 0.5))]; };
        double sum = 0;
        for (double x : s) sum += x;
        char b[256];
        snprintf(b, sizeof(b), "{\"mean\": %.4f, \"p50\": %.4f, \"p95\": %.4f, \"p99\": %.4f, \"min\": %.4f, \"max\": %.4f, \"first\": %.4f, \"calls\": %d}",
                 s.empty() ? 0.0 : sum / s.size(), q(0.5), q(0.95), q(0.99), s.empty() ? 0.0 : s.front(), s.empty() ? 0.0 : s.back(), v.empty() ? 0.0 : v.front(), (int)s.size());
        return b;
    }
    double mean() const { double sum = 0; for (double x : v) sum += x; return v.empty() ? 0.0 : sum / v.size(); }
};
static void put(FILE* f, const void* p, size_t n) { fwrite(p, 1, n, f); }
static void put_frame(FILE* f, const std::vector<cv::KeyPoint>& k, const cv::Mat& d) {
    const int32_t n = (int32_t)k.size();
    put(f, &n, 4);
    for (const auto& kp : k) { float v[3] = {kp.pt.x, kp.pt.y, kp.response}; put(f, v, sizeof(v)); }
    for (int r = 0; r < n; ++r) put(f, d.ptr<float>(r), 256 * 4);
}

// RFE_LAT_CPULIST ("0-15,64-79": the CPUs local to the GPU, /sys/class/drm/cardN/device/local_cpulist): run on those.  A Tracking thread on
// the far NUMA node pays for every doorbell, completion signal and staging copy across the socket link; bench.py passes the list so that
// the drop-in figures do not depend on where the scheduler happened to start this process (do the same in deployment).
static std::string pin_to_cpulist(const char* list) {
    if (!list || !*list) return "";
    cpu_set_t set; CPU_ZERO(&set);
    int count = 0;
    for (const char* p = list; *p;) {
        char* e; long a = strtol(p, &e, 10), b = a;
        if (e == p) break;
        if (*e == '-') { p = e + 1; b = strtol(p, &e, 10); }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) { CPU_SET((int)c, &set); ++count; }
        p = (*e == ',') ? e + 1 : e;
        if (*e && *e != ',') break;
    }
    if (count == 0 || sched_setaffinity(0, sizeof(set), &set) != 0) return "";
    return list;
}

int main(int argc, char** argv) {
    if (argc < 7) { fprintf(stderr, "usage: lat_driver pair.u8 stereo.u8 T steps warmup out.bin\n"); return 2; }
    const std::string pinned_to = pin_to_cpulist(getenv("RFE_LAT_CPULIST"));
    const int H = 480, W = 640, Hs = 480, Ws = 752;
    const int T = atoi(argv[3]), steps = atoi(argv[4]), warm = atoi(argv[5]);
    std::vector<unsigned char> pair((size_t)2 * H * W), stereo((size_t)T * 2 * Hs * Ws);
    FILE* fi = fopen(argv[1], "rb");
    if (!fi || fread(pair.data(), 1, pair.size(), fi) != pair.size()) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    fclose(fi);
    fi = fopen(argv[2], "rb");
    if (!fi || fread(stereo.data(), 1, stereo.size(), fi) != stereo.size()) { fprintf(stderr, "cannot read %s\n", argv[2]); return 2; }
    fclose(fi);

    // as Tracking builds them: two extractors (left / right, src/Tracking.cc:645-651) and one matcher (:70)
    ORB_SLAM3::SPextractor extL(1000, 1.2f, 1, 20, 7), extR(1000, 1.2f, 1, 20, 7);
    ORB_SLAM3::SPmatcher matcher(0.0f);
    if (!extL.featureExtractor->ExtractorSession || !extR.featureExtractor->ExtractorSession || !matcher.featureMatcher->MatcherSession) return 3;

    // ---- c2: one frame
    MockFrame f0, f1;
    f0.imgLeft = cv::Mat(H, W, CV_8UC1, pair.data());
    f1.imgLeft = cv::Mat(H, W, CV_8UC1, pair.data() + (size_t)H * W);
    Samples q2, q2abi, q2pin, q3, q3x, q3m, q5;
    for (int i = -warm; i < steps; ++i) {
        if (i >= 0) q2.start();
        f0.mvKeys.clear();
        extL(f0.imgLeft, f0.mvKeys, f0.mDescriptors);
        if (i >= 0) q2.stop();
    }
    const double c2 = q2.mean();
    // the C-ABI host entry alone on the same frame, caller-owned (reused) host buffers: c2 minus this = what the class shim adds per call
    // (tensor set-up, int64 keypoints, std::vector<cv::KeyPoint>, the K x 256 cv::Mat); this minus latency.resident c2 = staging + PCIe + the final sync
    {
        rfe_ctx* cx = extL.featureExtractor->ExtractorSession;
        const int K = extL.featureExtractor->max_keypoints;
        std::vector<int32_t> kxy((size_t)K * 2);
        std::vector<float> sc((size_t)K), de((size_t)K * 256);
        int32_t nn = 0;
        for (int i = -warm; i < steps; ++i) {
            if (i >= 0) q2abi.start();
            if (rfe_extract_u8(cx, pair.data(), H, W, W, 1, K, extL.featureExtractor->detection_threshold, &nn, kxy.data(), sc.data(), de.data()) != RFE_OK) return 6;
            if (i >= 0) q2abi.stop();
        }
        void* pd = nullptr;                                   // ... and with the descriptor output in rfe_host_malloc'ed memory (direct DMA, what the shim's tensors use)
        if (rfe_host_malloc((size_t)K * 1024, &pd) == RFE_OK) {
            for (int i = -warm; i < steps; ++i) {
                if (i >= 0) q2pin.start();
                if (rfe_extract_u8(cx, pair.data(), H, W, W, 1, K, extL.featureExtractor->detection_threshold, &nn, kxy.data(), sc.data(), (float*)pd) != RFE_OK) return 6;
                if (i >= 0) q2pin.stop();
            }
            if (memcmp(pd, de.data(), (size_t)K * 1024) != 0) { fprintf(stderr, "pinned-output descriptors differ from the staged ones\n"); return 7; }
            rfe_host_free(pd);
        }
    }

    // ---- c3: a pair = two extractions + one LightGlue match (Frame overload: true image size)
    std::vector<int> vn;
    int s3 = 0;
    for (int i = -warm; i < steps; ++i) {
        if (i >= 0) { q3.start(); q3x.start(); }
        f0.mvKeys.clear(); f1.mvKeys.clear();
        extL(f0.imgLeft, f0.mvKeys, f0.mDescriptors);
        extL(f1.imgLeft, f1.mvKeys, f1.mDescriptors);
        if (i >= 0) { q3x.stop(); q3m.start(); }
        vn.clear();                                           // callers hand over a fresh vector (resize(M, -1) keeps old entries, SPmatcher.cc:460)
        s3 = matcher.MatchingPoints_onnx(f0, f1, vn);
        if (i >= 0) { q3m.stop(); q3.stop(); }
    }
    const double c3 = q3.mean();

    // ---- c5: stereo stream
    MockFrame cur, prev;
    std::vector<int> vt;
    int s5 = 0, last_t = 0;
    bool have_prev = false;
    for (int i = -warm; i < steps; ++i) {
        if (i >= 0) q5.start();
        const int t = ((i + warm) % T);
        cur.imgLeft = cv::Mat(Hs, Ws, CV_8UC1, stereo.data() + (size_t)(2 * t) * Hs * Ws);
        cur.imgRight = cv::Mat(Hs, Ws, CV_8UC1, stereo.data() + (size_t)(2 * t + 1) * Hs * Ws);
        cur.mvKeys.clear(); cur.mvKeysRight.clear();
        std::thread thr([&] { extR(cur.imgRight, cur.mvKeysRight, cur.mDescriptorsRight); });      // src/Frame.cc:142-147
        extL(cur.imgLeft, cur.mvKeys, cur.mDescriptors);
        thr.join();
        if (ORB_SLAM3::ComputeStereoMatches_rfe(extL.featureExtractor->ExtractorSession, cur) != 0) return 5;
        vt.clear();
        s5 = have_prev ? matcher.MatchingPoints_onnx(cur, prev, vt) : 0;                             // SearchBySP(current, last)
        if (i + 1 < steps) {
            prev.mvKeys = cur.mvKeys; prev.mDescriptors = cur.mDescriptors.clone(); prev.imgLeft = cur.imgLeft;
            have_prev = true;
        }
        last_t = t;
        if (i >= 0) q5.stop();
    }
    const double c5 = q5.mean();

    FILE* fo = fopen(argv[6], "wb");
    if (!fo) return 4;
    put_frame(fo, f0.mvKeys, f0.mDescriptors); put_frame(fo, f1.mvKeys, f1.mDescriptors);
    int32_t m = (int32_t)vn.size();
    put(fo, &s3, 4); put(fo, &m, 4); put(fo, vn.data(), (size_t)m * 4);
    const int32_t tt[2] = {last_t, (last_t + T - 1) % T};
    put(fo, tt, 8);
    put_frame(fo, cur.mvKeys, cur.mDescriptors); put_frame(fo, cur.mvKeysRight, cur.mDescriptorsRight); put_frame(fo, prev.mvKeys, prev.mDescriptors);
    m = (int32_t)cur.mvuRight.size();
    put(fo, &m, 4); put(fo, cur.mvuRight.data(), (size_t)m * 4); put(fo, cur.mvDepth.data(), (size_t)m * 4);
    m = (int32_t)vt.size();
    put(fo, &s5, 4); put(fo, &m, 4); put(fo, vt.data(), (size_t)m * 4);
    fclose(fo);
    printf("{\"c2_ms\": %.4f, \"c3_ms\": %.4f, \"c5_ms\": %.4f, \"steps\": %d, \"warmup\": %d, \"c2_keypoints\": %d, \"c3_matches\": %d, "
           "\"c5_left_keypoints\": %d, \"c5_temporal_matches\": %d, \"cpu_affinity\": \"%s\", \"per_call_ms\": {\"c2\": %s, \"c2_c_abi_only\": %s, \"c2_c_abi_pinned_desc\": %s, \"c3\": %s, "
           "\"c3_two_extractions\": %s, \"c3_match\": %s, \"c5\": %s}}\n",
           c2, c3, c5, steps, warm, (int)f0.mvKeys.size(), s3, (int)cur.mvKeys.size(), s5,
           pinned_to.c_str(), q2.json().c_str(), q2abi.json().c_str(), q2pin.json().c_str(), q3.json().c_str(), q3x.json().c_str(), q3m.json().c_str(), q5.json().c_str());
    return 0;
}
