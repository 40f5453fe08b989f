// OpenCV-free replay of the reference driver's frame loop (main.cu:192-272) through the drop-in gpu:: surface.
//
// main.cu captures webcam frames with OpenCV; everything between capture and display is: grayscale -> bilateral
// pre-filter -> pyramid -> calc_opt_flow per level, coarse to fine -> the dense field its arrows sample (main.cu:138-147)
// -> swap pyramids.  This program runs exactly those calls, in that order, on a stream of raw frames:
//
//   replay_main W H N [frames.raw [field.raw]]
//
// frames.raw holds N+1 frames of W x H x 3 bytes (the first one primes the previous pyramid, main.cu:198-209); without it
// a synthetic stream is generated.  field.raw receives, per processed frame, the composed level-0 field (2 * W * H floats).
// A line per frame is printed with the number of finite vectors, the median and a checksum of the field's bits.
// tests/test_gpu_surface.py runs it and checks the field against the CPU oracle.  Build:
//   hipcc -std=c++17 -Iinclude examples/replay_main.cpp -Lcuda_optical_flow_2_amd -lofx_hip -o replay_main
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "OptFlowCpu.hpp"
#include "OptFlowGpu.cuh"
#include "OptFlowUtils.hpp"
#include "kernels.hpp"
#include "main.h"
#include "ofx.h"

// pyramid allocation as main.cu:95-104 does it: level k is (w>>k) x (h>>k) x channels
template <typename T, int CH>
static T **alloc_pyramid(int w, int h, int levels)
{
    T **p = (T **)malloc(levels * sizeof(T *));
    for (int k = 0; k < levels; ++k) p[k] = (T *)calloc((size_t)(w >> k) * (h >> k) * CH, sizeof(T));
    return p;
}

static void synth_frame(unsigned char *bgr, int w, int h, float dx, float dy)
{
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const float fx = x - dx, fy = y - dy;
            const float v = 127.5f + 60.0f * std::sin(fx * 0.11f) * std::cos(fy * 0.07f) + 50.0f * std::sin((fx + fy) * 0.05f);
            unsigned char *p = bgr + 3 * ((size_t)y * w + x);
            p[0] = (unsigned char)std::min(255.0f, std::max(0.0f, v));
            p[1] = (unsigned char)std::min(255.0f, std::max(0.0f, v * 0.9f + 10.0f));
            p[2] = (unsigned char)std::min(255.0f, std::max(0.0f, v * 1.05f));
        }
}

static bool failed(const char *what, int f)
{
    if (gpu_compat_last_status() == 0) return false;
    fprintf(stderr, "frame %d: %s failed: %s\n", f, what, ofx_last_error());
    return true;
}

int main(int argc, char **argv)
{
    const int w = argc > 1 ? atoi(argv[1]) : 640, h = argc > 2 ? atoi(argv[2]) : 480; // main.cu:183-184
    const int frames = argc > 3 ? atoi(argv[3]) : 3;
    FILE *in = argc > 4 ? fopen(argv[4], "rb") : nullptr, *out = argc > 5 ? fopen(argv[5], "wb") : nullptr;
    if ((argc > 4 && !in) || (argc > 5 && !out)) {
        fprintf(stderr, "cannot open %s\n", argc > 5 && !out ? argv[5] : argv[4]);
        return 2;
    }
    const int levels = 4;                                                               // main.cu:192
    const size_t n3 = (size_t)w * h * 3;
    std::vector<unsigned char> frame(n3), gray(n3), filtered(n3);
    std::vector<float> field((size_t)w * h * 2);
    auto next_frame = [&](int f) {
        if (in) return fread(frame.data(), 1, n3, in) == n3;
        synth_frame(frame.data(), w, h, 2.0f * f, 1.0f * f);
        return true;
    };

    if (!next_frame(0)) return 2;
    gpu::grayscale_avg(frame.data(), gray.data(), h, w);                                // main.cu:198 (rows, cols)
    unsigned char **prev_pyramid = alloc_pyramid<unsigned char, 3>(w, h, levels);       // main.cu:203-205
    unsigned char **pyramid = alloc_pyramid<unsigned char, 3>(w, h, levels);
    float **flow_pyramid = alloc_pyramid<float, 2>(w, h, levels);                       // main.cu:220
    memcpy(prev_pyramid[0], gray.data(), gray.size());                                  // main.cu:208
    gpu::gauss_pyramid(prev_pyramid, w, h, levels, GAUS_KERNEL_3x3, 3, 3);              // main.cu:209
    if (failed("priming", 0)) return 1;

    for (int f = 1; f <= frames; ++f) {
        if (!next_frame(f)) return 2;
        gpu::grayscale_avg(frame.data(), gray.data(), h, w);                            // main.cu:232
        gpu::bilinear_filter(gray.data(), gray.data(), filtered.data(), w, h, 9, 9, 2, 10); // main.cu:240
        memcpy(pyramid[0], filtered.data(), filtered.size());                           // main.cu:246
        gpu::gauss_pyramid(pyramid, w, h, levels, GAUS_KERNEL_3x3, 3, 3);               // main.cu:250
        for (int k = levels - 1; k >= 0; --k)                                           // main.cu:256-262
            gpu::calc_opt_flow(prev_pyramid[k], pyramid[k], w >> k, h >> k, flow_pyramid, k, levels);
        if (failed("flow", f)) return 1;
        // the dense field visualizeFlowField composes for its arrows (main.cu:138-147), here for every pixel
        if (ofx_compose_flow_host(flow_pyramid, w, h, levels, 0, field.data()) != 0) {
            fprintf(stderr, "frame %d: compose failed: %s\n", f, ofx_last_error());
            return 1;
        }
        if (out && fwrite(field.data(), sizeof(float), field.size(), out) != field.size()) return 2;
        std::vector<float> us, vs;
        uint64_t sum = 1469598103934665603ull; // FNV-1a over the field's bits
        for (size_t p = 0; p < (size_t)w * h; ++p) {
            const float u = field[2 * p], v = field[2 * p + 1];
            uint32_t b[2];
            memcpy(b, &field[2 * p], 8);
            for (int i = 0; i < 2; ++i) sum = (sum ^ b[i]) * 1099511628211ull;
            if (std::isfinite(u) && std::isfinite(v)) {
                us.push_back(u);
                vs.push_back(v);
            }
        }
        float mu = 0, mv = 0;
        if (!us.empty()) {
            std::nth_element(us.begin(), us.begin() + us.size() / 2, us.end());
            std::nth_element(vs.begin(), vs.begin() + vs.size() / 2, vs.end());
            mu = us[us.size() / 2];
            mv = vs[vs.size() / 2];
        }
        printf("frame %d: %zu finite vectors, median (%.3f, %.3f), fnv %016llx\n", f, us.size(), mu, mv, (unsigned long long)sum);
        std::swap(prev_pyramid, pyramid);                                               // main.cu:270-272
    }
    if (in) fclose(in);
    if (out) fclose(out);
    return 0;
}
