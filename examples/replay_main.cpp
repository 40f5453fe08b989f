// OpenCV-free replay of the reference driver's call sequence (main.cu:192-272) through the drop-in gpu:: surface.
//
// main.cu captures webcam frames with OpenCV; everything between capture and display is: grayscale -> bilateral
// pre-filter -> pyramid -> calc_opt_flow per level, coarse to fine -> swap pyramids.  This program runs exactly those
// calls on a synthetic frame stream (a smooth texture translating by (2,1) px per frame) and prints the median flow,
// so the boundary can be exercised where OpenCV is absent.  Build:
//   hipcc -std=c++17 -Iinclude examples/replay_main.cpp -Lcuda_optical_flow_2_amd -lofx_hip -o replay_main
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

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

int main(int argc, char **argv)
{
    const int w = argc > 1 ? atoi(argv[1]) : 640, h = argc > 2 ? atoi(argv[2]) : 480; // main.cu:183-184
    const int frames = argc > 3 ? atoi(argv[3]) : 3;
    const int levels = 4;                                                               // main.cu:192
    std::vector<unsigned char> frame((size_t)w * h * 3), gray((size_t)w * h * 3), filtered((size_t)w * h * 3);

    synth_frame(frame.data(), w, h, 0, 0);
    gpu::grayscale_avg(frame.data(), gray.data(), h, w);                                // main.cu:198 (rows, cols)
    unsigned char **prev_pyramid = alloc_pyramid<unsigned char, 3>(w, h, levels);       // main.cu:203-205
    unsigned char **pyramid = alloc_pyramid<unsigned char, 3>(w, h, levels);
    float **flow_pyramid = alloc_pyramid<float, 2>(w, h, levels);                       // main.cu:220
    memcpy(prev_pyramid[0], gray.data(), gray.size());                                  // main.cu:208
    gpu::gauss_pyramid(prev_pyramid, w, h, levels, GAUS_KERNEL_3x3, 3, 3);              // main.cu:209

    for (int f = 1; f <= frames; ++f) {
        synth_frame(frame.data(), w, h, 2.0f * f, 1.0f * f);
        gpu::grayscale_avg(frame.data(), gray.data(), h, w);                            // main.cu:232
        gpu::bilinear_filter(gray.data(), gray.data(), filtered.data(), w, h, 9, 9, 2, 10); // main.cu:240
        memcpy(pyramid[0], filtered.data(), filtered.size());                           // main.cu:246
        gpu::gauss_pyramid(pyramid, w, h, levels, GAUS_KERNEL_3x3, 3, 3);               // main.cu:250
        for (int k = levels - 1; k >= 0; --k)                                           // main.cu:256-262
            gpu::calc_opt_flow(prev_pyramid[k], pyramid[k], w >> k, h >> k, flow_pyramid, k, levels);
        if (gpu_compat_last_status() != 0) {
            fprintf(stderr, "frame %d failed: %s\n", f, ofx_last_error());
            return 1;
        }
        // the dense field main.cu:138-147 composes for its arrows; report the median of the finite level-0 residuals
        std::vector<float> us, vs;
        for (size_t p = 0; p < (size_t)w * h; ++p)
            if (std::isfinite(flow_pyramid[0][2 * p]) && std::isfinite(flow_pyramid[0][2 * p + 1])) {
                us.push_back(flow_pyramid[0][2 * p]);
                vs.push_back(flow_pyramid[0][2 * p + 1]);
            }
        if (!us.empty()) {
            std::nth_element(us.begin(), us.begin() + us.size() / 2, us.end());
            std::nth_element(vs.begin(), vs.begin() + vs.size() / 2, vs.end());
            printf("frame %d: %zu finite level-0 vectors, median residual (%.3f, %.3f)\n", f, us.size(), us[us.size() / 2],
                   vs[vs.size() / 2]);
        }
        std::swap(prev_pyramid, pyramid);                                               // main.cu:270-272
    }
    return 0;
}
