// Host-side helpers of the reference's utils namespace (its OptFlowUtils.hpp:5-52), exported by libofx_hip.so.
#pragma once

#define _USE_MATH_DEFINES
#include <math.h>

namespace utils {

// Binarise a 1-channel image in place: values in [20,240) become 255, everything else 0.
void cleanup_outliers(unsigned char *img1, int w, int h);

// dest = a - b element-wise (the GPU flow path's It = It2 - It1).
inline void arr_sub_float(float *a, float *b, int n, float *dest)
{
    for (int k = 0; k < n; ++k) dest[k] = a[k] - b[k];
}

// Nearest-neighbour upscale by 2^n (debug visualisation helpers).
void upscale_3ch(unsigned char *img3, int w, int h, int n, unsigned char *out3);
void upscale_1ch(unsigned char *img1, int w, int h, int n, unsigned char *out1);

// Normalised 2-D Gaussian; kernel_size == -1 picks 2*pi*sigma, even sizes are bumped to the next odd size.
void generate_gaussian_kernel(double sigmaS, int kernel_size, double *dest);

} // namespace utils
