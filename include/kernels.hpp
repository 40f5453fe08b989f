// Stencil tables of the reference (its kernels.hpp:3-13), exported by libofx_hip.so with the same names.
#pragma once

extern const float Dx_3x3[];          // Sobel x
extern const float Dx_3x3_t[];        // Sobel x / 3, sign flipped
extern const float Dy_3x3[];          // Sobel y
extern const float Dt_3x3[];          // temporal smoothing mask, gain 15
extern const float Dt_3x3_n[];        // the same, normalised
extern const float Dy_DIAGONAL_2x2[]; // Roberts-style pairs stored in 3x3 tables
extern const float Dy_2x2[];
extern const float Dz_2x2[];
extern const float Dx_5x5[];
extern const float GAUS_KERNEL_5x5[];
extern const float GAUS_KERNEL_3x3[]; // [1 2 1]^T [1 2 1] / 16
