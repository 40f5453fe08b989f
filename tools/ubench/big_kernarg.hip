// Does a kernel launch with a 12/16/24 KB by-value argument work on this stack, and what does it cost per launch?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
template <int N>
struct Big {
    int v[N];
};
template <int N>
__global__ void k(Big<N> b, int *out)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = b.v[0] + b.v[N / 2] + b.v[N - 1];
}
template <int N>
void run(int *d)
{
    Big<N> b;
    for (int i = 0; i < N; ++i) b.v[i] = i;
    hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, b, d);
    hipError_t e = hipDeviceSynchronize();
    int h = -1;
    hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    auto t0 = std::chrono::steady_clock::now();
    const int reps = 2000;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, b, d);
    auto t1 = std::chrono::steady_clock::now();
    hipDeviceSynchronize();
    auto t2 = std::chrono::steady_clock::now();
    printf("kernarg %6zu B: %s, result %d (expect %d), host %.2f us/launch, total %.2f us/launch\n", sizeof(b), hipGetErrorString(e), h,
           0 + N / 2 + N - 1, std::chrono::duration<double, std::micro>(t1 - t0).count() / reps,
           std::chrono::duration<double, std::micro>(t2 - t0).count() / reps);
}
int main()
{
    int *d;
    hipMalloc(&d, 4);
    run<64>(d);
    run<1024>(d);
    run<1600>(d);
    run<2048>(d);
    run<3072>(d);
    run<4096>(d);
    run<6144>(d);
    return 0;
}
