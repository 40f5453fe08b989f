// main.cu of the reference includes <cuda_runtime.h> (main.cu:6) but uses nothing from it -- no runtime call, no
// launch, no CUDA type: all device work sits behind the gpu:: / cpu:: functions of libofx_hip.so.  This file only lets
// that include line resolve when main.cu is compiled against this repo's include/ directory; it declares nothing.
#pragma once
