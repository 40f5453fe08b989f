// See cuda_runtime.h next to this file: main.cu:7 includes this header and uses nothing from it.
#pragma once
