// opencorr_gpu.h -- the include name of the reference's binary CUDA module (gpu_lib/opencorr_gpu.h:31-101: Img2D / Img3D,
// ICGN2D1GPU / ICGN2D2GPU / ICGN3D1GPU).  The same classes live in oc_engines.h on top of the HIP C-ABI, so programs
// written for that module (examples/test_2d_dic_gpu_icgn.cpp, examples/test_dvc_gpu_icgn.cpp) build unchanged.
#pragma once

#include "oc_engines.h"
