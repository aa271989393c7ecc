// Test infrastructure (oracle/): the two ATen names the reference's dcnv3_im2col_cuda.cuh uses, so that its bilinear
// device functions can be compiled from where they lie under /root/reference without torch's CUDA headers.
#pragma once
namespace at { template <typename T> using opmath_type = T; }
