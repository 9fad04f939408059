// Same surface as the reference's include/models/GaussianFFT.h:6-11.  forward() = sin(x B) (src/models/GaussianFFT.cpp:10-15)
// evaluated on the host with libtorch CPU ops -- the decoders call the fused kernels, not this module.
#pragma once
#include <torch/torch.h>

struct GaussianFFT : torch::nn::Module {
    GaussianFFT(int num_channels, int mapping_size, int scale);
    GaussianFFT();
    torch::Tensor forward(torch::Tensor x);
    torch::Tensor B;
};
