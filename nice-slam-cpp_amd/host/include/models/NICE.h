// Same surface as the reference's include/models/NICE.h:5-12 (constructor and forward signatures unchanged).
// The four torch::jit modules of the reference (src/models/NICE.cpp:8-11, files absent from its repository, D5) are
// gone: the native decoders are the arithmetic.
#pragma once
#include <torch/torch.h>
#include "models/MLP.h"

struct NICE : torch::nn::Module {
    NICE(int dim, int c_dim, int hidden_size, float coarse_grid_len, float middle_grid_len, float fine_grid_len, float color_grid_len,
         bool coarse, std::string pose_emb);
    // raw [M,4] = (rgb, occupancy) for p [1,M,3] or [M,3] (src/models/NICE.cpp:16-52); runs the fused decoder kernels
    torch::Tensor forward(torch::Tensor p, c10::Dict<std::string, torch::Tensor> c_grid, std::string stage);

    std::shared_ptr<MLP> middle_decoder, fine_decoder, color_decoder;
    std::shared_ptr<MLP_no_xyz> coarse_decoder;

    // marshalling (not in the reference): push parameters to / pull them from the kernel context
    void sync_to_device();                         // uploads decoders whose parameters changed since the last call
    void fetch_from_device(bool fine, bool color); // after optimisation
    static void mark_dirty();                      // force re-upload (e.g. after modifying parameters through raw pointers)
  private:
    int64_t param_version(int which);
};
