// Same surface as the reference's include/models/MLP.h:7-38.  The modules own their parameters as ordinary registered
// torch parameters (D16: the reference forgot to register them); the arithmetic runs in the HIP kernels, which read the
// packed form produced by packed()/unpack().
#pragma once
#include <map>
#include <torch/torch.h>
#include "models/GaussianFFT.h"
#include "torchlib/utils.h"      // as the reference's include/models/MLP.h:3 does: src/main.cpp relies on its `using namespace torch::indexing`

struct MLP : torch::nn::Module {
    MLP(std::string name, int dim, int c_dim, int hidden_size, int n_blocks, bool color, std::vector<int> skips, float grid_len,
        std::string pose_emb, bool concat_feat);
    torch::Tensor forward(torch::Tensor p, std::map<std::string, torch::Tensor> c_grid);   // reference MLP.cpp:76-102
    torch::Tensor packed();                       // B, pts_linear[0..4].{w,b}, fc[0..4].{w,b}, output.{w,b}  (nsk.h)
    void unpack(const torch::Tensor& flat);

    std::string name;
    bool color, concat_feat;
    int c_dim, n_blocks;
    std::vector<int> skips;
    float grid_len;
    torch::nn::ModuleList fc, pts_linear;
    torch::nn::Linear output_linear{nullptr};
    std::shared_ptr<GaussianFFT> embedder;
    int embedding_size;
};

struct MLP_no_xyz : torch::nn::Module {
    MLP_no_xyz(std::string name, int dim, int c_dim, int hidden_size, int n_blocks, bool color, std::vector<int> skips, float grid_len);
    torch::Tensor forward(torch::Tensor p, std::map<std::string, torch::Tensor> c_grid);   // reference MLP.cpp:165-182
    torch::Tensor packed();
    void unpack(const torch::Tensor& flat);

    std::string name;
    bool color;
    int c_dim, n_blocks;
    std::vector<int> skips;
    float grid_len;
    torch::nn::ModuleList pts_linear;
    torch::nn::Linear output_linear{nullptr};
};
