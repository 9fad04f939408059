// Same surface as the reference's include/Renderer.h:8-25: constructor, eval_points and render_batch_ray keep their
// signatures (note rays_d before rays_o).  Every call marshals into include/nsk.h.
#pragma once
#include <torch/torch.h>
#include "models/NICE.h"

class Renderer {
  public:
    Renderer();
    torch::Tensor eval_points(torch::Tensor p, NICE decoders, c10::Dict<std::string, torch::Tensor> c, std::string stage);
    void render_batch_ray(c10::Dict<std::string, torch::Tensor> c, NICE decoders, torch::Tensor rays_d, torch::Tensor rays_o,
                          std::string stage, torch::Tensor gt_depth, torch::Tensor& rgb_map, torch::Tensor& depth_map,
                          torch::Tensor& depth_var, torch::Tensor& weights);
    // not in the reference: the scene bound is hard-coded there in five places (src/Renderer.cpp:15 ...)
    void set_bound(torch::Tensor bound_3x2);
    torch::Tensor bound;
    int N_samples, N_surface, N_importance;
    bool lindisp, occupancy;
    float perturb;

  private:
    void push_opts();
    int points_batch_size, ray_batch_size;
    int scale;
};
