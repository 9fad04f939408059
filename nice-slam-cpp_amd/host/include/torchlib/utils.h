// The hot-path helpers of the reference's include/torchlib/utils.h with the same names and argument order, minus its
// Eigen / OpenCV includes (neither is installed here, neither is needed by these functions).
#pragma once
#include <torch/torch.h>

using namespace torch::indexing;
namespace F = torch::nn::functional;

// utils.h:13-55 (intended semantics D10/D11): random pixel pick on the host (torch::randint), rays through c2w.
// pix_i/pix_j (int32 [n]) are additionally returned so that pose gradients can be formed.
void raySampler(int H0, int H1, int W0, int W1, int n, float fx, float fy, float cx, float cy, torch::Tensor depth, torch::Tensor color,
                torch::Tensor c2w, torch::Tensor& rays_o, torch::Tensor& rays_d, torch::Tensor& gt_color, torch::Tensor& gt_depth,
                torch::Tensor* pix_i = nullptr, torch::Tensor* pix_j = nullptr);
// utils.h:141-146
void get_samples(int H0, int H1, int W0, int W1, int n, int H, int W, float fx, float fy, float cx, float cy, torch::Tensor c2w,
                 torch::Tensor depth, torch::Tensor color, torch::Tensor& rays_o, torch::Tensor& rays_d, torch::Tensor& sample_depth,
                 torch::Tensor& sample_color, torch::Tensor* pix_i = nullptr, torch::Tensor* pix_j = nullptr);
// utils.h:148-172 (runs k_raw2outputs)
void raw2outputs_nerf_color(torch::Tensor raw, torch::Tensor z_vals, bool occupancy, torch::Tensor rays_d, torch::Tensor& rgb_map,
                            torch::Tensor& depth_map, torch::Tensor& depth_var, torch::Tensor& weights);
// utils.h:174-231 (D23: rotation -> quaternion of R, order (w,x,y,z))
torch::Tensor quad2rotation(torch::Tensor quad);
torch::Tensor get_camera_from_tensor(torch::Tensor inputs);
torch::Tensor get_tensor_from_camera(torch::Tensor RT, bool Tquad = false);
