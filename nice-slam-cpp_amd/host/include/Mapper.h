// Same surface as the reference's include/Mapper.h:11-44 for the hot path: Mapper(), run(), optimize_map().
// get_mask_from_c2w (next row N2) runs on the device (nsk_frustum_mask) when mapping.frustum_feature_selection is set, except on
// levels where the caller installed its own mask with set_frustum_mask.  keyframe_selection_overlap (next row N3) is folded into optimize_map too:
// the per-keyframe overlap fractions come from nsk_keyframe_overlap, the ranking stays on the host.
#pragma once
#include <algorithm>
#include <iostream>
#include <memory>
#include <yaml-cpp/yaml.h>
#include "Renderer.h"

struct KeyFrame {
    torch::Tensor cur_c2w, est_c2w, gt_c2w, gt_color, gt_depth, color, depth;
    int idx;
};

class Mapper {
  public:
    Mapper(YAML::Node ns_config, YAML::Node cf_config, bool coarse_mapper);
    virtual ~Mapper();
    void run(NICE& decoders, c10::Dict<std::string, torch::Tensor>& c, std::vector<torch::Tensor>& estimate_c2w_vec, torch::Tensor gt_color_t,
             torch::Tensor gt_depth_t, torch::Tensor gt_c2w_t, int idx, int n_imgs);
    void optimize_map(int num_joint_iters_, c10::Dict<std::string, torch::Tensor>& c_dict, torch::Tensor cur_gt_color, torch::Tensor cur_gt_depth,
                      torch::Tensor gt_cur_c2w, torch::Tensor& cur_c2w, NICE& decoders);
    // not in the reference
    void set_frustum_mask(const std::string& grid_key, torch::Tensor mask_zyx);     // bool/uint8 [Z,Y,X]; undefined tensor = all
    void set_bound(torch::Tensor bound_3x2);
    void seed(uint64_t s) { rng_seed = s; }
    float lr_factor;
    float last_loss = 0.f;
    std::vector<float> last_overlap;       // overlap fraction of keyframes [0, n-1) in that call (empty if not ranked)
    std::vector<int> last_window;          // keyframe indices of the last optimize_map call (-1 = current frame)

  private:
    Renderer renderer;
    YAML::Node ns_cfg, cf_cfg;
    bool color_refine, coarse_mapper, fix_color, frustum_feature_selection, BA;
    int mapping_window_size, mapping_pixels;
    float middle_iter_ratio, fine_iter_ratio;
    int H, W;
    float fx, fy, cx, cy;
    std::vector<KeyFrame> keyframe_vector;
    std::vector<int> keyframe_lvector;
    std::string keyframe_selection_method;
    torch::Tensor bound;
    bool fix_fine;
    int num_joint_iters, keyframe_every;
    std::string stage;
    float BA_cam_lr;
    float w_color_loss;
    bool first_frame = true;
    bool user_mask[4] = {false, false, false, false};
    uint64_t rng_seed = 0;
};
