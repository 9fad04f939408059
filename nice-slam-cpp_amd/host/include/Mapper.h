// Same surface as the reference's include/Mapper.h:4-44: Mapper(), run(), optimize_map(), keyframe_selection_overlap(),
// get_mask_from_c2w().  The last two are thin methods over nsk_keyframe_overlap / nsk_frustum_mask (next rows N3 / N2); optimize_map
// calls the same entry points itself, except on levels where the caller installed its own mask with set_frustum_mask.
#pragma once
#include <iostream>
#include <memory>
#include "inputs/CoFusionReader.h"
#include <yaml-cpp/yaml.h>
#include "Renderer.h"
#include "nsk_host.h"
#include <algorithm>

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
    // src/Mapper.cpp:132-196: indices (into keyframe_vector_) of the k_overlap keyframes that see most of the current frame's samples
    void keyframe_selection_overlap(torch::Tensor gt_color_, torch::Tensor gt_depth_, torch::Tensor c2w, std::vector<KeyFrame> keyframe_vector_, int k_overlap,
                                    std::vector<int>& selected_kf);
    // src/Mapper.cpp:42-130: frustum mask of the level `key` ("grid_coarse" ...) for a depth image and pose; val_shape = (Z, Y, X) of the
    // level as the reference passes it (:258); mask comes back bool [X, Y, Z] like the reference's (its caller permutes it to Z,Y,X)
    void get_mask_from_c2w(cv::Mat depth_mat, torch::Tensor c2w, torch::Tensor val_shape, std::string key, torch::Tensor& mask);
    // not in the reference
    void set_frustum_mask(const std::string& grid_key, torch::Tensor mask_zyx);     // bool/uint8 [Z,Y,X]; undefined tensor = all
    void set_bound(torch::Tensor bound_3x2);
    void seed(uint64_t s) { rng_seed = s; draw_calls = 0; }
    // N > 1 (BASELINE configs[3], [4]): every rank draws the window's whole batch (the pixel draw is a counter hash of the seed: identical on
    // every rank, one small launch), renders the contiguous shard [lo, hi) of it with the batch's max(gt_depth) taken over the whole batch
    // (nsk_set_depth_max_batch: no collective), and ONE all-reduce per iteration sums the marked voxels' gradients, the colour decoder's, the
    // loss and -- in bundle-adjustment iterations -- the 8 floats per window frame of pose gradient (nsk_grad_extra); every rank then applies the
    // same Adam steps to grids, decoder and poses, so the replicas stay bit-identical without a broadcast.  All ranks must call seed() alike.
    void set_distributed(const nskh::Dist& d) { dist = d; }
    nskh::Dist dist;
    float last_kept_rays = 0.f;            // kept rays of the last bundle-adjustment iteration, summed over the ranks (N > 1; the shards must add up to the batch)
    // seed of the d-th pixel draw of this Mapper (d = 1, 2, ...: every keyframe_selection_overlap and every optimize_map call takes the next
    // one, so no two calls draw the same pixels -- the reference draws fresh torch::randint pixels every call, utils.h:19-36)
    static uint64_t draw_seed(uint64_t seed, uint64_t d) { return seed + 0xD1B54A32D192ED03ull * d; }
    float lr_factor;
    float last_loss = 0.f;
    std::vector<float> last_ba_grad;       // [window frames][7] pose gradients of the last bundle-adjustment step (zeros for frames it leaves fixed)
    std::vector<float> last_losses;        // loss of every iteration of the last optimize_map call (downloaded once, after the loop)
    int n_keyframes() const { return (int)keyframe_vector.size(); }
    torch::Tensor keyframe_est_c2w(int k) const { return keyframe_vector.at((size_t)k).est_c2w; }      // (bundle adjustment rewrites these, :467-489)
    std::vector<float> last_overlap;       // overlap fraction of keyframes [0, n-1) in that call (empty if not ranked)
    std::vector<int> last_window;          // keyframe indices of the last optimize_map call (-1 = current frame)
    double last_iter_us = 0.0;             // mean wall time of one iteration of the last optimize_map loop (stream-synchronised at its end)
    bool render_stage_literal_color = true;   // src/Mapper.cpp:430 renders the literal "color" whatever the stage (D19, reproduced by default);
                                              // false = render `stage`, the intended graph

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
    uint64_t rng_seed = 0, draw_calls = 0;
    struct Dev;                            // device-resident state of optimize_map: frame images, ray buffers, BA poses and their Adam moments
    std::shared_ptr<Dev> dev;
};
