// Same surface as the reference's include/Tracker.h:8-30 minus the dataset reader include (out of scope).
#pragma once
#include <iostream>
#include <yaml-cpp/yaml.h>
#include "Renderer.h"

class Tracker {
  public:
    Tracker(YAML::Node ns_config, YAML::Node cf_config, c10::Dict<std::string, torch::Tensor> c_dict);
    virtual ~Tracker();
    void run(NICE decoders, torch::Tensor gt_color_t, torch::Tensor gt_depth_t, torch::Tensor gt_c2w_t, int idx);
    torch::Tensor optimize_cam_in_batch(torch::Tensor cam_tensor, torch::Tensor gt_color, torch::Tensor gt_depth, int batch_size,
                                        torch::optim::Adam& optimizer, NICE decoders);
    void update_para_from_mapping();               // declared in the reference, never defined (D3): no-op here
    torch::Tensor last_camera_tensor;              // result of run() (the reference discards it)
    void set_bound(torch::Tensor bound_3x2);

  private:
    int H, W;
    float fx, fy, cx, cy;
    int idx, ignore_edge_w, ignore_edge_h;
    torch::Tensor bound;
    Renderer renderer;
    c10::Dict<std::string, torch::Tensor> c;
    bool handle_dynamic, use_color_in_tracking;
    float w_color_loss;
    float lr;
    int num_cam_iters, tracking_pixels;
};
