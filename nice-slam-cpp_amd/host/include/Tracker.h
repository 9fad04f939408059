// Same surface as the reference's include/Tracker.h:3-30 (it pulls in the dataset reader, as there), plus the overload that
// src/main.cpp:96 calls and the reference never declared: run(CoFusionReader&, NICE).
#pragma once
#include <iostream>
#include "inputs/CoFusionReader.h"
#include "inputs/SequenceReader.h"
#include "Renderer.h"
#include <yaml-cpp/yaml.h>

class Tracker {
  public:
    Tracker(YAML::Node ns_config, YAML::Node cf_config, c10::Dict<std::string, torch::Tensor> c_dict);
    virtual ~Tracker();
    void run(NICE decoders, torch::Tensor gt_color_t, torch::Tensor gt_depth_t, torch::Tensor gt_c2w_t, int idx);
    // src/main.cpp:96 (D1): every frame of the reader through the 5-argument form; frames_limit < 0 = until reader.hasMore() is false
    void run(CoFusionReader& reader, NICE decoders);
    // the same loop over a posed sequence (Replica / ScanNet / TUM layouts, inputs/SequenceReader.h: not in the reference); the reader's
    // pose of each frame is passed as gt_c2w
    void run(SequenceReader& reader, NICE decoders);
    int frames_limit = -1;
    torch::Tensor optimize_cam_in_batch(torch::Tensor cam_tensor, torch::Tensor gt_color, torch::Tensor gt_depth, int batch_size,
                                        torch::optim::Adam& optimizer, NICE decoders);
    void update_para_from_mapping();               // declared in the reference, never defined (D3): no-op here
    torch::Tensor last_camera_tensor;              // result of run() (the reference discards it)
    std::vector<float> last_losses;                // loss of every iteration of the last run() (the reference prints them)
    double last_run_us = 0.0;                      // wall time of the last run()'s iteration loop, stream-synchronised
    void seed(uint64_t s) { rng_seed = s; }
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
    uint64_t rng_seed = 0;
    struct Dev;                                    // device-resident state of run(): frame images, pose, Adam moments, ray buffers
    std::shared_ptr<Dev> dev;
};
