// host_test.cpp -- exercises the C++ host classes (reference surface) end to end on the GPU and dumps inputs / outputs
// as .npy files for tests/test_gpu_host_cpp.py, which checks them against the CPU oracle.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

#include "Mapper.h"
#include "Tracker.h"
#include "nsk_host.h"
#include "torchlib/utils.h"

static void save_npy(const std::string& path, torch::Tensor t)
{
    t = t.detach().to(torch::kCPU, torch::kFloat32).contiguous();
    std::ostringstream shape;
    shape << "(";
    for (int64_t i = 0; i < t.dim(); ++i) shape << t.size(i) << (t.dim() == 1 || i + 1 < t.dim() ? "," : "");
    shape << ")";
    std::string hdr = "{'descr': '<f4', 'fortran_order': False, 'shape': " + shape.str() + ", }";
    while ((10 + hdr.size() + 1) % 64 != 0) hdr += ' ';
    hdr += '\n';
    std::ofstream f(path, std::ios::binary);
    const char magic[] = "\x93NUMPY\x01\x00";
    f.write(magic, 8);
    uint16_t hl = (uint16_t)hdr.size();
    f.write((const char*)&hl, 2);
    f.write(hdr.data(), hdr.size());
    f.write((const char*)t.data_ptr<float>(), t.numel() * sizeof(float));
}

static const char* NS_YAML =                      // only the keys the drivers consume, our own values
    "coarse: True\n"
    "tracking:\n  ignore_edge_W: 4\n  ignore_edge_H: 4\n  use_color_in_tracking: True\n  handle_dynamic: True\n  w_color_loss: 0.5\n"
    "  lr: 0.001\n  pixels: 100\n  iters: 3\n"
    "mapping:\n  color_refine: True\n  middle_iter_ratio: 0.4\n  fine_iter_ratio: 0.6\n  BA: True\n  BA_cam_lr: 0.001\n  fix_fine: True\n"
    "  fix_color: False\n  keyframe_every: 50\n  mapping_window_size: 5\n  w_color_loss: 0.2\n  frustum_feature_selection: True\n"
    "  keyframe_selection_method: 'overlap'\n  lr_first_factor: 5\n  lr_factor: 1\n  pixels: 200\n  iters_first: 10\n  iters: 5\n"
    "  stage:\n"
    "    coarse:\n      decoders_lr: 0.0\n      coarse_lr: 0.001\n      middle_lr: 0.0\n      fine_lr: 0.0\n      color_lr: 0.0\n"
    "    middle:\n      decoders_lr: 0.0\n      coarse_lr: 0.0\n      middle_lr: 0.1\n      fine_lr: 0.0\n      color_lr: 0.0\n"
    "    fine:\n      decoders_lr: 0.0\n      coarse_lr: 0.0\n      middle_lr: 0.005\n      fine_lr: 0.005\n      color_lr: 0.0\n"
    "    color:\n      decoders_lr: 0.005\n      coarse_lr: 0.0\n      middle_lr: 0.005\n      fine_lr: 0.005\n      color_lr: 0.005\n";
static const char* CF_YAML =
    "mapping:\n  pixels: 200   # per mapping iteration\n"
    "cam:\n  H: 48\n  W: 64\n  fx: 40.0\n  fy: 40.0\n  cx: 32.0\n  cy: 24.0\n";

int main(int argc, char** argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: host_test <out_dir>\n"); return 2; }
    const std::string out = std::string(argv[1]) + "/";
    try {
        torch::manual_seed(7);
        std::istringstream ns_s(NS_YAML), cf_s(CF_YAML);
        YAML::Node ns = YAML::Load(ns_s), cf = YAML::Load(cf_s);
        if (ns["mapping"]["stage"]["color"]["fine_lr"].as<float>() != 0.005f || !ns["coarse"].as<bool>() || cf["cam"]["W"].as<int>() != 64) {
            std::fprintf(stderr, "yaml shim failed\n"); return 1;
        }
        torch::Tensor bound = torch::tensor({{-4.5, 3.82}, {-1.5, 2.02}, {-3.0, 2.76}});
        c10::Dict<std::string, torch::Tensor> c;
        c.insert("grid_coarse", torch::zeros({1, 32, 3, 2, 4}).normal_(0, 0.3));
        c.insert("grid_middle", torch::zeros({1, 32, 6, 5, 7}).normal_(0, 0.3));
        c.insert("grid_fine", torch::zeros({1, 32, 9, 8, 11}).normal_(0, 0.3));
        c.insert("grid_color", torch::zeros({1, 32, 9, 8, 11}).normal_(0, 0.3));
        NICE decoders(3, 32, 32, 2.f, 0.32f, 0.16f, 0.16f, true, "fourier");
        save_npy(out + "bound.npy", bound);
        for (auto k : {"grid_coarse", "grid_middle", "grid_fine", "grid_color"}) save_npy(out + k + ".npy", c.at(k));
        save_npy(out + "dec_coarse.npy", decoders.coarse_decoder->packed());
        save_npy(out + "dec_middle.npy", decoders.middle_decoder->packed());
        save_npy(out + "dec_fine.npy", decoders.fine_decoder->packed());
        save_npy(out + "dec_color.npy", decoders.color_decoder->packed());

        // ---- a synthetic frame: camera in the room centre looking along -z, depth = z-depth of the room walls
        const int H = 48, W = 64; const float fx = 40, fy = 40, cx = 32, cy = 24;
        torch::Tensor c2w = torch::eye(4);
        c2w.index_put_({Slice(None, 3), 3}, torch::tensor({-0.3f, 0.2f, 0.1f}));
        auto jj = torch::arange(H).to(torch::kFloat32).unsqueeze(1).expand({H, W});
        auto ii = torch::arange(W).to(torch::kFloat32).unsqueeze(0).expand({H, W});
        auto dirs = torch::stack({(ii - cx) / fx, -(jj - cy) / fy, -torch::ones({H, W})}, -1).reshape({-1, 3});
        auto o3 = c2w.index({Slice(None, 3), 3}).unsqueeze(0).expand({H * W, 3});
        torch::Tensor room = bound.clone();
        room.index_put_({Slice(), 0}, room.index({Slice(), 0}) + 0.3);
        room.index_put_({Slice(), 1}, room.index({Slice(), 1}) - 0.3);
        auto t_ = (room.unsqueeze(0) - o3.unsqueeze(-1)) / dirs.unsqueeze(-1);
        torch::Tensor depth_img = std::get<0>(torch::min(std::get<0>(torch::max(t_, 2)), 1)).reshape({H, W}).contiguous();
        torch::Tensor hit = o3 + dirs * depth_img.reshape({-1, 1});
        torch::Tensor color_img = (0.5 + 0.5 * torch::sin(hit * torch::tensor({1.3f, 2.1f, 0.7f}))).reshape({H, W, 3}).contiguous();

        // ---- Renderer::render_batch_ray / eval_points / raw2outputs
        Renderer renderer;
        renderer.set_bound(bound);
        torch::Tensor rays_o, rays_d, gt_color, gt_depth;
        get_samples(0, H, 0, W, 150, H, W, fx, fy, cx, cy, c2w, depth_img, color_img, rays_o, rays_d, gt_depth, gt_color);
        gt_depth.index_put_({Slice(None, 7)}, 0.0);                       // exercise the zero-depth branch (Renderer.cpp:94-98)
        torch::Tensor rgb, depth, var, weights;
        renderer.render_batch_ray(c, decoders, rays_d, rays_o, "color", gt_depth, rgb, depth, var, weights);
        save_npy(out + "rays_o.npy", rays_o); save_npy(out + "rays_d.npy", rays_d); save_npy(out + "gt_depth.npy", gt_depth);
        save_npy(out + "r_rgb.npy", rgb); save_npy(out + "r_depth.npy", depth); save_npy(out + "r_var.npy", var); save_npy(out + "r_weights.npy", weights);
        torch::Tensor rgb2, depth2, var2, w2;
        renderer.render_batch_ray(c, decoders, rays_d, rays_o, "middle", torch::Tensor(), rgb2, depth2, var2, w2);   // no gt depth (Renderer.cpp:54-57)
        save_npy(out + "r2_depth.npy", depth2); save_npy(out + "r2_weights.npy", w2);
        torch::Tensor pts = rays_o + rays_d * 1.5;
        torch::Tensor raw = renderer.eval_points(pts, decoders, c, "color");
        save_npy(out + "pts.npy", pts); save_npy(out + "raw.npy", raw);
        torch::Tensor raw_fwd = decoders.forward(pts.unsqueeze(0), c, "fine");                                        // NICE::forward [1,M,3]
        save_npy(out + "raw_fine.npy", raw_fwd);

        // ---- pose helpers (utils.h:174-231)
        torch::Tensor cam = torch::tensor({0.9f, 0.1f, -0.2f, 0.3f, 0.5f, -0.4f, 0.3f});
        torch::Tensor RT = get_camera_from_tensor(cam);
        torch::Tensor cam_back = get_tensor_from_camera(RT, false);
        torch::Tensor qn = cam.index({Slice(None, 4)}) / cam.index({Slice(None, 4)}).norm();
        if ((cam_back.index({Slice(None, 4)}) - qn).abs().max().item<float>() > 1e-5f || (cam_back.index({Slice(4, None)}) - cam.index({Slice(4, None)})).abs().max().item<float>() > 1e-6f) {
            std::fprintf(stderr, "get_tensor_from_camera does not invert get_camera_from_tensor\n"); return 1;
        }
        save_npy(out + "cam.npy", cam); save_npy(out + "cam_RT.npy", RT);

        // ---- Tracker::optimize_cam_in_batch / run
        Tracker tracker(ns, cf, c);
        tracker.set_bound(bound);
        torch::Tensor cam_t = get_tensor_from_camera(c2w, false).clone();
        cam_t.index_put_({Slice(4, None)}, cam_t.index({Slice(4, None)}) + torch::tensor({0.03f, -0.02f, 0.02f}));
        save_npy(out + "trk_cam0.npy", cam_t);
        cam_t.requires_grad_(true);
        std::vector<torch::Tensor> pl{cam_t};
        torch::optim::Adam opt(pl, torch::optim::AdamOptions(1e-2));
        torch::Tensor l0 = tracker.optimize_cam_in_batch(cam_t, color_img, depth_img, 100, opt, decoders);
        torch::Tensor l1 = tracker.optimize_cam_in_batch(cam_t, color_img, depth_img, 100, opt, decoders);
        save_npy(out + "trk_cam2.npy", cam_t); save_npy(out + "trk_loss.npy", torch::stack({l0, l1}));
        tracker.seed(1000);                                                // iteration i of run() draws its pixels with seed 1000 + i
        tracker.run(decoders, color_img, depth_img, c2w, 0);
        save_npy(out + "trk_run_cam.npy", tracker.last_camera_tensor);
        save_npy(out + "trk_run_losses.npy", torch::tensor(tracker.last_losses));
        save_npy(out + "trk_color_img.npy", color_img);
        tracker.run(decoders, color_img, depth_img, c2w, 1);               // a second frame: buffers and images are reused, nothing is reallocated
        std::printf("Tracker::run device-resident: %.1f us per iteration (%d iterations, %d rays)\n", tracker.last_run_us / 3.0, 3, 100);

        // ---- frame identity: the device copy of a frame is keyed by its host tensors; a NEW frame must be uploaded even when the allocator hands
        // its tensors the addresses the previous frame's (freed) tensors had -- the result for frame B must be that of a Tracker that never saw A
        {
            Tracker ta(ns, cf, c); ta.set_bound(bound); ta.seed(500);
            { torch::Tensor dA = depth_img.clone(), cA = color_img.clone(); ta.run(decoders, cA, dA, c2w, 0); }      // A's tensors die here
            const float loss_a = ta.last_losses[0];
            torch::Tensor dB = (depth_img * 0.9f).clone(), cB = (color_img * 0.5f).clone();                            // same sizes: the freed blocks are the first candidates
            ta.seed(500); ta.run(decoders, cB, dB, c2w, 1);
            Tracker tb(ns, cf, c); tb.set_bound(bound); tb.seed(500);
            tb.run(decoders, cB, dB, c2w, 0);
            save_npy(out + "frame_identity.npy", torch::tensor({loss_a, ta.last_losses[0], tb.last_losses[0]}));
        }

        // ---- Mapper::run (first frame: iters_first, lr_first_factor) then a second frame with a keyframe in the window
        Mapper mapper(ns, cf, false);
        mapper.set_bound(bound);
        torch::Tensor fmask = torch::rand({9, 8, 11}) < 0.8;
        mapper.set_frustum_mask("grid_fine", fmask);
        save_npy(out + "map_fine_mask.npy", fmask.to(torch::kFloat32));
        save_npy(out + "map_depth_img.npy", depth_img); save_npy(out + "map_c2w.npy", c2w);
        std::vector<torch::Tensor> est{c2w.clone(), c2w.clone()};
        torch::Tensor color_before = decoders.color_decoder->packed().clone(), fine_before = decoders.fine_decoder->packed().clone();
        mapper.run(decoders, c, est, color_img, depth_img, c2w, 0, 10);
        float loss_a = mapper.last_loss;
        mapper.run(decoders, c, est, color_img, depth_img, c2w, 1, 10);
        float loss_b = mapper.last_loss;
        std::printf("Mapper::optimize_map device-resident: %.1f us per iteration (200 rays over the window)\n", mapper.last_iter_us);
        save_npy(out + "map_loss.npy", torch::tensor({loss_a, loss_b}));
        // the thin methods of the reference's surface (include/Mapper.h:24-25)
        {
            cv::Mat depth_mat(H, W, CV_32FC1);
            std::memcpy(depth_mat.data, depth_img.data_ptr<float>(), sizeof(float) * H * W);
            torch::Tensor m;
            mapper.get_mask_from_c2w(depth_mat, c2w, torch::tensor({(int64_t)6, (int64_t)5, (int64_t)7}), "grid_middle", m);      // val_shape = (Z,Y,X) (:258)
            save_npy(out + "thin_mask_middle_xyz.npy", m.to(torch::kFloat32));
            KeyFrame k0; k0.est_c2w = c2w.clone();
            KeyFrame k1; k1.est_c2w = c2w.clone(); k1.est_c2w[0][0] = -1.f; k1.est_c2w[2][2] = -1.f;      // looks the other way: no overlap
            std::vector<int> sel;
            mapper.keyframe_selection_overlap(color_img, depth_img, c2w, std::vector<KeyFrame>{k1, k0}, 3, sel);
            torch::Tensor st = torch::zeros({(int64_t)sel.size()});
            for (size_t k = 0; k < sel.size(); ++k) st[k] = (float)sel[k];
            save_npy(out + "thin_selected.npy", st.numel() ? st : torch::full({1}, -7.f));
            save_npy(out + "thin_overlap.npy", torch::tensor(mapper.last_overlap));
        }
        for (auto k : {"grid_middle", "grid_fine", "grid_color"}) save_npy(out + "map_" + k + ".npy", c.at(k));
        save_npy(out + "map_dec_color_delta.npy", (decoders.color_decoder->packed() - color_before).abs().max().reshape({1}));
        save_npy(out + "map_dec_fine_delta.npy", (decoders.fine_decoder->packed() - fine_before).abs().max().reshape({1}));
        // ---- keyframe window by overlap (Mapper.cpp:132-216): five keyframes with very different views of the current frame
        {
            std::string ns2 = NS_YAML;
            ns2.replace(ns2.find("keyframe_every: 50"), 18, "keyframe_every: 1 ");
            std::istringstream ns2_s(ns2);
            YAML::Node nsb = YAML::Load(ns2_s);
            Mapper mapper2(nsb, cf, false);
            mapper2.set_bound(bound);
            auto roty = [&](float a) { torch::Tensor m = torch::eye(4); m[0][0] = std::cos(a); m[0][2] = std::sin(a); m[2][0] = -std::sin(a); m[2][2] = std::cos(a); return m; };
            std::vector<torch::Tensor> poses;
            for (float a : {0.0f, 3.14159f, 0.65f, 0.78f, 0.25f, 0.0f}) {          // same view, opposite, 37 deg, 45 deg, 14 deg; frame 5 = current
                torch::Tensor p = torch::matmul(c2w.clone(), roty(a));
                p.index_put_({Slice(None, 3), 3}, c2w.index({Slice(None, 3), 3}));
                poses.push_back(p);
            }
            c10::Dict<std::string, torch::Tensor> c2;
            for (auto k : {"grid_coarse", "grid_middle", "grid_fine", "grid_color"}) c2.insert(k, c.at(k).clone());
            for (int idx = 0; idx < 6; ++idx) mapper2.run(decoders, c2, poses, color_img, depth_img, poses[idx], idx, 100);
            torch::Tensor win = torch::zeros({(int64_t)mapper2.last_window.size()});
            for (size_t k = 0; k < mapper2.last_window.size(); ++k) win[k] = (float)mapper2.last_window[k];
            save_npy(out + "kf_window.npy", win);
            save_npy(out + "kf_overlap.npy", torch::tensor(mapper2.last_overlap));
            save_npy(out + "kf_poses.npy", torch::stack(poses));
        }
        // ---- Mapper::run against the oracle (tests/test_gpu_host_cpp.py::test_mapper_run_matches_oracle_on_the_same_pixel_draws): six frames,
        // every one a keyframe, so the last one optimises with bundle adjustment (more than four keyframes, :530); the stage schedule,
        // lr_factor, overlap window, frustum masks, pixs_per_image and the BA pose update all come from the class.  Own grids and decoders.
        {
            std::string ns3 = NS_YAML;
            ns3.replace(ns3.find("keyframe_every: 50"), 18, "keyframe_every: 1 ");
            ns3.replace(ns3.find("lr_first_factor: 5"), 18, "lr_first_factor: 2");
            ns3.replace(ns3.find("iters_first: 10"), 15, "iters_first: 4 ");
            ns3.replace(ns3.find("  iters: 5\n  stage"), 10, "  iters: 3");
            std::istringstream ns3_s(ns3);
            YAML::Node nsc = YAML::Load(ns3_s);
            if (nsc["mapping"]["iters"].as<int>() != 3 || nsc["mapping"]["iters_first"].as<int>() != 4 || nsc["tracking"]["iters"].as<int>() != 3) { std::fprintf(stderr, "mo yaml edit failed\n"); return 1; }
            torch::manual_seed(21);
            c10::Dict<std::string, torch::Tensor> c3;
            c3.insert("grid_coarse", torch::zeros({1, 32, 3, 2, 4}).normal_(0, 0.3));
            c3.insert("grid_middle", torch::zeros({1, 32, 6, 5, 7}).normal_(0, 0.3));
            c3.insert("grid_fine", torch::zeros({1, 32, 9, 8, 11}).normal_(0, 0.3));
            c3.insert("grid_color", torch::zeros({1, 32, 9, 8, 11}).normal_(0, 0.3));
            NICE dec3(3, 32, 32, 2.f, 0.32f, 0.16f, 0.16f, true, "fourier");
            for (auto k : {"grid_coarse", "grid_middle", "grid_fine", "grid_color"}) save_npy(out + "mo_" + k + "_0.npy", c3.at(k));
            save_npy(out + "mo_dec_coarse.npy", dec3.coarse_decoder->packed()); save_npy(out + "mo_dec_middle.npy", dec3.middle_decoder->packed());
            save_npy(out + "mo_dec_fine.npy", dec3.fine_decoder->packed()); save_npy(out + "mo_dec_color_0.npy", dec3.color_decoder->packed());
            Mapper mo(nsc, cf, false);
            mo.set_bound(bound);
            mo.seed(4321);
            auto roty = [&](float a) { torch::Tensor m = torch::eye(4); m[0][0] = std::cos(a); m[0][2] = std::sin(a); m[2][0] = -std::sin(a); m[2][2] = std::cos(a); return m; };
            std::vector<torch::Tensor> est3;
            for (float a : {0.0f, 0.10f, -0.08f, 0.15f, 0.05f, -0.04f}) {            // six nearby views of the same room (every window frame overlaps)
                torch::Tensor p3 = torch::matmul(c2w.clone(), roty(a));
                p3.index_put_({Slice(None, 3), 3}, c2w.index({Slice(None, 3), 3}) + torch::tensor({0.05f * a, 0.f, -0.1f * a}));
                est3.push_back(p3);
            }
            save_npy(out + "mo_poses_0.npy", torch::stack(est3));
            torch::Tensor losses = torch::full({6, 4}, -1.f), windows = torch::full({6, 6}, -9.f);
            for (int idx = 0; idx < 6; ++idx) {
                mo.run(dec3, c3, est3, color_img, depth_img, est3[idx], idx, 100);
                for (size_t k = 0; k < mo.last_losses.size(); ++k) losses[idx][(int64_t)k] = mo.last_losses[k];
                for (size_t k = 0; k < mo.last_window.size(); ++k) windows[idx][(int64_t)k] = (float)mo.last_window[k];
            }
            save_npy(out + "mo_losses.npy", losses); save_npy(out + "mo_windows.npy", windows);
            save_npy(out + "mo_ba_grad.npy", torch::tensor(mo.last_ba_grad).reshape({-1, 7}));      // of the last frame's (only) BA step
            for (auto k : {"grid_middle", "grid_fine", "grid_color"}) save_npy(out + "mo_" + k + "_1.npy", c3.at(k));
            save_npy(out + "mo_dec_color_1.npy", dec3.color_decoder->packed());
            save_npy(out + "mo_poses_1.npy", torch::stack(est3));                      // estimate_c2w_vec: the BA frame's entry is rewritten (:533)
            std::vector<torch::Tensor> kfp;
            for (int k = 0; k < mo.n_keyframes(); ++k) kfp.push_back(mo.keyframe_est_c2w(k));
            save_npy(out + "mo_kf_poses_1.npy", torch::stack(kfp));
        }
        // the renderer must see the optimised grids / decoder without any explicit upload
        renderer.render_batch_ray(c, decoders, rays_d, rays_o, "color", gt_depth, rgb, depth, var, weights);
        save_npy(out + "r3_depth.npy", depth); save_npy(out + "r3_rgb.npy", rgb);
        save_npy(out + "dec_color_after.npy", decoders.color_decoder->packed());
        std::printf("host_test ok\n");
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "host_test failed: %s\n", e.what());
        return 1;
    }
}
