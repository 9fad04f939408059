// slam_loop.cpp -- the frame loop BASELINE.json's K5 describes (Tracker on every frame, Mapper on every `every`-th, over a posed RGB-D
// sequence) through the C++ classes with the reference's surface and a SequenceReader; src/main.cpp wires up the Tracker only.
//   slam_loop <tum|replica|scannet> <sequence dir> <nice_slam.yaml> <cofusion.yaml> <out dir> [frames] [map every]
// <sequence dir>/bound.txt holds the scene bound (6 numbers: x0 x1 y0 y1 z0 z1; the reference hard-codes its own).  Writes est_poses.npy,
// gt_poses.npy [F,4,4], track_loss.npy [F] (last iteration's loss; 0 for frame 0), map_loss.npy [mapped frames].
// On a node (BASELINE configs[4], one process per GPU): NSK_RANK / NSK_WORLD / NSK_DEVICE (= local rank) / NSK_RCCL_ID_FILE (a path every rank
// can read; rank 0 publishes the ncclUniqueId there) in the environment of every process.  The Tracker runs on rank 0 only (200 rays: nothing
// to shard, and its median couples all rays, src/Tracker.cpp:70); its pose goes to every rank as a sum with zeros (8 floats); the Mapper's rays
// shard over the ranks with ONE all-reduce per iteration (Mapper::set_distributed).  Rank 0 writes the outputs.
#include <cstdio>
#include <cstdlib>
#include <fstream>

#include <hip/hip_runtime_api.h>

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
    f.write("\x93NUMPY\x01\x00", 8);
    uint16_t hl = (uint16_t)hdr.size();
    f.write((const char*)&hl, 2); f.write(hdr.data(), hdr.size()); f.write((const char*)t.data_ptr<float>(), t.numel() * sizeof(float));
}

int main(int argc, char** argv)
{
    if (argc < 6) { std::fprintf(stderr, "usage: slam_loop <tum|replica|scannet> <sequence dir> <nice_slam.yaml> <cofusion.yaml> <out dir> [frames] [map every]\n"); return 2; }
    const std::string kind = argv[1], seq = argv[2], out = std::string(argv[5]) + "/";
    const int max_frames = argc > 6 ? std::atoi(argv[6]) : -1, every = argc > 7 ? std::atoi(argv[7]) : 1;
    try {
        torch::manual_seed(11);
        YAML::Node ns = YAML::LoadFile(argv[3]), cf = YAML::LoadFile(argv[4]);
        SequenceReader reader(kind == "tum" ? SequenceReader::TUM : (kind == "replica" ? SequenceReader::Replica : SequenceReader::ScanNet), seq);
        float b[6];
        { std::ifstream bf(seq + "/bound.txt"); for (float& v : b) if (!(bf >> v)) { std::fprintf(stderr, "bound.txt: 6 numbers expected\n"); return 1; } }
        torch::Tensor bound = torch::tensor({{b[0], b[1]}, {b[2], b[3]}, {b[4], b[5]}});
        // grids as src/main.cpp:33-78 builds them: cells of grid_len.* over the bound (the coarse one over the enlarged bound), N(0, 0.01)
        c10::Dict<std::string, torch::Tensor> c;
        const float lens[4] = {ns["grid_len"]["coarse"].as<float>(), ns["grid_len"]["middle"].as<float>(), ns["grid_len"]["fine"].as<float>(), ns["grid_len"]["color"].as<float>()};
        const char* keys[4] = {"grid_coarse", "grid_middle", "grid_fine", "grid_color"};
        for (int l = 0; l < 4; ++l) {
            int64_t dims[3];
            for (int k = 0; k < 3; ++k) dims[k] = std::max<int64_t>(2, (int64_t)((b[2 * k + 1] - b[2 * k]) / lens[l]));
            c.insert(keys[l], torch::zeros({1, 32, dims[2], dims[1], dims[0]}).normal_(0, 0.01));
        }
        NICE decoders(3, 32, 32, lens[0], lens[1], lens[2], lens[3], true, "fourier");
        Tracker tracker(ns, cf, c);
        Mapper mapper(ns, cf, false);
        tracker.set_bound(bound); mapper.set_bound(bound);
        tracker.seed(100); mapper.seed(200);
        nskh::Dist dist;
        if (const char* w = std::getenv("NSK_WORLD")) {
            dist.world = std::atoi(w);
            dist.rank = std::getenv("NSK_RANK") ? std::atoi(std::getenv("NSK_RANK")) : 0;
            if (dist.world > 1) {
                const char* idf = std::getenv("NSK_RCCL_ID_FILE");
                if (!idf) { std::fprintf(stderr, "NSK_WORLD > 1 needs NSK_RCCL_ID_FILE\n"); return 2; }
                dist.comm = nskh::rccl_comm_from_file(dist.rank, dist.world, idf);
                mapper.set_distributed(dist);
            }
        }
        float* d_pose = nullptr;                                              // 8 floats on the device for the pose broadcast
        if (dist.on() && hipMalloc((void**)&d_pose, 8 * sizeof(float)) != hipSuccess) { std::fprintf(stderr, "hipMalloc failed\n"); return 1; }
        const int F = max_frames > 0 ? std::min(max_frames, reader.n_imgs) : reader.n_imgs;
        std::vector<torch::Tensor> est(F), gts(F);
        std::vector<float> tl, ml;
        for (int i = 0; i < F && reader.hasMore(); ++i) {
            reader.getNext();
            torch::Tensor depth_t = torch::from_blob(reader.depth.data, {reader.depth.rows, reader.depth.cols}, torch::kFloat32).clone();
            torch::Tensor color_t = torch::from_blob(reader.rgb.data, {reader.rgb.rows, reader.rgb.cols, 3}, torch::kFloat32).clone();
            torch::Tensor gt = torch::zeros({4, 4});
            for (int r = 0; r < 4; ++r) for (int q = 0; q < 4; ++q) gt[r][q] = reader.c2w(r, q);
            gts[i] = gt;
            if (i == 0) { est[i] = gt.clone(); tl.push_back(0.f); }          // the first pose is given (NICE-SLAM's convention)
            else {
                torch::Tensor cam7 = torch::zeros({8});
                if (dist.rank == 0) {
                    tracker.run(decoders, color_t, depth_t, est[i - 1], i);   // initialised from the previous estimate
                    cam7.index_put_({Slice(None, 7)}, tracker.last_camera_tensor);
                }
                if (dist.on()) {                                              // rank 0's pose to every rank
                    if (hipMemcpy(d_pose, cam7.data_ptr<float>(), 8 * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) throw std::runtime_error("H2D failed");
                    dist.broadcast0(d_pose, 8);
                    nskh::check(nsk_sync(nskh::ctx()));
                    if (hipMemcpy(cam7.data_ptr<float>(), d_pose, 8 * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) throw std::runtime_error("D2H failed");
                }
                torch::Tensor RT = get_camera_from_tensor(cam7.index({Slice(None, 7)}).contiguous());
                est[i] = torch::cat({RT, torch::tensor({{0.f, 0.f, 0.f, 1.f}})}, 0);
                tl.push_back(dist.rank == 0 && !tracker.last_losses.empty() ? tracker.last_losses.back() : 0.f);
            }
            for (int k = i + 1; k < F; ++k) if (!est[k].defined()) est[k] = est[i].clone();      // Mapper::run indexes the whole vector
            if (i % every == 0) {
                mapper.run(decoders, c, est, color_t, depth_t, gt, i, F);
                ml.push_back(mapper.last_loss);
            }
            std::printf("frame %d: track loss %.4f, |t_est - t_gt| %.4f m%s\n", i, tl.back(),
                        (est[i].index({Slice(None, 3), 3}) - gt.index({Slice(None, 3), 3})).norm().item<float>(), i % every == 0 ? ", mapped" : "");
            // device time of the frame's loops (stream-synchronised walls the classes keep): the Tracker's whole iteration loop, the Mapper's mean iteration
            if (i > 0 && dist.rank == 0) std::printf("time frame %d: tracker loop %.1f us\n", i, tracker.last_run_us);
            if (i % every == 0) std::printf("time frame %d: mapper iteration %.1f us\n", i, mapper.last_iter_us);
        }
        if (dist.rank != 0) { std::printf("slam_loop rank %d ok\n", dist.rank); return 0; }
        save_npy(out + "est_poses.npy", torch::stack(est)); save_npy(out + "gt_poses.npy", torch::stack(gts));
        save_npy(out + "track_loss.npy", torch::tensor(tl)); save_npy(out + "map_loss.npy", torch::tensor(ml));
        for (auto k : keys) save_npy(out + k + ".npy", c.at(k));
        std::printf("slam_loop ok\n");
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "slam_loop failed: %s\n", e.what());
        return 1;
    }
}
