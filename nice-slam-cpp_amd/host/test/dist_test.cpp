// dist_test.cpp -- the C++ Mapper with rays sharded over `world` ranks (Mapper::set_distributed, BASELINE configs[3] / [4]): six frames of
// Mapper::run, every one a keyframe, so the sixth optimises with bundle adjustment (src/Mapper.cpp:305-329,366-368,467-489, :530).
//     dist_test <out_dir> <rank> <world> [<shm_file>]
// world = 1: the single-process run.  world > 1: one process per rank, all on this box's one GPU (RCCL refuses two ranks on a device), the
// exchange through `shm_file` (a zero-filled file in /dev/shm the test created): every rank copies its packed buffer into its slot, all
// ranks add the slots in rank order, so every rank ends with the same bits.  tests/test_gpu_host_cpp.py compares the dumps.
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <thread>

#include <hip/hip_runtime_api.h>

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

// ---- the exchange of the test: host-staged sum through shared memory ----------------------------------------------------------------
struct Shm {
    struct Head { std::atomic<int> count, gen; };
    static constexpr size_t SLOT = 4u << 20;                  // floats per rank
    Head* head = nullptr; float* slots = nullptr; int rank = 0, world = 1;
    void open(const char* path, int r, int w)
    {
        rank = r; world = w;
        const size_t bytes = 4096 + (size_t)w * SLOT * sizeof(float);
        int fd = ::open(path, O_RDWR);
        if (fd < 0) throw std::runtime_error(std::string("cannot open ") + path);
        void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        ::close(fd);
        if (p == MAP_FAILED) throw std::runtime_error("mmap failed (the test must create the file at its full size)");
        head = reinterpret_cast<Head*>(p); slots = reinterpret_cast<float*>((char*)p + 4096);
    }
    void barrier()
    {
        const int g = head->gen.load();
        if (head->count.fetch_add(1) + 1 == world) { head->count.store(0); head->gen.fetch_add(1); return; }
        for (long spins = 0; head->gen.load() == g; ++spins) {
            if (spins > 600000) throw std::runtime_error("dist_test: a rank did not reach the barrier within 60 s");
            std::this_thread::sleep_for(std::chrono::microseconds(100));
        }
    }
    static void sum(float* d_buf, size_t n, void* user)
    {
        Shm& S = *reinterpret_cast<Shm*>(user);
        if (n > SLOT) throw std::runtime_error("dist_test: exchange buffer larger than the shared slot");
        nskh::check(nsk_sync(nskh::ctx()));
        float* mine = S.slots + (size_t)S.rank * SLOT;
        if (hipMemcpy(mine, d_buf, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) throw std::runtime_error("D2H failed");
        S.barrier();
        std::vector<float> acc(n, 0.f);
        for (int r = 0; r < S.world; ++r) { const float* s = S.slots + (size_t)r * SLOT; for (size_t i = 0; i < n; ++i) acc[i] += s[i]; }      // rank order on every rank
        S.barrier();                                            // nobody overwrites a slot that is still being read
        if (hipMemcpy(d_buf, acc.data(), n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) throw std::runtime_error("H2D failed");
    }
};

static const char* NS_YAML =
    "coarse: True\n"
    "tracking:\n  ignore_edge_W: 4\n  ignore_edge_H: 4\n  use_color_in_tracking: True\n  handle_dynamic: True\n  w_color_loss: 0.5\n"
    "  lr: 0.001\n  pixels: 100\n  iters: 3\n"
    "mapping:\n  color_refine: True\n  middle_iter_ratio: 0.4\n  fine_iter_ratio: 0.6\n  BA: True\n  BA_cam_lr: 0.001\n  fix_fine: True\n"
    "  fix_color: False\n  keyframe_every: 1\n  mapping_window_size: 5\n  w_color_loss: 0.2\n  frustum_feature_selection: True\n"
    "  keyframe_selection_method: 'overlap'\n  lr_first_factor: 2\n  lr_factor: 1\n  pixels: 200\n  iters_first: 4\n  iters: 3\n"
    "  stage:\n"
    "    coarse:\n      decoders_lr: 0.0\n      coarse_lr: 0.001\n      middle_lr: 0.0\n      fine_lr: 0.0\n      color_lr: 0.0\n"
    "    middle:\n      decoders_lr: 0.0\n      coarse_lr: 0.0\n      middle_lr: 0.1\n      fine_lr: 0.0\n      color_lr: 0.0\n"
    "    fine:\n      decoders_lr: 0.0\n      coarse_lr: 0.0\n      middle_lr: 0.005\n      fine_lr: 0.005\n      color_lr: 0.0\n"
    "    color:\n      decoders_lr: 0.005\n      coarse_lr: 0.0\n      middle_lr: 0.005\n      fine_lr: 0.005\n      color_lr: 0.005\n";
static const char* CF_YAML =
    "mapping:\n  pixels: 203   # per mapping iteration (not a multiple of the window or of the ranks: uneven shards, frames split across ranks)\n"
    "cam:\n  H: 48\n  W: 64\n  fx: 40.0\n  fy: 40.0\n  cx: 32.0\n  cy: 24.0\n";

int main(int argc, char** argv)
{
    if (argc < 4) { std::fprintf(stderr, "usage: dist_test <out_dir> <rank> <world> [<shm_file>]\n"); return 2; }
    const int rank = std::atoi(argv[2]), world = std::atoi(argv[3]);
    const std::string out = std::string(argv[1]) + "/r" + std::to_string(rank) + "_";
    try {
        Shm shm;
        nskh::Dist dist;
        dist.rank = rank; dist.world = world;
        if (world > 1) {
            if (argc < 5) { std::fprintf(stderr, "world > 1 needs the shared-memory file\n"); return 2; }
            shm.open(argv[4], rank, world);
            dist.hook = &Shm::sum; dist.user = &shm;
        }
        torch::manual_seed(21);                                 // every rank builds the same grids and decoders
        std::istringstream ns_s(NS_YAML), cf_s(CF_YAML);
        YAML::Node ns = YAML::Load(ns_s), cf = YAML::Load(cf_s);
        torch::Tensor bound = torch::tensor({{-4.5, 3.82}, {-1.5, 2.02}, {-3.0, 2.76}});
        c10::Dict<std::string, torch::Tensor> c;
        c.insert("grid_coarse", torch::zeros({1, 32, 3, 2, 4}).normal_(0, 0.3));
        c.insert("grid_middle", torch::zeros({1, 32, 6, 5, 7}).normal_(0, 0.3));
        c.insert("grid_fine", torch::zeros({1, 32, 9, 8, 11}).normal_(0, 0.3));
        c.insert("grid_color", torch::zeros({1, 32, 9, 8, 11}).normal_(0, 0.3));
        NICE dec(3, 32, 32, 2.f, 0.32f, 0.16f, 0.16f, true, "fourier");
        // a synthetic frame: camera in the room looking along -z, depth = distance to the room's walls along the pixel ray's z
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

        Mapper mo(ns, cf, false);
        mo.set_bound(bound);
        mo.seed(4321);
        mo.set_distributed(dist);
        auto roty = [&](float a) { torch::Tensor m = torch::eye(4); m[0][0] = std::cos(a); m[0][2] = std::sin(a); m[2][0] = -std::sin(a); m[2][2] = std::cos(a); return m; };
        std::vector<torch::Tensor> est;
        for (float a : {0.0f, 0.10f, -0.08f, 0.15f, 0.05f, -0.04f}) {
            torch::Tensor p = torch::matmul(c2w.clone(), roty(a));
            p.index_put_({Slice(None, 3), 3}, c2w.index({Slice(None, 3), 3}) + torch::tensor({0.05f * a, 0.f, -0.1f * a}));
            est.push_back(p);
        }
        // The Tracker's side of the loop at N > 1 (slam_loop.cpp): it runs on rank 0 only (its median couples all rays, src/Tracker.cpp:70) and its
        // pose reaches every rank as a sum with zeros -- Dist::broadcast0, here through the same exchange hook the Mapper's all-reduce takes.
        {
            torch::Tensor cam7 = torch::full({8}, 777.f);                     // what a rank holds before the exchange must not matter
            if (rank == 0) {
                Tracker tr(ns, cf, c);
                tr.set_bound(bound); tr.seed(99);
                tr.run(dec, color_img, depth_img, est[1], 1);
                cam7.zero_();
                cam7.index_put_({Slice(None, 7)}, tr.last_camera_tensor);
            }
            if (dist.on()) {
                float* d_pose = nullptr;
                if (hipMalloc((void**)&d_pose, 8 * sizeof(float)) != hipSuccess) throw std::runtime_error("hipMalloc failed");
                if (hipMemcpy(d_pose, cam7.data_ptr<float>(), 8 * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) throw std::runtime_error("H2D failed");
                dist.broadcast0(d_pose, 8);
                nskh::check(nsk_sync(nskh::ctx()));
                if (hipMemcpy(cam7.data_ptr<float>(), d_pose, 8 * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) throw std::runtime_error("D2H failed");
                (void)hipFree(d_pose);
            }
            save_npy(out + "tracked.npy", cam7);
        }
        torch::Tensor losses = torch::full({6, 4}, -1.f);
        for (int idx = 0; idx < 6; ++idx) {
            mo.run(dec, c, est, color_img, depth_img, est[idx], idx, 100);
            for (size_t k = 0; k < mo.last_losses.size(); ++k) losses[idx][(int64_t)k] = mo.last_losses[k];
        }
        save_npy(out + "losses.npy", losses);
        save_npy(out + "ba_grad.npy", torch::tensor(mo.last_ba_grad).reshape({-1, 7}));
        save_npy(out + "kept.npy", torch::tensor({mo.last_kept_rays}));
        for (auto k : {"grid_middle", "grid_fine", "grid_color"}) save_npy(out + k + ".npy", c.at(k));
        save_npy(out + "dec_color.npy", dec.color_decoder->packed());
        save_npy(out + "poses.npy", torch::stack(est));
        std::vector<torch::Tensor> kfp;
        for (int k = 0; k < mo.n_keyframes(); ++k) kfp.push_back(mo.keyframe_est_c2w(k));
        save_npy(out + "kf_poses.npy", torch::stack(kfp));
        std::printf("dist_test rank %d of %d ok: %.1f us per iteration in the last call\n", rank, world, mo.last_iter_us);
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "dist_test rank %d failed: %s\n", rank, e.what());
        return 1;
    }
}
