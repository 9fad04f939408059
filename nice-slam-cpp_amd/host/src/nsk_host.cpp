// nsk_host.cpp -- C++ host classes with the reference's surface (Renderer / NICE / MLP / Tracker / Mapper, utils.h
// helpers) marshalling into the C-ABI of include/nsk.h.  libtorch is used as a host-side tensor container and for
// host-side bookkeeping (pixel sampling, the 7-parameter pose optimiser); no libtorch GPU compute op is called.
// Each function cites the reference code whose behaviour it reproduces (intended semantics).
#include <hip/hip_runtime_api.h>

#include <dlfcn.h>

#include <chrono>
#include <thread>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "Mapper.h"
#include "Tracker.h"
#include "nsk_host.h"
#include "torchlib/utils.h"

// ---------------------------------------------------------------------------------------------------------
// glue
// ---------------------------------------------------------------------------------------------------------
namespace nskh {

static nsk_ctx* g_ctx = nullptr;
static const char* KEYS[4] = {"grid_coarse", "grid_middle", "grid_fine", "grid_color"};
static const void* g_grid_ptr[4] = {nullptr, nullptr, nullptr, nullptr};
static int64_t g_grid_ver[4] = {-1, -1, -1, -1};
static std::vector<int64_t> g_grid_shape[4];

void check(int rc)
{
    if (rc != 0) throw std::runtime_error(std::string("nsk: ") + nsk_last_error());
}

nsk_ctx* ctx()
{
    if (!g_ctx) {
        const char* d = std::getenv("NSK_DEVICE");
        check(nsk_ctx_create(d ? std::atoi(d) : 0, nullptr, &g_ctx));
    }
    return g_ctx;
}

int stage_id(const std::string& s)
{
    if (s == "coarse") return NSK_COARSE;
    if (s == "middle") return NSK_MIDDLE;
    if (s == "fine") return NSK_FINE;
    if (s == "color") return NSK_COLOR;
    throw std::runtime_error("unknown stage '" + s + "'");
}

// ---- N > 1 ---------------------------------------------------------------------------------------------
void Dist::shard(int n, int* lo, int* hi) const
{
    const int base = n / world, rem = n % world;
    *lo = rank * base + std::min(rank, rem);
    *hi = *lo + base + (rank < rem ? 1 : 0);
}
static void* rccl_sym(const char* name)
{
    static void* h = nullptr;
    if (!h) { h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL); if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL); }
    if (!h) throw std::runtime_error(std::string("cannot load librccl.so: ") + dlerror());
    void* f = dlsym(h, name);
    if (!f) throw std::runtime_error(std::string("librccl: symbol not found: ") + name);
    return f;
}
void Dist::allreduce(float* d_buf, size_t n) const
{
    if (!on() || n == 0) return;
    if (hook) { hook(d_buf, n, user); return; }
    if (!comm) throw std::runtime_error("nskh::Dist: world > 1 needs an RCCL communicator or an exchange hook");
    typedef int (*allreduce_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
    static allreduce_t fn = (allreduce_t)rccl_sym("ncclAllReduce");
    if (fn(d_buf, d_buf, n, 7 /* ncclFloat32 */, 0 /* ncclSum */, comm, (hipStream_t)nsk_stream(ctx())) != 0) throw std::runtime_error("ncclAllReduce failed");
}
void Dist::allreduce_grads() const
{
    if (!on()) return;
    if (comm && !hook) { check(nsk_allreduce_grads(ctx(), comm)); return; }
    float* buf = nullptr; size_t n = 0;
    check(nsk_grad_pack(ctx(), &buf, &n));
    allreduce(buf, n);
    check(nsk_grad_unpack(ctx()));
}
void Dist::broadcast0(float* d_buf, size_t n) const
{
    if (!on()) return;
    if (rank != 0 && hipMemsetAsync(d_buf, 0, n * sizeof(float), (hipStream_t)nsk_stream(ctx())) != hipSuccess) throw std::runtime_error("hipMemsetAsync failed");
    allreduce(d_buf, n);
}
void* rccl_comm_from_file(int rank, int world, const std::string& id_file)
{
    struct UniqueId { char internal[128]; } id;
    typedef int (*getid_t)(UniqueId*);
    typedef int (*init_t)(void**, int, UniqueId, int);
    std::memset(&id, 0, sizeof(id));
    if (rank == 0) {
        if (((getid_t)rccl_sym("ncclGetUniqueId"))(&id) != 0) throw std::runtime_error("ncclGetUniqueId failed");
        const std::string tmp = id_file + ".tmp";
        FILE* f = std::fopen(tmp.c_str(), "wb");
        if (!f || std::fwrite(&id, 1, sizeof(id), f) != sizeof(id)) throw std::runtime_error("cannot write " + tmp);
        std::fclose(f);
        if (std::rename(tmp.c_str(), id_file.c_str()) != 0) throw std::runtime_error("cannot publish " + id_file);     // atomic: readers never see a partial id
    } else {
        for (int tries = 0; ; ++tries) {
            FILE* f = std::fopen(id_file.c_str(), "rb");
            if (f) { const size_t got = std::fread(&id, 1, sizeof(id), f); std::fclose(f); if (got == sizeof(id)) break; }
            if (tries > 6000) throw std::runtime_error("timed out waiting for " + id_file);
            std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
    }
    ctx();                                              // the communicator lives on this process's device (NSK_DEVICE)
    void* comm = nullptr;
    if (((init_t)rccl_sym("ncclCommInitRank"))(&comm, world, id, rank) != 0 || !comm) throw std::runtime_error("ncclCommInitRank failed");
    return comm;
}

DevBuf::~DevBuf() { if (p) hipFree(p); }
void DevBuf::ensure(size_t count)
{
    if (count <= n) return;
    if (p) hipFree(p);
    if (hipMalloc((void**)&p, count * sizeof(float)) != hipSuccess) throw std::runtime_error("hipMalloc failed");
    n = count;
}
void DevBuf::upload(const torch::Tensor& t)
{
    torch::Tensor h = t.detach().to(torch::kCPU, torch::kFloat32).contiguous();
    ensure((size_t)h.numel());
    if (hipMemcpy(p, h.data_ptr<float>(), h.numel() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
        throw std::runtime_error("hipMemcpy H2D failed");
}
torch::Tensor DevBuf::download(at::IntArrayRef shape) const
{
    torch::Tensor h = torch::empty(shape, torch::kFloat32);
    check(nsk_sync(ctx()));
    if (hipMemcpy(h.data_ptr<float>(), p, h.numel() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
        throw std::runtime_error("hipMemcpy D2H failed");
    return h;
}

void sync_grids(const GridDict& c)
{
    for (int l = 0; l < 4; ++l) {
        if (!c.contains(KEYS[l])) continue;
        torch::Tensor t = c.at(KEYS[l]);
        int64_t ver = (int64_t)t._version();
        if (t.data_ptr() == g_grid_ptr[l] && ver == g_grid_ver[l]) continue;
        TORCH_CHECK(t.dim() == 5 && t.size(0) == 1, KEYS[l], " must be [1,C,Z,Y,X]");
        torch::Tensor h = t.detach().to(torch::kCPU, torch::kFloat32).contiguous();
        check(nsk_grid_upload(ctx(), l, h.data_ptr<float>(), (int)h.size(1), (int)h.size(2), (int)h.size(3), (int)h.size(4)));
        g_grid_ptr[l] = t.data_ptr(); g_grid_ver[l] = ver; g_grid_shape[l] = t.sizes().vec();
    }
}

void fetch_grids(GridDict& c)
{
    for (int l = 0; l < 4; ++l) {
        if (!c.contains(KEYS[l]) || g_grid_shape[l].empty()) continue;
        torch::Tensor t = c.at(KEYS[l]);
        torch::Tensor h = torch::empty(g_grid_shape[l], torch::kFloat32);
        check(nsk_grid_download(ctx(), l, h.data_ptr<float>()));
        {
            torch::NoGradGuard ng;
            t.copy_(h);                              // a memcpy (H2D if the Dict lives on the GPU), not a compute op
        }
        g_grid_ptr[l] = t.data_ptr(); g_grid_ver[l] = (int64_t)t._version();
    }
}

static void set_bound_ctx(const torch::Tensor& b)
{
    torch::Tensor h = b.detach().to(torch::kCPU, torch::kFloat32).contiguous();
    TORCH_CHECK(h.numel() == 6, "bound must be [3,2]");
    check(nsk_set_bound(ctx(), h.data_ptr<float>()));
}

}  // namespace nskh

using nskh::check;
using nskh::ctx;
using nskh::DevBuf;

// ---------------------------------------------------------------------------------------------------------
// models
// ---------------------------------------------------------------------------------------------------------
GaussianFFT::GaussianFFT(int num_channels, int mapping_size, int scale)
{
    B = register_parameter("GFF", torch::randn({num_channels, mapping_size}) * scale);      // GaussianFFT.cpp:6
}
GaussianFFT::GaussianFFT() {}
torch::Tensor GaussianFFT::forward(torch::Tensor x)                                          // GaussianFFT.cpp:10-15
{
    x = x.squeeze(0);
    return torch::sin(torch::matmul(x, B));
}

static torch::nn::Linear dense(int in, int out, bool relu_gain)                               // MLP.cpp:65-74 (D17: linear gain 1)
{
    torch::nn::Linear l(in, out);
    torch::NoGradGuard ng;
    torch::nn::init::xavier_uniform_(l->weight, relu_gain ? std::sqrt(2.0) : 1.0);
    torch::nn::init::zeros_(l->bias);
    return l;
}

MLP::MLP(std::string name_, int dim, int c_dim_, int hidden_size, int n_blocks_, bool color_, std::vector<int> skips_, float grid_len_,
         std::string /*pose_emb*/, bool concat_feat_)
    : name(name_), color(color_), concat_feat(concat_feat_), c_dim(c_dim_), n_blocks(n_blocks_), skips(skips_), grid_len(grid_len_)
{
    embedding_size = 93;                                                                      // MLP.cpp:21
    embedder = register_module("embedder", std::make_shared<GaussianFFT>(dim, embedding_size, 25));
    fc = register_module("fc", torch::nn::ModuleList());
    pts_linear = register_module("pts_linear", torch::nn::ModuleList());
    for (int i = 0; i < 5; ++i) fc->push_back(torch::nn::Linear(c_dim, hidden_size));         // MLP.cpp:14-20
    const int in_dims[5] = {embedding_size, hidden_size, hidden_size, hidden_size + embedding_size, hidden_size};   // :24-41
    for (int i = 0; i < 5; ++i) pts_linear->push_back(dense(in_dims[i], hidden_size, true));
    output_linear = register_module("linear", dense(hidden_size, color ? 4 : 1, false));      // :43-46
}

static void append_flat(std::vector<torch::Tensor>& v, const torch::Tensor& t) { v.push_back(t.detach().to(torch::kCPU, torch::kFloat32).reshape({-1})); }

torch::Tensor MLP::packed()
{
    std::vector<torch::Tensor> v;
    append_flat(v, embedder->B);
    for (int i = 0; i < 5; ++i) { auto* l = pts_linear[i]->as<torch::nn::Linear>(); append_flat(v, l->weight); append_flat(v, l->bias); }
    for (int i = 0; i < 5; ++i) { auto* l = fc[i]->as<torch::nn::Linear>(); append_flat(v, l->weight); append_flat(v, l->bias); }
    append_flat(v, output_linear->weight); append_flat(v, output_linear->bias);
    return torch::cat(v).contiguous();
}

static int64_t take(torch::Tensor dst, const torch::Tensor& flat, int64_t o)
{
    torch::NoGradGuard ng;
    int64_t n = dst.numel();
    dst.copy_(flat.slice(0, o, o + n).reshape(dst.sizes()));
    return o + n;
}

void MLP::unpack(const torch::Tensor& flat)
{
    int64_t o = 0;
    o = take(embedder->B, flat, o);
    for (int i = 0; i < 5; ++i) { auto* l = pts_linear[i]->as<torch::nn::Linear>(); o = take(l->weight, flat, o); o = take(l->bias, flat, o); }
    for (int i = 0; i < 5; ++i) { auto* l = fc[i]->as<torch::nn::Linear>(); o = take(l->weight, flat, o); o = take(l->bias, flat, o); }
    o = take(output_linear->weight, flat, o); o = take(output_linear->bias, flat, o);
    TORCH_CHECK(o == flat.numel(), "MLP::unpack: size mismatch");
}

MLP_no_xyz::MLP_no_xyz(std::string name_, int /*dim*/, int c_dim_, int hidden_size, int n_blocks_, bool color_, std::vector<int> skips_,
                       float grid_len_)
    : name(name_), color(color_), c_dim(c_dim_), n_blocks(n_blocks_), skips(skips_), grid_len(grid_len_)
{
    pts_linear = register_module("pts_linear", torch::nn::ModuleList());
    const int in_dims[5] = {hidden_size, hidden_size, hidden_size, hidden_size + c_dim, hidden_size};               // MLP.cpp:114-131
    for (int i = 0; i < 5; ++i) pts_linear->push_back(dense(in_dims[i], hidden_size, true));
    output_linear = register_module("linear", dense(hidden_size, color ? 4 : 1, false));
}

torch::Tensor MLP_no_xyz::packed()
{
    std::vector<torch::Tensor> v;
    for (int i = 0; i < 5; ++i) { auto* l = pts_linear[i]->as<torch::nn::Linear>(); append_flat(v, l->weight); append_flat(v, l->bias); }
    append_flat(v, output_linear->weight); append_flat(v, output_linear->bias);
    return torch::cat(v).contiguous();
}

void MLP_no_xyz::unpack(const torch::Tensor& flat)
{
    int64_t o = 0;
    for (int i = 0; i < 5; ++i) { auto* l = pts_linear[i]->as<torch::nn::Linear>(); o = take(l->weight, flat, o); o = take(l->bias, flat, o); }
    o = take(output_linear->weight, flat, o); o = take(output_linear->bias, flat, o);
    TORCH_CHECK(o == flat.numel(), "MLP_no_xyz::unpack: size mismatch");
}

// the per-decoder forwards exist for API compatibility; the kernels evaluate whole stages (NICE::forward)
static torch::Tensor decoder_alone(int which, torch::Tensor p, std::map<std::string, torch::Tensor> c_grid, const torch::Tensor& packed)
{
    nskh::GridDict d;
    for (auto& kv : c_grid) d.insert(kv.first, kv.second);
    nskh::sync_grids(d);
    check(nsk_decoder_upload(ctx(), which, packed.data_ptr<float>(), (size_t)packed.numel()));
    TORCH_CHECK(which == NSK_COARSE || which == NSK_MIDDLE, "stand-alone forward is defined for the coarse and middle decoders "
                "(fine and color are only meaningful inside a stage: NICE::forward)");
    torch::Tensor pts = p.reshape({-1, 3});
    DevBuf dp, dr;
    dp.upload(pts); dr.ensure((size_t)pts.size(0) * 4);
    check(nsk_eval_points(ctx(), which, (int)pts.size(0), dp.p, dr.p));
    return dr.download({pts.size(0), 4}).index({Slice(), 3});
}
torch::Tensor MLP::forward(torch::Tensor p, std::map<std::string, torch::Tensor> c_grid)
{
    return decoder_alone(name == "middle" ? NSK_MIDDLE : (name == "fine" ? NSK_FINE : NSK_COLOR), p, c_grid, packed());
}
torch::Tensor MLP_no_xyz::forward(torch::Tensor p, std::map<std::string, torch::Tensor> c_grid) { return decoder_alone(NSK_COARSE, p, c_grid, packed()); }

NICE::NICE(int dim, int c_dim, int hidden_size, float coarse_grid_len, float middle_grid_len, float fine_grid_len, float color_grid_len,
           bool /*coarse*/, std::string pose_emb)                                             // NICE.cpp:3-14
{
    coarse_decoder = register_module("coarse_decoder", std::make_shared<MLP_no_xyz>("coarse", dim, c_dim, hidden_size, 5, false, std::vector<int>({2}), coarse_grid_len));
    middle_decoder = register_module("middle_decoder", std::make_shared<MLP>("middle", dim, c_dim, hidden_size, 5, false, std::vector<int>({2}), middle_grid_len, pose_emb, false));
    fine_decoder = register_module("fine_decoder", std::make_shared<MLP>("fine", dim, c_dim * 2, hidden_size, 5, false, std::vector<int>({2}), fine_grid_len, pose_emb, true));
    color_decoder = register_module("color_decoder", std::make_shared<MLP>("color", dim, c_dim, hidden_size, 5, true, std::vector<int>({2}), color_grid_len, pose_emb, false));
}

// NICE is passed by value through the reference's signatures; the upload cache therefore lives outside the object
static int64_t uploaded_version_[4] = {-1, -1, -1, -1};
void NICE::mark_dirty() { for (auto& v : uploaded_version_) v = -1; }

int64_t NICE::param_version(int which)
{
    int64_t v = 0;
    torch::nn::Module* m = which == 0 ? (torch::nn::Module*)coarse_decoder.get() : which == 1 ? (torch::nn::Module*)middle_decoder.get()
                          : which == 2 ? (torch::nn::Module*)fine_decoder.get() : (torch::nn::Module*)color_decoder.get();
    for (auto& p : m->parameters()) v = v * 31 + (int64_t)p._version() + (int64_t)(reinterpret_cast<uintptr_t>(p.data_ptr()) >> 4);
    return v;
}

void NICE::sync_to_device()
{
    for (int w = 0; w < 4; ++w) {
        int64_t v = param_version(w);
        if (v == uploaded_version_[w]) continue;
        torch::Tensor flat = w == 0 ? coarse_decoder->packed() : w == 1 ? middle_decoder->packed() : w == 2 ? fine_decoder->packed() : color_decoder->packed();
        check(nsk_decoder_upload(ctx(), w, flat.data_ptr<float>(), (size_t)flat.numel()));
        uploaded_version_[w] = v;
    }
}

void NICE::fetch_from_device(bool fine, bool color)
{
    for (int w = 2; w < 4; ++w) {
        if ((w == 2 && !fine) || (w == 3 && !color)) continue;
        size_t n = nsk_decoder_param_count(w);
        torch::Tensor flat = torch::empty({(int64_t)n}, torch::kFloat32);
        check(nsk_decoder_download(ctx(), w, flat.data_ptr<float>(), n));
        if (w == 2) fine_decoder->unpack(flat); else color_decoder->unpack(flat);
        uploaded_version_[w] = param_version(w);
    }
}

torch::Tensor NICE::forward(torch::Tensor p, c10::Dict<std::string, torch::Tensor> c_grid, std::string stage)      // NICE.cpp:16-52
{
    nskh::sync_grids(c_grid);
    sync_to_device();
    torch::Tensor pts = p.reshape({-1, 3});
    DevBuf dp, dr;
    dp.upload(pts); dr.ensure((size_t)pts.size(0) * 4);
    check(nsk_eval_points(ctx(), nskh::stage_id(stage), (int)pts.size(0), dp.p, dr.p));
    return dr.download({pts.size(0), 4}).to(p.device());
}

// ---------------------------------------------------------------------------------------------------------
// Renderer
// ---------------------------------------------------------------------------------------------------------
Renderer::Renderer()                                                                          // Renderer.cpp:3-17
{
    ray_batch_size = 500000; points_batch_size = 100000;
    lindisp = false; perturb = 0;
    N_samples = 32; N_surface = 16; N_importance = 0;
    scale = 1;
    occupancy = false;        // the reference's member says true but it passes the literal false to raw2outputs (:125, D9)
    bound = torch::tensor({{-4.5, 3.82}, {-1.5, 2.02}, {-3.0, 2.76}});
}

void Renderer::set_bound(torch::Tensor b) { bound = b.detach().to(torch::kCPU, torch::kFloat32).clone(); }

void Renderer::push_opts()
{
    nskh::set_bound_ctx(bound);
    check(nsk_set_render_opts(ctx(), N_samples, N_surface, lindisp ? 1 : 0, perturb, occupancy ? 1 : 0, 0));
}

torch::Tensor Renderer::eval_points(torch::Tensor p, NICE decoders, c10::Dict<std::string, torch::Tensor> c, std::string stage)    // :19-42
{
    push_opts();
    return decoders.forward(p, c, stage);           // chunking by points_batch_size is unnecessary: the kernels stream tiles
}

void Renderer::render_batch_ray(c10::Dict<std::string, torch::Tensor> c, NICE decoders, torch::Tensor rays_d, torch::Tensor rays_o,
                                std::string stage, torch::Tensor gt_depth, torch::Tensor& rgb_map, torch::Tensor& depth_map,
                                torch::Tensor& depth_var, torch::Tensor& weights)                // :44-126
{
    push_opts();
    nskh::sync_grids(c);
    decoders.sync_to_device();
    const int N = (int)rays_o.size(0);
    const bool has_gt = gt_depth.defined();
    const int S = N_samples + (has_gt ? N_surface : 0);                                       // D8: N_surface is not mutated
    DevBuf ro, rd, gd, o_rgb, o_d, o_v, o_w;
    ro.upload(rays_o.reshape({-1, 3})); rd.upload(rays_d.reshape({-1, 3}));
    if (has_gt) gd.upload(gt_depth.reshape({-1}));
    o_rgb.ensure((size_t)N * 3); o_d.ensure(N); o_v.ensure(N); o_w.ensure((size_t)N * S);
    check(nsk_render_forward(ctx(), nskh::stage_id(stage), N, ro.p, rd.p, has_gt ? gd.p : nullptr, -1.f, o_rgb.p, o_d.p, o_v.p, o_w.p));
    auto dev = rays_o.device();
    rgb_map = o_rgb.download({N, 3}).to(dev); depth_map = o_d.download({N}).to(dev);
    depth_var = o_v.download({N}).to(dev); weights = o_w.download({N, S}).to(dev);
}

// ---------------------------------------------------------------------------------------------------------
// torchlib/utils.h
// ---------------------------------------------------------------------------------------------------------
void raySampler(int H0, int H1, int W0, int W1, int n, float fx, float fy, float cx, float cy, torch::Tensor depth, torch::Tensor color,
                torch::Tensor c2w, torch::Tensor& rays_o, torch::Tensor& rays_d, torch::Tensor& gt_color, torch::Tensor& gt_depth,
                torch::Tensor* pix_i, torch::Tensor* pix_j)                                    // utils.h:13-55
{
    depth = depth.to(torch::kCPU).index({Slice(H0, H1), Slice(W0, W1)});
    color = color.to(torch::kCPU).index({Slice(H0, H1), Slice(W0, W1)});
    const int64_t Wc = W1 - W0, Hc = H1 - H0;
    torch::Tensor ind = torch::randint(Wc * Hc, {n}, torch::kLong);                          // :32
    torch::Tensor i = (ind % Wc + W0).to(torch::kFloat32);                                    // column
    torch::Tensor j = (torch::div(ind, Wc, "floor") + H0).to(torch::kFloat32);                // row
    gt_color = color.reshape({-1, 3}).index({ind});
    gt_depth = depth.reshape({-1}).index({ind});
    torch::Tensor dirs = torch::stack({(i - cx) / fx, -(j - cy) / fy, -torch::ones_like(i)}, -1).reshape({-1, 1, 3});   // D11 intended
    torch::Tensor m = c2w.detach().to(torch::kCPU, torch::kFloat32);
    rays_d = torch::sum(dirs * m.index({Slice(None, 3), Slice(None, 3)}), -1);               // :51
    rays_o = m.index({Slice(None, 3), -1}).expand(rays_d.sizes()).contiguous();              // :52
    if (pix_i) *pix_i = i.to(torch::kInt32);
    if (pix_j) *pix_j = j.to(torch::kInt32);
}

void get_samples(int H0, int H1, int W0, int W1, int n, int /*H*/, int /*W*/, float fx, float fy, float cx, float cy, torch::Tensor c2w,
                 torch::Tensor depth, torch::Tensor color, torch::Tensor& rays_o, torch::Tensor& rays_d, torch::Tensor& sample_depth,
                 torch::Tensor& sample_color, torch::Tensor* pix_i, torch::Tensor* pix_j)     // utils.h:141-146
{
    raySampler(H0, H1, W0, W1, n, fx, fy, cx, cy, depth, color, c2w, rays_o, rays_d, sample_color, sample_depth, pix_i, pix_j);
}

void raw2outputs_nerf_color(torch::Tensor raw, torch::Tensor z_vals, bool occupancy, torch::Tensor rays_d, torch::Tensor& rgb_map,
                            torch::Tensor& depth_map, torch::Tensor& depth_var, torch::Tensor& weights)      // utils.h:148-172
{
    const int N = (int)z_vals.size(0), S = (int)z_vals.size(1);
    DevBuf r, z, d, o_rgb, o_d, o_v, o_w;
    r.upload(raw.reshape({-1, 4})); z.upload(z_vals); d.upload(rays_d.reshape({-1, 3}));
    o_rgb.ensure((size_t)N * 3); o_d.ensure(N); o_v.ensure(N); o_w.ensure((size_t)N * S);
    check(nsk_raw2outputs(ctx(), N, S, r.p, z.p, d.p, occupancy ? 1 : 0, o_rgb.p, o_d.p, o_v.p, o_w.p));
    auto dev = raw.device();
    rgb_map = o_rgb.download({N, 3}).to(dev); depth_map = o_d.download({N}).to(dev);
    depth_var = o_v.download({N}).to(dev); weights = o_w.download({N, S}).to(dev);
}

torch::Tensor quad2rotation(torch::Tensor quad)                                               // utils.h:174-195
{
    auto qr = quad.index({Slice(), 0}), qi = quad.index({Slice(), 1}), qj = quad.index({Slice(), 2}), qk = quad.index({Slice(), 3});
    auto two_s = 2 / (quad * quad).sum(-1);
    auto r0 = torch::stack({1 - two_s * (qj.pow(2) + qk.pow(2)), two_s * (qi * qj - qk * qr), two_s * (qi * qk + qj * qr)}, -1);
    auto r1 = torch::stack({two_s * (qi * qj + qk * qr), 1 - two_s * (qi.pow(2) + qk.pow(2)), two_s * (qj * qk - qi * qr)}, -1);
    auto r2 = torch::stack({two_s * (qi * qk - qj * qr), two_s * (qj * qk + qi * qr), 1 - two_s * (qi.pow(2) + qj.pow(2))}, -1);
    return torch::stack({r0, r1, r2}, 1);
}

torch::Tensor get_camera_from_tensor(torch::Tensor inputs)                                    // utils.h:198-210
{
    const bool one = inputs.dim() == 1;
    if (one) inputs = inputs.unsqueeze(0);
    auto R = quad2rotation(inputs.index({Slice(), Slice(None, 4)}));
    auto T = inputs.index({Slice(), Slice(4, None)});
    auto RT = torch::cat({R, T.index({Slice(), Slice(), None})}, 2);
    return one ? RT[0] : RT;
}

torch::Tensor get_tensor_from_camera(torch::Tensor RT, bool Tquad)                            // utils.h:212-231 (D23)
{
    torch::Tensor m = RT.detach().to(torch::kCPU, torch::kFloat64);
    auto a = m.accessor<double, 2>();
    double R[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[i][j] = a[i][j];
    double q[4];                                                                              // (w,x,y,z), Shepperd's method
    double tr = R[0][0] + R[1][1] + R[2][2];
    if (tr > 0) { double s = std::sqrt(tr + 1.0) * 2; q[0] = 0.25 * s; q[1] = (R[2][1] - R[1][2]) / s; q[2] = (R[0][2] - R[2][0]) / s; q[3] = (R[1][0] - R[0][1]) / s; }
    else if (R[0][0] > R[1][1] && R[0][0] > R[2][2]) { double s = std::sqrt(1.0 + R[0][0] - R[1][1] - R[2][2]) * 2; q[0] = (R[2][1] - R[1][2]) / s; q[1] = 0.25 * s; q[2] = (R[0][1] + R[1][0]) / s; q[3] = (R[0][2] + R[2][0]) / s; }
    else if (R[1][1] > R[2][2]) { double s = std::sqrt(1.0 + R[1][1] - R[0][0] - R[2][2]) * 2; q[0] = (R[0][2] - R[2][0]) / s; q[1] = (R[0][1] + R[1][0]) / s; q[2] = 0.25 * s; q[3] = (R[1][2] + R[2][1]) / s; }
    else { double s = std::sqrt(1.0 + R[2][2] - R[0][0] - R[1][1]) * 2; q[0] = (R[1][0] - R[0][1]) / s; q[1] = (R[0][2] + R[2][0]) / s; q[2] = (R[1][2] + R[2][1]) / s; q[3] = 0.25 * s; }
    torch::Tensor quad = torch::tensor({(float)q[0], (float)q[1], (float)q[2], (float)q[3]});
    torch::Tensor T = torch::tensor({(float)a[0][3], (float)a[1][3], (float)a[2][3]});
    return Tquad ? torch::cat({T, quad}, 0) : torch::cat({quad, T}, 0);
}

// t = min_axis max_side (bound - o)/d >= gt_depth (src/Mapper.cpp:416-427, src/Tracker.cpp:48-58), on the host
static torch::Tensor inside_mask(const torch::Tensor& bound, const torch::Tensor& ro, const torch::Tensor& rd, const torch::Tensor& gd)
{
    torch::NoGradGuard ng;
    auto t_ = (bound.unsqueeze(0) - ro.unsqueeze(-1)) / rd.unsqueeze(-1);
    auto t = std::get<0>(torch::min(std::get<0>(torch::max(t_, 2)), 1));
    return t >= gd;
}

// ---------------------------------------------------------------------------------------------------------
// device-resident state shared by Tracker::run and Mapper::optimize_map
// ---------------------------------------------------------------------------------------------------------
#define HIPOK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw std::runtime_error(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

namespace {

// grow-only device array; uploads and downloads go through the kernels' stream, so they are ordered with the launches
template <typename T>
struct DevArr {
    T* p = nullptr; size_t cap = 0;
    ~DevArr() { if (p) hipFree(p); }
    DevArr() {}
    DevArr(const DevArr&) = delete;
    DevArr& operator=(const DevArr&) = delete;
    void ensure(size_t n)
    {
        if (n <= cap) return;
        check(nsk_sync(ctx()));                          // nobody may still be reading the old buffer
        if (p) HIPOK(hipFree(p));
        HIPOK(hipMalloc((void**)&p, n * sizeof(T)));
        cap = n;
    }
    void upload(const T* h, size_t n)
    {
        ensure(n);
        HIPOK(hipMemcpyAsync(p, h, n * sizeof(T), hipMemcpyHostToDevice, (hipStream_t)nsk_stream(ctx())));
        check(nsk_sync(ctx()));                          // h may be a temporary
    }
    void zero(size_t n) { ensure(n); HIPOK(hipMemsetAsync(p, 0, n * sizeof(T), (hipStream_t)nsk_stream(ctx()))); }
    void download(T* h, size_t n) const
    {
        HIPOK(hipMemcpyAsync(h, p, n * sizeof(T), hipMemcpyDeviceToHost, (hipStream_t)nsk_stream(ctx())));
        check(nsk_sync(ctx()));
    }
};

// one frame's images on the device, keyed by the host tensors they came from (a keyframe is uploaded once, not once per iteration)
struct DevFrame {
    DevArr<float> depth, color;
    const void* key_d = nullptr; const void* key_c = nullptr; int64_t ver_d = -1, ver_c = -1;
    // The key tensors are HELD: a (data_ptr, version) pair identifies an image only while its storage is alive.  The callers build a fresh
    // from_blob(...).clone() per frame, the CPU allocator hands the freed address out again and a fresh clone starts at version 0, so
    // an unheld key let frame k+1 pass for frame k (no upload, the previous frame optimised again, no error).
    torch::Tensor hold_d, hold_c;
    int H = 0, W = 0;
    void set(const torch::Tensor& depth_t, const torch::Tensor& color_t)
    {
        if (hold_d.defined() && hold_c.defined() && depth_t.data_ptr() == key_d && (int64_t)depth_t._version() == ver_d &&
            color_t.data_ptr() == key_c && (int64_t)color_t._version() == ver_c) return;
        torch::Tensor d = depth_t.detach().to(torch::kCPU, torch::kFloat32).contiguous();
        torch::Tensor c = color_t.detach().to(torch::kCPU, torch::kFloat32).contiguous();
        TORCH_CHECK(d.dim() == 2 && c.dim() == 3 && c.size(2) == 3 && c.size(0) == d.size(0) && c.size(1) == d.size(1), "frame images must be depth [H,W] and colour [H,W,3]");
        H = (int)d.size(0); W = (int)d.size(1);
        depth.upload(d.data_ptr<float>(), (size_t)d.numel());
        color.upload(c.data_ptr<float>(), (size_t)c.numel());
        key_d = depth_t.data_ptr(); ver_d = (int64_t)depth_t._version(); key_c = color_t.data_ptr(); ver_c = (int64_t)color_t._version();
        hold_d = depth_t; hold_c = color_t;              // (handles: no copy; they pin the storage, hence the address, until the next set)
    }
};

struct RayBufs {                 // one batch of rays on the device
    DevArr<int32_t> pi, pj;
    DevArr<float> ro, rd, gd, gc, g_ro, g_rd;
    DevArr<uint8_t> keep;
    void ensure(size_t n)
    {
        pi.ensure(n); pj.ensure(n); ro.ensure(3 * n); rd.ensure(3 * n); gd.ensure(n); gc.ensure(3 * n); g_ro.ensure(3 * n); g_rd.ensure(3 * n); keep.ensure(n);
    }
};

double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
// Tracker
// ---------------------------------------------------------------------------------------------------------
struct Tracker::Dev {
    DevFrame frame;
    RayBufs rays;
    DevArr<float> cam, m, v, losses;     // pose 7-vector, its Adam moments, one loss per iteration
};

Tracker::Tracker(YAML::Node ns_config, YAML::Node cf_config, c10::Dict<std::string, torch::Tensor> c_dict) : renderer()      // Tracker.cpp:5-34
{
    handle_dynamic = ns_config["tracking"]["handle_dynamic"].as<bool>();
    use_color_in_tracking = ns_config["tracking"]["use_color_in_tracking"].as<bool>();
    w_color_loss = ns_config["tracking"]["w_color_loss"].as<float>();
    lr = ns_config["tracking"]["lr"].as<float>();
    num_cam_iters = ns_config["tracking"]["iters"].as<int>();
    tracking_pixels = ns_config["tracking"]["pixels"].as<int>();
    ignore_edge_w = ns_config["tracking"]["ignore_edge_W"].as<int>();
    ignore_edge_h = ns_config["tracking"]["ignore_edge_H"].as<int>();
    bound = torch::tensor({{-4.5, 3.82}, {-1.5, 2.02}, {-3.0, 2.76}});
    H = cf_config["cam"]["H"].as<int>(); W = cf_config["cam"]["W"].as<int>();
    fx = cf_config["cam"]["fx"].as<float>(); fy = cf_config["cam"]["fy"].as<float>();
    cx = cf_config["cam"]["cx"].as<float>(); cy = cf_config["cam"]["cy"].as<float>();
    idx = 0;
    c = c_dict;
    dev = std::make_shared<Dev>();
}
Tracker::~Tracker() {}
void Tracker::update_para_from_mapping() {}
void Tracker::set_bound(torch::Tensor b) { bound = b.detach().to(torch::kCPU, torch::kFloat32).clone(); renderer.set_bound(bound); }

// One iteration through the public signature (the reference's: cam tensor and torch optimiser on the host).  The batch lives on the
// device (pixel draw, gather, rays, bound filter, render, loss, backward, d loss / d pose); only the 7 pose gradients and the loss come
// back, because the caller's torch::optim::Adam owns the step.  Tracker::run keeps the pose and its Adam state on the device as well.
torch::Tensor Tracker::optimize_cam_in_batch(torch::Tensor cam_tensor, torch::Tensor gt_color, torch::Tensor gt_depth, int batch_size,
                                             torch::optim::Adam& optimizer, NICE decoders)      // Tracker.cpp:41-89
{
    optimizer.zero_grad();
    Dev& D = *dev;
    D.frame.set(gt_depth, gt_color);
    torch::Tensor cam_cpu = cam_tensor.detach().to(torch::kCPU, torch::kFloat32).contiguous();
    D.cam.upload(cam_cpu.data_ptr<float>(), 7);
    const int N = batch_size;
    D.rays.ensure((size_t)N); D.losses.ensure(8); D.m.ensure(16);      // losses: [loss, d loss / d pose (7)]
    nskh::set_bound_ctx(bound);
    check(nsk_set_render_opts(ctx(), renderer.N_samples, renderer.N_surface, renderer.lindisp, renderer.perturb, renderer.occupancy, 0));
    nskh::sync_grids(c);
    decoders.sync_to_device();
    RayBufs& R = D.rays;
    // utils.h:19-52 + :198 + Tracker.cpp:48-58 in one launch: pixel draw, ground-truth gather, pose -> rays, inside filter
    const nsk_frame_rays fr = {D.frame.depth.p, D.frame.color.p, D.cam.p, 1, rng_seed++};
    check(nsk_prepare_rays(ctx(), 1, &fr, N, ignore_edge_h, H - ignore_edge_h, ignore_edge_w, W - ignore_edge_w, D.frame.H, D.frame.W, fx, fy, cx, cy, 0,
                           R.pi.p, R.pj.p, R.gd.p, R.gc.p, R.ro.p, R.rd.p, R.keep.p));
    check(nsk_set_ray_mask(ctx(), R.keep.p));
    check(nsk_track_step(ctx(), NSK_COLOR, N, R.ro.p, R.rd.p, R.gd.p, R.gc.p, -1.f, w_color_loss, use_color_in_tracking ? 1 : 0,
                         handle_dynamic ? 1 : 0, 1, NSK_GRAD_RAYS, D.losses.p, R.g_ro.p, R.g_rd.p));      // stage "color": :61 (D19)
    check(nsk_set_ray_mask(ctx(), nullptr));
    float* g_c2w = D.m.p;              // 12 floats of scratch
    check(nsk_rays_backward(ctx(), N, R.pi.p, R.pj.p, fx, fy, cx, cy, 0, R.g_ro.p, R.g_rd.p, g_c2w));
    check(nsk_camera_backward(ctx(), D.cam.p, g_c2w, D.losses.p + 1));          // losses[1..7] = d loss / d pose
    float h[8];
    D.losses.download(h, 8);
    cam_tensor.mutable_grad() = torch::from_blob(h + 1, {7}, torch::kFloat32).clone().to(cam_tensor.device());       // loss.backward() :84
    optimizer.step();                                                                                              // :85
    optimizer.zero_grad();
    return torch::tensor(h[0]);
}

void Tracker::run(NICE decoders, torch::Tensor gt_color_t, torch::Tensor gt_depth_t, torch::Tensor gt_c2w_t, int idx_)       // Tracker.cpp:92-113
{
    idx = idx_;
    Dev& D = *dev;
    torch::Tensor cam0 = get_tensor_from_camera(gt_c2w_t, false).contiguous();      // :100 (initialised from the GT pose, D25)
    D.frame.set(gt_depth_t, gt_color_t);
    D.cam.upload(cam0.data_ptr<float>(), 7);
    D.m.zero(16); D.v.zero(16);                                                       // torch::optim::Adam re-created per frame (:103)
    const int N = tracking_pixels, iters = num_cam_iters;
    D.rays.ensure((size_t)N); D.losses.ensure((size_t)std::max(iters, 8));
    nskh::set_bound_ctx(bound);
    check(nsk_set_render_opts(ctx(), renderer.N_samples, renderer.N_surface, renderer.lindisp, renderer.perturb, renderer.occupancy, 0));
    nskh::sync_grids(c);
    decoders.sync_to_device();
    RayBufs& R = D.rays;
    check(nsk_sync(ctx()));
    const double t0 = now_us();
    for (int i = 0; i < iters; ++i) {                                                 // :108-112, every operand resident on the device
        const nsk_frame_rays fr = {D.frame.depth.p, D.frame.color.p, D.cam.p, 1, rng_seed++};
        check(nsk_prepare_rays(ctx(), 1, &fr, N, ignore_edge_h, H - ignore_edge_h, ignore_edge_w, W - ignore_edge_w, D.frame.H, D.frame.W, fx, fy, cx, cy, 0,
                               R.pi.p, R.pj.p, R.gd.p, R.gc.p, R.ro.p, R.rd.p, R.keep.p));
        check(nsk_set_ray_mask(ctx(), R.keep.p));
        check(nsk_track_step(ctx(), NSK_COLOR, N, R.ro.p, R.rd.p, R.gd.p, R.gc.p, -1.f, w_color_loss, use_color_in_tracking ? 1 : 0,
                             handle_dynamic ? 1 : 0, 1, NSK_GRAD_RAYS, D.losses.p + i, R.g_ro.p, R.g_rd.p));
        check(nsk_set_ray_mask(ctx(), nullptr));
        check(nsk_pose_step(ctx(), N, R.pi.p, R.pj.p, fx, fy, cx, cy, 0, R.g_ro.p, R.g_rd.p, D.cam.p, D.m.p, D.v.p, lr, 0.9f, 0.999f, 1e-8f, i + 1, nullptr));   // config tracking.lr (D25)
    }
    check(nsk_sync(ctx()));
    last_run_us = now_us() - t0;
    last_losses.assign((size_t)iters, 0.f);
    if (iters > 0) D.losses.download(last_losses.data(), (size_t)iters);              // the only device-to-host traffic of the frame: losses and pose
    for (float l : last_losses) std::cout << "loss: " << l << std::endl;              // :111
    float h[7];
    D.cam.download(h, 7);
    last_camera_tensor = torch::from_blob(h, {7}, torch::kFloat32).clone();
}

void Tracker::run(CoFusionReader& reader, NICE decoders)                              // src/main.cpp:96 (D1)
{
    int n = 0;
    while (reader.hasMore() && (frames_limit < 0 || n < frames_limit)) {
        const int frame = reader.getIdx();
        reader.getNext();
        // cv::Mat -> tensors: depth [H,W] fp32, colour [H,W,3] fp32; the dataset has no poses: c2w = the reader's identity (CoFusionReader.cpp:14)
        torch::Tensor depth_t = torch::from_blob(reader.depth.data, {reader.depth.rows, reader.depth.cols}, torch::kFloat32).clone();
        torch::Tensor color_t = torch::from_blob(reader.rgb.data, {reader.rgb.rows, reader.rgb.cols, reader.rgb.channels()}, torch::kFloat32).clone();
        if (color_t.size(2) > 3) color_t = color_t.index({Slice(), Slice(), Slice(None, 3)}).contiguous();
        torch::Tensor c2w_t = torch::zeros({4, 4});
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) c2w_t[i][j] = reader.c2w(i, j);
        run(decoders, color_t, depth_t, c2w_t, frame);
        ++n;
    }
}

void Tracker::run(SequenceReader& reader, NICE decoders)
{
    int n = 0;
    while (reader.hasMore() && (frames_limit < 0 || n < frames_limit)) {
        const int frame = reader.getIdx();
        reader.getNext();
        torch::Tensor depth_t = torch::from_blob(reader.depth.data, {reader.depth.rows, reader.depth.cols}, torch::kFloat32).clone();
        torch::Tensor color_t = torch::from_blob(reader.rgb.data, {reader.rgb.rows, reader.rgb.cols, reader.rgb.channels()}, torch::kFloat32).clone();
        torch::Tensor c2w_t = torch::zeros({4, 4});
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) c2w_t[i][j] = reader.c2w(i, j);
        run(decoders, color_t, depth_t, c2w_t, frame);
        ++n;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Mapper
// ---------------------------------------------------------------------------------------------------------
struct Mapper::Dev {
    DevFrame cur;                                        // current frame
    std::vector<std::shared_ptr<DevFrame>> kf;           // keyframes, index = position in keyframe_vector
    RayBufs rays, rays2;             // two sets: the next iteration's batch is drawn and registered (nsk_map_prepare) while this one's is optimised
    DevArr<float> poses;                                 // [frames of the window][12] fixed c2w rows (3x4)
    DevArr<float> cams, cam_m, cam_v, cam_g;             // BA: [frames][8] pose 7-vectors, their Adam moments and their last gradients
    DevArr<float> loss;
    DevArr<float> x_ba, x_plain;                         // N > 1: what travels beside the grids -- BA iterations [frames][8] pose gradients | loss | kept rays | 0 x 6; others loss | 0 x 3
    DevArr<float> kf_ro, kf_rd, kf_gd;                   // the 100 overlap-ranking rays
    DevArr<int32_t> kf_pi, kf_pj;
};

Mapper::Mapper(YAML::Node ns_config, YAML::Node cf_config, bool cmapr) : renderer()           // Mapper.cpp:6-36
{
    ns_cfg = ns_config; cf_cfg = cf_config;
    color_refine = ns_cfg["mapping"]["color_refine"].as<bool>();
    mapping_window_size = ns_cfg["mapping"]["mapping_window_size"].as<int>();
    middle_iter_ratio = ns_cfg["mapping"]["middle_iter_ratio"].as<float>();
    fine_iter_ratio = ns_cfg["mapping"]["fine_iter_ratio"].as<float>();
    fix_color = ns_cfg["mapping"]["fix_color"].as<bool>();
    fix_fine = ns_cfg["mapping"]["fix_fine"].as<bool>();
    keyframe_selection_method = ns_cfg["mapping"]["keyframe_selection_method"].as<std::string>();
    frustum_feature_selection = ns_cfg["mapping"]["frustum_feature_selection"].as<bool>();
    keyframe_every = ns_cfg["mapping"]["keyframe_every"].as<int>();
    BA = false;
    coarse_mapper = cmapr;
    H = cf_config["cam"]["H"].as<int>(); W = cf_config["cam"]["W"].as<int>();
    fx = cf_config["cam"]["fx"].as<float>(); fy = cf_config["cam"]["fy"].as<float>();
    cx = cf_config["cam"]["cx"].as<float>(); cy = cf_config["cam"]["cy"].as<float>();
    mapping_pixels = cf_config["mapping"]["pixels"].as<int>();                                // :28 (from cofusion.yaml, D30)
    bound = torch::tensor({{-4.5, 3.82}, {-1.5, 2.02}, {-3.0, 2.76}});
    num_joint_iters = ns_cfg["mapping"]["iters"].as<int>();
    lr_factor = ns_cfg["mapping"]["lr_first_factor"].as<float>();
    BA_cam_lr = ns_cfg["mapping"]["BA_cam_lr"].as<float>();
    w_color_loss = ns_cfg["tracking"]["w_color_loss"].as<float>();                           // :33 as written (D20)
    dev = std::make_shared<Dev>();
}
Mapper::~Mapper() {}
void Mapper::set_bound(torch::Tensor b) { bound = b.detach().to(torch::kCPU, torch::kFloat32).clone(); renderer.set_bound(bound); }

static int level_of_key(const std::string& key)
{
    int level = key == "grid_coarse" ? 0 : key == "grid_middle" ? 1 : key == "grid_fine" ? 2 : key == "grid_color" ? 3 : -1;
    TORCH_CHECK(level >= 0, "unknown grid key ", key);
    return level;
}

void Mapper::set_frustum_mask(const std::string& key, torch::Tensor mask)
{
    const int level = level_of_key(key);
    user_mask[level] = true;
    if (!mask.defined()) { check(nsk_set_mask(ctx(), level, nullptr)); return; }
    torch::Tensor m = mask.to(torch::kCPU, torch::kUInt8).contiguous();
    check(nsk_set_mask(ctx(), level, m.data_ptr<uint8_t>()));
}

void Mapper::get_mask_from_c2w(cv::Mat depth_mat, torch::Tensor c2w, torch::Tensor val_shape, std::string key, torch::Tensor& mask)      // Mapper.cpp:42-130
{
    const int level = level_of_key(key);
    TORCH_CHECK(depth_mat.type() == CV_32FC1, "get_mask_from_c2w: depth must be CV_32FC1");
    torch::Tensor vs = val_shape.to(torch::kCPU, torch::kInt64).contiguous();
    const int64_t Z = vs[0].item<int64_t>(), Y = vs[1].item<int64_t>(), X = vs[2].item<int64_t>();
    nskh::set_bound_ctx(bound);
    torch::Tensor pose = c2w.detach().to(torch::kCPU, torch::kFloat32).contiguous();
    TORCH_CHECK(pose.numel() == 16, "get_mask_from_c2w: c2w must be 4x4");
    DevArr<float> img;
    img.upload(depth_mat.ptr<float>(), depth_mat.total());
    torch::Tensor out = torch::zeros({Z, Y, X}, torch::kUInt8);
    check(nsk_frustum_mask(ctx(), level, img.p, depth_mat.rows, depth_mat.cols, fx, fy, cx, cy, pose.data_ptr<float>(), out.data_ptr<uint8_t>()));
    mask = out.to(torch::kBool).permute({2, 1, 0}).contiguous();          // [X,Y,Z], the layout the reference's caller permutes back (:260)
}

void Mapper::keyframe_selection_overlap(torch::Tensor gt_color_, torch::Tensor gt_depth_, torch::Tensor c2w, std::vector<KeyFrame> keyframe_vector_, int k_overlap,
                                        std::vector<int>& selected_kf)                                                          // Mapper.cpp:132-196
{
    selected_kf.clear();
    last_overlap.clear();
    const int K = (int)keyframe_vector_.size();
    if (K == 0) return;
    Dev& D = *dev;
    const int n = 100, ns = 16;                                                               // :136
    D.cur.set(gt_depth_, gt_color_);
    D.kf_pi.ensure(n); D.kf_pj.ensure(n); D.kf_ro.ensure(3 * n); D.kf_rd.ensure(3 * n); D.kf_gd.ensure(n);
    torch::Tensor pose = c2w.detach().to(torch::kCPU, torch::kFloat32).contiguous();
    D.poses.upload(pose.data_ptr<float>(), 12);
    const uint64_t call_seed = draw_seed(rng_seed, ++draw_calls);                             // a new draw every call (the reference: torch::randint)
    check(nsk_sample_pixels(ctx(), call_seed + 0x9e3779b9ull * (uint64_t)(K + 1), n, 0, H, 0, W, D.kf_pi.p, D.kf_pj.p));        // get_samples(0,H,0,W,100,...) :137
    check(nsk_gather_pixels(ctx(), n, D.kf_pi.p, D.kf_pj.p, D.cur.H, D.cur.W, D.cur.depth.p, nullptr, D.kf_gd.p, nullptr));
    check(nsk_rays_from_pixels(ctx(), n, D.kf_pi.p, D.kf_pj.p, fx, fy, cx, cy, D.poses.p, 0, D.kf_ro.p, D.kf_rd.p));
    std::vector<float> poses((size_t)K * 16), pct((size_t)K);
    for (int k = 0; k < K; ++k) {
        torch::Tensor m = keyframe_vector_[k].est_c2w.detach().to(torch::kCPU, torch::kFloat32).contiguous();
        std::memcpy(poses.data() + 16 * k, m.data_ptr<float>(), 16 * sizeof(float));
    }
    check(nsk_keyframe_overlap(ctx(), n, D.kf_ro.p, D.kf_rd.p, D.kf_gd.p, ns, H, W, fx, fy, cx, cy, K, poses.data(), pct.data()));
    std::vector<int> order;
    for (int k = 0; k < K; ++k) if (pct[k] > 0.f) order.push_back(k);                         // :178-179
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return pct[a] > pct[b]; });   // :186-190
    if ((int)order.size() > std::max(0, k_overlap)) order.resize((size_t)std::max(0, k_overlap));  // :194-195 (size_t underflow read as intended)
    selected_kf = order;
    last_overlap = pct;
}

void Mapper::optimize_map(int num_joint_iters_, c10::Dict<std::string, torch::Tensor>& c, torch::Tensor cur_gt_color, torch::Tensor cur_gt_depth,
                          torch::Tensor gt_cur_c2w, torch::Tensor& cur_c2w, NICE& decoders)     // Mapper.cpp:198-491
{
    (void)gt_cur_c2w;
    Dev& D = *dev;
    // window (:200-216): the mapping_window_size-2 keyframes that overlap the current frame most (over all keyframes but the last, as in
    // the original's keyframe_dict[:-1]), the last keyframe, the current frame (-1)
    std::vector<int> optimize_frame;
    const int nkf = (int)keyframe_vector.size();
    if (nkf > 1 && keyframe_selection_method == "overlap") {
        std::vector<KeyFrame> but_last(keyframe_vector.begin(), keyframe_vector.end() - 1);
        keyframe_selection_overlap(cur_gt_color, cur_gt_depth, cur_c2w, but_last, mapping_window_size - 2, optimize_frame);
    } else {
        for (int k = std::max(0, nkf - (mapping_window_size - 1)); k < nkf - 1; ++k) optimize_frame.push_back(k);
    }
    if (nkf > 0) optimize_frame.push_back(nkf - 1);                                           // :211
    const int oldest_frame = optimize_frame.empty() ? -1 : *std::min_element(optimize_frame.begin(), optimize_frame.end());
    optimize_frame.push_back(-1);                                                             // :216
    last_window = optimize_frame;
    const int nf = (int)optimize_frame.size();
    const int pixs_per_image = mapping_pixels / nf;                                           // :223 (D30)
    const int N = pixs_per_image * nf;
    if (N == 0 || num_joint_iters_ <= 0) return;

    nskh::set_bound_ctx(bound);
    check(nsk_set_render_opts(ctx(), renderer.N_samples, renderer.N_surface, renderer.lindisp, renderer.perturb, renderer.occupancy, 0));
    nskh::sync_grids(c);
    decoders.sync_to_device();
    // frame images: the current frame and every keyframe of the window, resident on the device (uploaded when first seen)
    D.cur.set(cur_gt_depth, cur_gt_color);
    if ((int)D.kf.size() < nkf) D.kf.resize((size_t)nkf);
    for (int f : optimize_frame) if (f >= 0) { if (!D.kf[f]) D.kf[f] = std::make_shared<DevFrame>(); D.kf[f]->set(keyframe_vector[f].depth, keyframe_vector[f].color); }
    if (frustum_feature_selection) {                                            // :231-281 (depth_mat = cur_gt_depth intended, D21)
        torch::Tensor pose = cur_c2w.detach().to(torch::kCPU, torch::kFloat32).contiguous();
        for (int level = 0; level < 4; ++level)
            if (!user_mask[level]) check(nsk_frustum_mask(ctx(), level, D.cur.depth.p, D.cur.H, D.cur.W, fx, fy, cx, cy, pose.data_ptr<float>(), nullptr));
    } else {
        for (int level = 0; level < 4; ++level) if (!user_mask[level]) check(nsk_set_mask(ctx(), level, nullptr));
    }
    check(nsk_decoder_set_trainable(ctx(), NSK_FINE, fix_fine ? 0 : 1));                       // :292-301
    check(nsk_decoder_set_trainable(ctx(), NSK_COLOR, fix_color ? 0 : 1));
    check(nsk_adam_reset(ctx()));                                                             // the optimiser is re-created per call (:330)
    check(nsk_zero_grads(ctx()));

    // poses of the window's frames: fixed ones as 3x4 rows, BA ones (every frame but the oldest, :305-329) as 7-vectors with Adam moments
    std::vector<float> h_pose((size_t)nf * 12), h_cam((size_t)nf * 8, 0.f);
    std::vector<int> is_ba((size_t)nf, 0);
    std::vector<nsk_frame_rays> frame_tab;
    for (int i = 0; i < nf; ++i) {
        const int f = optimize_frame[i];
        torch::Tensor c2w = (f != -1 ? keyframe_vector[f].est_c2w : cur_c2w).detach().to(torch::kCPU, torch::kFloat32).contiguous();
        std::memcpy(h_pose.data() + 12 * i, c2w.data_ptr<float>(), 12 * sizeof(float));
        if (BA && f != oldest_frame) {
            is_ba[i] = 1;
            torch::Tensor cam = get_tensor_from_camera(c2w, false).contiguous();
            std::memcpy(h_cam.data() + 8 * i, cam.data_ptr<float>(), 7 * sizeof(float));
        }
    }
    D.poses.upload(h_pose.data(), h_pose.size());
    D.cams.upload(h_cam.data(), h_cam.size());
    D.cam_m.zero((size_t)nf * 8); D.cam_v.zero((size_t)nf * 8); D.cam_g.zero((size_t)nf * 8 + 8);
    // N > 1: this rank's contiguous shard [lo, hi) of the frame-major batch ("rank r takes rays [r N / G, (r + 1) N / G)")
    const bool sharded = dist.on();
    int lo = 0, hi = N;
    if (sharded) dist.shard(N, &lo, &hi);
    const int Nl = hi - lo;
    if (sharded) { D.x_ba.zero((size_t)nf * 8 + 8); D.x_plain.zero(4); }
    std::vector<int> fr_first((size_t)nf), fr_count((size_t)nf);
    std::vector<uint8_t> fr_active((size_t)nf);
    for (int i = 0; i < nf; ++i) {                                                             // frame i's rays inside the shard
        const int a = std::max(lo, i * pixs_per_image), b = std::min(hi, (i + 1) * pixs_per_image);
        fr_first[i] = a; fr_count[i] = std::max(0, b - a); fr_active[i] = is_ba[i] ? 1 : 0;
    }
    D.rays.ensure((size_t)N); D.rays2.ensure((size_t)N); D.loss.ensure((size_t)std::max(4, num_joint_iters_));           // one loss slot per iteration
    RayBufs* bufs[2] = {&D.rays, &D.rays2};
    const bool any_ba = BA && std::count(is_ba.begin(), is_ba.end(), 1) > 0;
    int ba_step = 0;
    const uint64_t call_seed = draw_seed(rng_seed, ++draw_calls);                             // a new pixel stream every call (the reference draws
                                                                                              // fresh torch::randint pixels, utils.h:19-36, Mapper.cpp:376-414)
    check(nsk_sync(ctx()));
    const double t0 = now_us();
    auto stage_of = [&](int it) -> std::string {                                              // :351-358 (D18 intended)
        if (coarse_mapper) return "coarse";
        if (it <= int(num_joint_iters_ * middle_iter_ratio)) return "middle";
        if (it <= int(num_joint_iters_ * fine_iter_ratio)) return "fine";
        return "color";
    };
    // src/Mapper.cpp:430 renders the literal "color" whatever the stage (D19); the colour term of the loss follows `stage` (:438)
    auto render_stage_of = [&](int it) { return (render_stage_literal_color && !coarse_mapper) ? std::string("color") : stage_of(it); };
    // rays of every window frame (:376-414) for iteration `it`: pixel draw, ground-truth gather and ray generation on the device
    // (one launch for the window: it was 3 per frame + the filter = 16 launches of a 118 us iteration with five frames)
    auto draw_rays = [&](int it, RayBufs& R) {
        frame_tab.resize(nf);
        for (int i = 0; i < nf; ++i) {
            const int f = optimize_frame[i];
            const DevFrame& F = f >= 0 ? *D.kf[f] : D.cur;
            frame_tab[i] = nsk_frame_rays{F.depth.p, F.color.p, is_ba[i] ? D.cams.p + 8 * i : D.poses.p + 12 * i, is_ba[i] ? 1 : 0,
                                          call_seed + 0x100000001b3ull * (uint64_t)(it * nf + i + 1)};
        }
        check(nsk_prepare_rays(ctx(), nf, frame_tab.data(), pixs_per_image, 0, H, 0, W, D.cur.H, D.cur.W, fx, fy, cx, cy, 0, R.pi.p, R.pj.p, R.gd.p, R.gc.p,
                               R.ro.p, R.rd.p, R.keep.p));                                    // :416-427 included: rays that leave the bound before their depth
    };                                                                                        // ... are neutralised in place (no count on the host)
    int drawn_upto = -1;                                                                      // last iteration whose rays have been drawn
    for (int joint_iter = 0; joint_iter < num_joint_iters_; ++joint_iter) {
        stage = stage_of(joint_iter);
        RayBufs& R = *bufs[joint_iter & 1];
        float lr[NSK_NUM_GROUPS];
        auto st = ns_cfg["mapping"]["stage"][stage];
        lr[NSK_GROUP_DECODERS] = st["decoders_lr"].as<float>() * lr_factor;                   // :360-364
        lr[NSK_GROUP_COARSE] = st["coarse_lr"].as<float>() * lr_factor;
        lr[NSK_GROUP_MIDDLE] = st["middle_lr"].as<float>() * lr_factor;
        lr[NSK_GROUP_FINE] = st["fine_lr"].as<float>() * lr_factor;
        lr[NSK_GROUP_COLOR] = st["color_lr"].as<float>() * lr_factor;
        lr[NSK_GROUP_CAMERA] = 0.f;
        const bool ba_now = any_ba && stage == "color";                                       // :366-368
        if (drawn_upto < joint_iter) { draw_rays(joint_iter, R); drawn_upto = joint_iter; }
        // The next iteration's batch is drawn and registered before this one is optimised, so that its sampling and cell sort ride in this
        // iteration's launches (nsk_map_prepare) -- unless its rays depend on this iteration's result (bundle adjustment moves the poses)
        if (joint_iter + 1 < num_joint_iters_ && !(any_ba && stage_of(joint_iter + 1) == "color") && Nl > 0) {
            RayBufs& Rn = *bufs[(joint_iter + 1) & 1];
            draw_rays(joint_iter + 1, Rn); drawn_upto = joint_iter + 1;
            if (sharded) check(nsk_set_depth_max_batch(ctx(), Rn.gd.p, Rn.keep.p, N));          // (remembered with the registration, like the mask)
            check(nsk_set_ray_mask(ctx(), Rn.keep.p + lo));
            check(nsk_map_prepare(ctx(), nskh::stage_id(render_stage_of(joint_iter + 1)), Nl, Rn.ro.p + 3 * (size_t)lo, Rn.rd.p + 3 * (size_t)lo, Rn.gd.p + lo, -1.f,
                                  NSK_GRAD_GRIDS | NSK_GRAD_DECODERS));
        }
        unsigned flags = NSK_GRAD_GRIDS | NSK_GRAD_DECODERS | (ba_now ? NSK_GRAD_RAYS : 0u);
        const std::string render_stage = render_stage_of(joint_iter);
        if (!sharded) {
            check(nsk_set_ray_mask(ctx(), R.keep.p));
            check(nsk_map_step(ctx(), nskh::stage_id(render_stage), N, R.ro.p, R.rd.p, R.gd.p, R.gc.p, -1.f, w_color_loss, stage == "color" ? 1 : 0, flags,
                               D.loss.p + joint_iter, nullptr, nullptr, nullptr, ba_now ? R.g_ro.p : nullptr, ba_now ? R.g_rd.p : nullptr));      // :430-444
            check(nsk_set_ray_mask(ctx(), nullptr));
            check(nsk_adam_step(ctx(), lr, 0.9f, 0.999f, 1e-8f));                             // :445-446
            if (ba_now)                                                                       // pose gradients through the ray generator and quad2rotation + Adam: every frame of the window in one launch
                check(nsk_pose_step_multi(ctx(), nf, fr_first.data(), fr_count.data(), fr_active.data(), R.pi.p, R.pj.p, fx, fy, cx, cy, 0, R.g_ro.p, R.g_rd.p,
                                          D.cams.p, D.cam_m.p, D.cam_v.p, BA_cam_lr, 0.9f, 0.999f, 1e-8f, ++ba_step, D.cam_g.p, nullptr, 0));
            continue;
        }
        // ---- N > 1: the shard's step, one exchange, the replicated optimiser steps ------------------------------------------------
        float* xt = ba_now ? D.x_ba.p : D.x_plain.p;                                          // what travels beside the grids this iteration
        float* loss_slot = ba_now ? xt + (size_t)nf * 8 : xt;
        check(nsk_set_depth_max_batch(ctx(), R.gd.p, R.keep.p, N));                           // max(gt_depth) over the WHOLE batch (Renderer.cpp:76,93)
        check(nsk_set_ray_mask(ctx(), R.keep.p + lo));
        if (Nl > 0)
            check(nsk_map_step(ctx(), nskh::stage_id(render_stage), Nl, R.ro.p + 3 * (size_t)lo, R.rd.p + 3 * (size_t)lo, R.gd.p + lo, R.gc.p + 3 * (size_t)lo, -1.f,
                               w_color_loss, stage == "color" ? 1 : 0, flags, loss_slot, nullptr, nullptr, nullptr,
                               ba_now ? R.g_ro.p + 3 * (size_t)lo : nullptr, ba_now ? R.g_rd.p + 3 * (size_t)lo : nullptr));
        check(nsk_set_ray_mask(ctx(), nullptr));
        check(nsk_set_depth_max_batch(ctx(), nullptr, nullptr, 0));
        if (ba_now)                                                                           // this shard's part of every frame's pose gradient (no step yet) + its kept-ray count
            check(nsk_pose_step_multi(ctx(), nf, fr_first.data(), fr_count.data(), fr_active.data(), R.pi.p, R.pj.p, fx, fy, cx, cy, 0, R.g_ro.p, R.g_rd.p,
                                      D.cams.p, nullptr, nullptr, 0.f, 0.9f, 0.999f, 1e-8f, 0, xt, R.keep.p + lo, Nl));
        check(nsk_grad_extra(ctx(), xt, ba_now ? (size_t)nf * 8 + 8 : 4));
        dist.allreduce_grads();                                                               // the ONE exchange of the iteration
        check(nsk_grad_extra(ctx(), nullptr, 0));
        HIPOK(hipMemcpyAsync(D.loss.p + joint_iter, loss_slot, sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)nsk_stream(ctx())));
        check(nsk_adam_step(ctx(), lr, 0.9f, 0.999f, 1e-8f));
        if (ba_now)                                                                           // every pose at once on the summed gradients (zero for the frames left fixed: no move)
            check(nsk_adam_vector(ctx(), nf * 8, D.cams.p, xt, D.cam_m.p, D.cam_v.p, BA_cam_lr, 0.9f, 0.999f, 1e-8f, ++ba_step));
    }
    check(nsk_sync(ctx()));
    last_iter_us = (now_us() - t0) / num_joint_iters_;
    last_losses.assign((size_t)num_joint_iters_, 0.f);
    D.loss.download(last_losses.data(), (size_t)num_joint_iters_);                            // the only per-call device-to-host traffic besides the results
    last_loss = last_losses.back();
    nskh::fetch_grids(c);                                                                     // :448-464 (once instead of per iteration)
    decoders.fetch_from_device(!fix_fine, !fix_color);
    last_ba_grad.assign((size_t)nf * 7, 0.f);
    if (any_ba) {                                                                             // :467-489
        std::vector<float> hg((size_t)nf * 8 + 8);
        (sharded ? D.x_ba : D.cam_g).download(hg.data(), hg.size());
        if (sharded) last_kept_rays = hg[(size_t)nf * 8 + 1];
        for (int i = 0; i < nf; ++i) for (int k = 0; k < 7; ++k) last_ba_grad[(size_t)i * 7 + k] = hg[(size_t)i * 8 + k];
        D.cams.download(h_cam.data(), h_cam.size());
        torch::Tensor bottom = torch::tensor({{0.f, 0.f, 0.f, 1.f}});
        for (int i = 0; i < nf; ++i) {
            if (!is_ba[i]) continue;
            torch::Tensor cam = torch::from_blob(h_cam.data() + 8 * i, {7}, torch::kFloat32).clone();
            torch::Tensor c2w = torch::cat({get_camera_from_tensor(cam), bottom}, 0);
            const int f = optimize_frame[i];
            if (f != -1) keyframe_vector[f].est_c2w = c2w; else cur_c2w = c2w;                // D24: .back()
        }
    }
}

void Mapper::run(NICE& decoders, c10::Dict<std::string, torch::Tensor>& c, std::vector<torch::Tensor>& estimate_c2w_vec, torch::Tensor gt_color_t,
                 torch::Tensor gt_depth_t, torch::Tensor gt_c2w_t, int idx, int n_imgs)        // Mapper.cpp:493-552
{
    int outer_joint_iters = 1, iters;
    if (!first_frame) {                                                                       // D29: the flag is cleared after the first frame
        lr_factor = ns_cfg["mapping"]["lr_factor"].as<float>();
        iters = ns_cfg["mapping"]["iters"].as<int>();
        if ((idx == n_imgs - 1) && color_refine && !coarse_mapper) {
            outer_joint_iters = 5; mapping_window_size *= 2; middle_iter_ratio = 0.0; fine_iter_ratio = 0.0;
            iters *= 5; fix_color = true; frustum_feature_selection = false;
        }
    } else {
        lr_factor = ns_cfg["mapping"]["lr_first_factor"].as<float>();
        iters = ns_cfg["mapping"]["iters_first"].as<int>();
    }
    torch::Tensor cur_c2w = estimate_c2w_vec[idx];
    iters = iters / outer_joint_iters;
    for (int outer = 0; outer < outer_joint_iters; ++outer) {
        BA = (keyframe_lvector.size() > 4) && ns_cfg["mapping"]["BA"].as<bool>() && !coarse_mapper;                      // :530
        optimize_map(iters, c, gt_color_t, gt_depth_t, gt_c2w_t, cur_c2w, decoders);
        if (BA) estimate_c2w_vec[idx] = cur_c2w;
        if (outer == outer_joint_iters - 1) {
            if (((idx % keyframe_every == 0) || (idx == n_imgs - 2)) &&
                std::find(keyframe_lvector.begin(), keyframe_lvector.end(), idx) == keyframe_lvector.end()) {             // :539-549
                keyframe_lvector.push_back(idx);
                KeyFrame kf;
                kf.gt_c2w = gt_c2w_t; kf.idx = idx; kf.color = gt_color_t; kf.depth = gt_depth_t; kf.est_c2w = cur_c2w;
                keyframe_vector.push_back(kf);
            }
        }
    }
    first_frame = false;
}
