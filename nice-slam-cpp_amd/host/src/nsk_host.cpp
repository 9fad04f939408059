// nsk_host.cpp -- C++ host classes with the reference's surface (Renderer / NICE / MLP / Tracker / Mapper, utils.h
// helpers) marshalling into the C-ABI of include/nsk.h.  libtorch is used as a host-side tensor container and for
// host-side bookkeeping (pixel sampling, the 7-parameter pose optimiser); no libtorch GPU compute op is called.
// Each function cites the reference code whose behaviour it reproduces (intended semantics).
#include <hip/hip_runtime_api.h>

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
// Tracker
// ---------------------------------------------------------------------------------------------------------
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
}
Tracker::~Tracker() {}
void Tracker::update_para_from_mapping() {}
void Tracker::set_bound(torch::Tensor b) { bound = b.detach().to(torch::kCPU, torch::kFloat32).clone(); renderer.set_bound(bound); }

torch::Tensor Tracker::optimize_cam_in_batch(torch::Tensor cam_tensor, torch::Tensor gt_color, torch::Tensor gt_depth, int batch_size,
                                             torch::optim::Adam& optimizer, NICE decoders)      // Tracker.cpp:41-89
{
    optimizer.zero_grad();
    torch::Tensor cam_cpu = cam_tensor.detach().to(torch::kCPU, torch::kFloat32).contiguous();
    torch::Tensor c2w = get_camera_from_tensor(cam_cpu);
    torch::Tensor ro, rd, gd, gc, pi, pj;
    get_samples(ignore_edge_h, H - ignore_edge_h, ignore_edge_w, W - ignore_edge_w, batch_size, H, W, fx, fy, cx, cy, c2w, gt_depth, gt_color,
                ro, rd, gd, gc, &pi, &pj);
    torch::Tensor keep = inside_mask(bound, ro, rd, gd);                                      // :48-58 (detached, D7)
    ro = ro.index({keep}).contiguous(); rd = rd.index({keep}).contiguous(); gd = gd.index({keep}).contiguous();
    gc = gc.index({keep}).contiguous(); pi = pi.index({keep}).contiguous(); pj = pj.index({keep}).contiguous();
    const int N = (int)ro.size(0);
    TORCH_CHECK(N > 0, "optimize_cam_in_batch: every sampled ray was rejected by the bound test");
    // render + loss + backward onto the rays in the kernels, then the pose chain (rays -> c2w -> quaternion/translation)
    nskh::set_bound_ctx(bound);
    check(nsk_set_render_opts(ctx(), renderer.N_samples, renderer.N_surface, renderer.lindisp, renderer.perturb, renderer.occupancy, 0));
    nskh::sync_grids(c);
    decoders.sync_to_device();
    DevBuf d_ro, d_rd, d_gd, d_gc, d_gro, d_grd, d_loss, d_cam, d_gc2w, d_gcam, d_pi, d_pj;
    d_ro.upload(ro); d_rd.upload(rd); d_gd.upload(gd); d_gc.upload(gc); d_cam.upload(cam_cpu);
    d_gro.ensure((size_t)N * 3); d_grd.ensure((size_t)N * 3); d_loss.ensure(4); d_gc2w.ensure(12); d_gcam.ensure(8);
    d_pi.ensure(N); d_pj.ensure(N);      // int32 payloads in float-sized slots
    hipMemcpy(d_pi.p, pi.data_ptr<int32_t>(), N * sizeof(int32_t), hipMemcpyHostToDevice);
    hipMemcpy(d_pj.p, pj.data_ptr<int32_t>(), N * sizeof(int32_t), hipMemcpyHostToDevice);
    check(nsk_track_step(ctx(), NSK_COLOR, N, d_ro.p, d_rd.p, d_gd.p, d_gc.p, -1.f, w_color_loss, use_color_in_tracking ? 1 : 0,
                         handle_dynamic ? 1 : 0, 1, NSK_GRAD_RAYS, d_loss.p, d_gro.p, d_grd.p));      // stage "color": :61 (D19)
    check(nsk_rays_backward(ctx(), N, (const int32_t*)d_pi.p, (const int32_t*)d_pj.p, fx, fy, cx, cy, 0, d_gro.p, d_grd.p, d_gc2w.p));
    check(nsk_camera_backward(ctx(), d_cam.p, d_gc2w.p, d_gcam.p));
    torch::Tensor g = d_gcam.download({8}).index({Slice(None, 7)}).clone();
    torch::Tensor loss = d_loss.download({4}).index({0}).clone();
    cam_tensor.mutable_grad() = g.to(cam_tensor.device());                                    // loss.backward() :84
    optimizer.step();                                                                         // :85
    optimizer.zero_grad();
    return loss;
}

void Tracker::run(NICE decoders, torch::Tensor gt_color_t, torch::Tensor gt_depth_t, torch::Tensor gt_c2w_t, int idx_)       // Tracker.cpp:92-113
{
    idx = idx_;
    torch::Tensor camera_tensor = get_tensor_from_camera(gt_c2w_t, false).requires_grad_(true);   // :100 (initialised from the GT pose, D25)
    std::vector<torch::Tensor> cam_para_list{camera_tensor};
    torch::optim::Adam optimizer(cam_para_list, torch::optim::AdamOptions(lr));               // config tracking.lr (D25)
    for (int i = 0; i < num_cam_iters; ++i) {
        auto loss = optimize_cam_in_batch(camera_tensor, gt_color_t, gt_depth_t, tracking_pixels, optimizer, decoders);
        std::cout << "loss: " << loss.item<float>() << std::endl;                             // :111
    }
    last_camera_tensor = camera_tensor.detach().clone();
}

// ---------------------------------------------------------------------------------------------------------
// Mapper
// ---------------------------------------------------------------------------------------------------------
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
}
Mapper::~Mapper() {}
void Mapper::set_bound(torch::Tensor b) { bound = b.detach().to(torch::kCPU, torch::kFloat32).clone(); renderer.set_bound(bound); }

void Mapper::set_frustum_mask(const std::string& key, torch::Tensor mask)
{
    int level = key == "grid_coarse" ? 0 : key == "grid_middle" ? 1 : key == "grid_fine" ? 2 : key == "grid_color" ? 3 : -1;
    TORCH_CHECK(level >= 0, "unknown grid key ", key);
    user_mask[level] = true;
    if (!mask.defined()) { check(nsk_set_mask(ctx(), level, nullptr)); return; }
    torch::Tensor m = mask.to(torch::kCPU, torch::kUInt8).contiguous();
    check(nsk_set_mask(ctx(), level, m.data_ptr<uint8_t>()));
}

void Mapper::optimize_map(int num_joint_iters_, c10::Dict<std::string, torch::Tensor>& c, torch::Tensor cur_gt_color, torch::Tensor cur_gt_depth,
                          torch::Tensor gt_cur_c2w, torch::Tensor& cur_c2w, NICE& decoders)     // Mapper.cpp:198-491
{
    (void)gt_cur_c2w;
    // window (:200-216): the mapping_window_size-2 keyframes that overlap the current frame most (keyframe_selection_overlap,
    // :132-196, over all keyframes but the last as in the original's keyframe_dict[:-1]), the last keyframe, the current frame (-1)
    std::vector<int> optimize_frame;
    const int nkf = (int)keyframe_vector.size();
    if (nkf > 1 && keyframe_selection_method == "overlap") {
        torch::Tensor ro, rd, gd, gc;
        get_samples(0, H, 0, W, 100, H, W, fx, fy, cx, cy, cur_c2w, cur_gt_depth, cur_gt_color, ro, rd, gd, gc);     // :137
        DevBuf d_ro, d_rd, d_gd;
        d_ro.upload(ro); d_rd.upload(rd); d_gd.upload(gd);
        std::vector<float> poses((size_t)(nkf - 1) * 16), pct(nkf - 1);
        for (int k = 0; k < nkf - 1; ++k) {
            torch::Tensor m = keyframe_vector[k].est_c2w.detach().to(torch::kCPU, torch::kFloat32).contiguous();
            std::memcpy(poses.data() + 16 * k, m.data_ptr<float>(), 16 * sizeof(float));
        }
        check(nsk_keyframe_overlap(ctx(), (int)ro.size(0), d_ro.p, d_rd.p, d_gd.p, 16, H, W, fx, fy, cx, cy, nkf - 1, poses.data(), pct.data()));
        std::vector<int> order;
        for (int k = 0; k < nkf - 1; ++k) if (pct[k] > 0.f) order.push_back(k);                                     // :178-179
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return pct[a] > pct[b]; });                  // :186-190
        const int num = std::max(0, mapping_window_size - 2);
        if ((int)order.size() > num) order.resize(num);                                                             // :194-195 (size_t underflow read as intended)
        optimize_frame = order;
        last_overlap = pct;
    } else {
        for (int k = std::max(0, nkf - (mapping_window_size - 1)); k < nkf - 1; ++k) optimize_frame.push_back(k);
    }
    if (nkf > 0) optimize_frame.push_back(nkf - 1);                                           // :211
    int oldest_frame = optimize_frame.empty() ? -1 : *std::min_element(optimize_frame.begin(), optimize_frame.end());
    optimize_frame.push_back(-1);                                                             // :216
    last_window = optimize_frame;
    const int pixs_per_image = mapping_pixels / (int)optimize_frame.size();                   // :223 (D30)

    nskh::set_bound_ctx(bound);
    check(nsk_set_render_opts(ctx(), renderer.N_samples, renderer.N_surface, renderer.lindisp, renderer.perturb, renderer.occupancy, 0));
    nskh::sync_grids(c);
    decoders.sync_to_device();
    if (frustum_feature_selection) {                                            // :231-281 (depth_mat = cur_gt_depth intended, D21)
        torch::Tensor dimg = cur_gt_depth.detach().to(torch::kCPU, torch::kFloat32).contiguous();
        torch::Tensor pose = cur_c2w.detach().to(torch::kCPU, torch::kFloat32).contiguous();
        DevBuf d_img; d_img.upload(dimg);
        for (int level = 0; level < 4; ++level)
            if (!user_mask[level]) check(nsk_frustum_mask(ctx(), level, d_img.p, (int)dimg.size(0), (int)dimg.size(1), fx, fy, cx, cy, pose.data_ptr<float>(), nullptr));
        check(nsk_sync(ctx()));
    } else {
        for (int level = 0; level < 4; ++level) if (!user_mask[level]) check(nsk_set_mask(ctx(), level, nullptr));
    }
    check(nsk_decoder_set_trainable(ctx(), NSK_FINE, fix_fine ? 0 : 1));                       // :292-301
    check(nsk_decoder_set_trainable(ctx(), NSK_COLOR, fix_color ? 0 : 1));
    check(nsk_adam_reset(ctx()));                                                             // the optimiser is re-created per call (:330)
    check(nsk_zero_grads(ctx()));

    // bundle adjustment: one 7-vector per window frame except the oldest (:305-329), optimised on the host (7 parameters each)
    std::vector<torch::Tensor> camera_tensor_list;
    if (BA) {
        for (int frame : optimize_frame) {
            if (frame == oldest_frame) continue;
            torch::Tensor c2w = frame != -1 ? keyframe_vector[frame].est_c2w : cur_c2w;
            camera_tensor_list.push_back(get_tensor_from_camera(c2w, false).requires_grad_(true));
        }
    }
    std::unique_ptr<torch::optim::Adam> cam_opt;
    if (BA && !camera_tensor_list.empty()) cam_opt.reset(new torch::optim::Adam(camera_tensor_list, torch::optim::AdamOptions(0.0)));

    DevBuf d_ro, d_rd, d_gd, d_gc, d_gro, d_grd, d_loss;
    d_loss.ensure(4);
    for (int joint_iter = 0; joint_iter < num_joint_iters_; ++joint_iter) {
        if (coarse_mapper) stage = "coarse";                                                  // :351-358 (D18 intended)
        else if (joint_iter <= int(num_joint_iters_ * middle_iter_ratio)) stage = "middle";
        else if (joint_iter <= int(num_joint_iters_ * fine_iter_ratio)) stage = "fine";
        else stage = "color";
        float lr[NSK_NUM_GROUPS];
        auto st = ns_cfg["mapping"]["stage"][stage];
        lr[NSK_GROUP_DECODERS] = st["decoders_lr"].as<float>() * lr_factor;                   // :360-364
        lr[NSK_GROUP_COARSE] = st["coarse_lr"].as<float>() * lr_factor;
        lr[NSK_GROUP_MIDDLE] = st["middle_lr"].as<float>() * lr_factor;
        lr[NSK_GROUP_FINE] = st["fine_lr"].as<float>() * lr_factor;
        lr[NSK_GROUP_COLOR] = st["color_lr"].as<float>() * lr_factor;
        lr[NSK_GROUP_CAMERA] = 0.f;
        const bool ba_now = cam_opt && stage == "color";                                      // :366-368
        if (cam_opt) {
            static_cast<torch::optim::AdamOptions&>(cam_opt->param_groups()[0].options()).lr(ba_now ? BA_cam_lr : 0.0);
            cam_opt->zero_grad();
        }
        // rays of every window frame (:376-414), sampled on the host like the reference does
        std::vector<torch::Tensor> v_ro, v_rd, v_gd, v_gc, v_pi, v_pj;
        std::vector<int> v_cam;      // camera tensor index per frame, -1 = fixed pose
        int camera_tensor_id = 0;
        for (int frame : optimize_frame) {
            torch::Tensor gt_depth = frame != -1 ? keyframe_vector[frame].depth : cur_gt_depth;
            torch::Tensor gt_color = frame != -1 ? keyframe_vector[frame].color : cur_gt_color;
            torch::Tensor c2w; int cam_id = -1;
            if (BA && frame != oldest_frame) { cam_id = camera_tensor_id++; c2w = get_camera_from_tensor(camera_tensor_list[cam_id].detach()); }
            else c2w = frame != -1 ? keyframe_vector[frame].est_c2w : cur_c2w;
            torch::Tensor ro, rd, gd, gc, pi, pj;
            get_samples(0, H, 0, W, pixs_per_image, H, W, fx, fy, cx, cy, c2w, gt_depth, gt_color, ro, rd, gd, gc, &pi, &pj);
            torch::Tensor keep = inside_mask(bound, ro, rd, gd);                              // :416-427, per frame (same rays kept)
            v_ro.push_back(ro.index({keep})); v_rd.push_back(rd.index({keep})); v_gd.push_back(gd.index({keep}));
            v_gc.push_back(gc.index({keep})); v_pi.push_back(pi.index({keep})); v_pj.push_back(pj.index({keep}));
            v_cam.push_back(cam_id);
        }
        torch::Tensor ro = torch::cat(v_ro).contiguous(), rd = torch::cat(v_rd).contiguous();
        torch::Tensor gd = torch::cat(v_gd).contiguous(), gc = torch::cat(v_gc).contiguous();
        const int N = (int)ro.size(0);
        if (N == 0) continue;
        d_ro.upload(ro); d_rd.upload(rd); d_gd.upload(gd); d_gc.upload(gc);
        unsigned flags = NSK_GRAD_GRIDS | NSK_GRAD_DECODERS;
        if (ba_now) { flags |= NSK_GRAD_RAYS; d_gro.ensure((size_t)N * 3); d_grd.ensure((size_t)N * 3); }
        // the reference renders with the literal "color" whatever the stage (:430, D19); the intended graph renders `stage`
        check(nsk_map_step(ctx(), nskh::stage_id(stage), N, d_ro.p, d_rd.p, d_gd.p, d_gc.p, -1.f, w_color_loss, stage == "color" ? 1 : 0, flags,
                           d_loss.p, nullptr, nullptr, nullptr, ba_now ? d_gro.p : nullptr, ba_now ? d_grd.p : nullptr));      // :430-444
        if (ba_now) {         // pose gradients per frame through the ray generator and quad2rotation
            torch::Tensor g_ro = d_gro.download({N, 3}), g_rd = d_grd.download({N, 3});
            int64_t off = 0;
            for (size_t f = 0; f < v_ro.size(); ++f) {
                int64_t n = v_ro[f].size(0);
                if (v_cam[f] >= 0 && n > 0) {
                    torch::Tensor cam = camera_tensor_list[v_cam[f]].detach();
                    torch::Tensor i = v_pi[f].to(torch::kFloat32), j = v_pj[f].to(torch::kFloat32);
                    torch::Tensor dirs = torch::stack({(i - cx) / fx, -(j - cy) / fy, -torch::ones_like(i)}, -1);        // [n,3]
                    torch::Tensor gR = torch::matmul(g_rd.slice(0, off, off + n).t(), dirs);                              // [3,3]
                    torch::Tensor gt = g_ro.slice(0, off, off + n).sum(0);                                                // [3]
                    torch::Tensor q = cam.clone().requires_grad_(true);                                                   // 9x4 Jacobian of quad2rotation on the host
                    torch::Tensor RT = get_camera_from_tensor(q);
                    ((RT.index({Slice(), Slice(None, 3)}) * gR).sum() + (RT.index({Slice(), 3}) * gt).sum()).backward();
                    camera_tensor_list[v_cam[f]].mutable_grad() = q.grad().clone();
                }
                off += n;
            }
        }
        check(nsk_adam_step(ctx(), lr, 0.9f, 0.999f, 1e-8f));                                 // :445-446
        if (cam_opt) { if (ba_now) cam_opt->step(); cam_opt->zero_grad(); }
    }
    last_loss = d_loss.download({4}).index({0}).item<float>();
    nskh::fetch_grids(c);                                                                     // :448-464 (once instead of per iteration)
    decoders.fetch_from_device(!fix_fine, !fix_color);
    if (BA) {                                                                                 // :467-489
        torch::Tensor bottom = torch::tensor({{0.f, 0.f, 0.f, 1.f}});
        int camera_tensor_id = 0;
        for (int frame : optimize_frame) {
            if (frame == oldest_frame) continue;
            torch::Tensor c2w = torch::cat({get_camera_from_tensor(camera_tensor_list[camera_tensor_id++].detach()), bottom}, 0);
            if (frame != -1) keyframe_vector[frame].est_c2w = c2w; else cur_c2w = c2w;        // D24: .back()
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
