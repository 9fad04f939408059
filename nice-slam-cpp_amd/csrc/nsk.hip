// nsk.hip -- C-ABI (include/nsk.h) of the MI355X-native NICE-SLAM hot path: context, data layouts, launches.
// Built only for gfx950:  hipcc --offload-arch=gfx950 -O3 -shared -fPIC nsk.hip -o libnsk.so
#include "../../include/nsk.h"
#include "nsk_device.h"
#include "nsk_train.h"
#include "nsk_bf16.h"

#include <dlfcn.h>
#include <cstdarg>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#define NSK_VERSION 100

static thread_local std::string g_err;
static int fail(const char* fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    g_err = buf;
    return -1;
}
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)
#define CHK(x) do { int r_ = (x); if (r_ != 0) return r_; } while (0)

// ---------------------------------------------------------------------------------------------------------
// small kernels that only the host file needs
// ---------------------------------------------------------------------------------------------------------
__global__ void k_pack(float* __restrict__ img, const int* __restrict__ idx, const float* __restrict__ P, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { int k = idx[i]; img[i] = k >= 0 ? P[k] : 0.f; }
}

// torch::optim::Adam (reference src/Mapper.cpp:330,445-446); mask is per voxel (32 floats) or nullptr; zeroes g
__global__ void k_adam(int n, float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                       const uint8_t* __restrict__ mask, float step_size, float bc2s, float b1, float b2, float eps)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;       // one f4 per thread
    if (4 * i >= n) return;
    f4* g4 = reinterpret_cast<f4*>(g) + i;
    if (mask && !mask[i >> 3]) { *g4 = (f4)(0.f); return; }
    f4 gg = *g4, pp = reinterpret_cast<f4*>(p)[i], mm = reinterpret_cast<f4*>(m)[i], vv = reinterpret_cast<f4*>(v)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        mm[k] = b1 * mm[k] + (1.f - b1) * gg[k];
        vv[k] = b2 * vv[k] + (1.f - b2) * gg[k] * gg[k];
        float denom = sqrtf(vv[k]) / bc2s + eps;
        pp[k] -= step_size * (mm[k] / denom);
    }
    reinterpret_cast<f4*>(p)[i] = pp; reinterpret_cast<f4*>(m)[i] = mm; reinterpret_cast<f4*>(v)[i] = vv;
    *g4 = (f4)(0.f);
}

struct AdamSeg { float* p; float* g; float* m; float* v; const int* idx; int nidx; int n; float step_size, bc2s; int blk_end;      // idx: the marked voxels (nullptr = all)
                 const int* inv_f; const int* inv_b; float* fimg; float* bimg;      // decoders: image position of each parameter (-1 none)
                 const int* inv16; unsigned short* img16; float* img16_tail; int tail_off; int np16;
                 const int* invh; unsigned short* imgh; float* imgh_tail; int btail_off;      // fp16 backward image (MlpBwdImgH) and its fp32 tail
                 const float* slabs; int nslabs, slab_stride; };                      // pending per-workgroup gradient slabs (k_decode_bwd_multi)   // bf16 3-piece image (nsk_bf16.h), its fp32 tail
struct AdamArgs { AdamSeg s[8]; int n; float b1, b2, eps;
                  PlaceArgs place; int adam_blocks; };       // optional (place.nblocks > 0): the NEXT batch's cell-sort placement rides behind the segments (nsk_map_prepare)
// all parameter groups of one optimiser step in one launch (3 grid levels + trainable decoders)
__global__ void k_adam_multi(AdamArgs A)
{
    if (A.place.nblocks > 0 && (int)blockIdx.x >= A.adam_blocks) { sort_place_body(A.place, (int)blockIdx.x - A.adam_blocks); return; }
    int r = 0;
    while (r < A.n - 1 && (int)blockIdx.x >= A.s[r].blk_end) ++r;
    const AdamSeg& S = A.s[r];
    const int b0 = r == 0 ? 0 : A.s[r - 1].blk_end;
    int i = (blockIdx.x - b0) * blockDim.x + threadIdx.x;
    f4 extra = (f4)(0.f);
    // a trainable decoder's parameters are walked 8 float4 per block; the thread that will update float4 `i` fetches everything that does not
    // depend on the gradient -- moments, parameter, the image positions of its four parameters -- TOGETHER with the slab loads below: the launch was
    // three dependent round trips (slabs -> moments / parameter -> index tables -> image stores), and its time is latency, not traffic
    f4 pre_m = (f4)(0.f), pre_v = (f4)(0.f), pre_p = (f4)(0.f);
    int pf[4] = {-1, -1, -1, -1}, pb[4] = {-1, -1, -1, -1}, p16[4] = {-1, -1, -1, -1}, ph[4] = {-1, -1, -1, -1};
    bool pre = false;
    if (S.slabs) {      // decoder whose gradient still sits in per-workgroup slabs: 8 float4 per block, 32 thread groups sum 1/32 of the slabs each
        // (all of a thread's ~6 slab reads are in flight at once: this sum, not the grids' Adam traffic, was most of the launch's time --
        // 18 us with 8 groups reading ~24 slabs one after the other, 14 us with four reads in flight, see profiles/)
        __shared__ f4 red[32][8];
        const int pi = threadIdx.x & 7, sg = threadIdx.x >> 3;
        i = (blockIdx.x - b0) * 8 + pi;
        f4 part = (f4)(0.f);
        if (4 * i < S.n) {
            if (sg == 0) {
                pre = true;
                pre_m = reinterpret_cast<const f4*>(S.m)[i]; pre_v = reinterpret_cast<const f4*>(S.v)[i]; pre_p = reinterpret_cast<const f4*>(S.p)[i];
                if (S.inv_f) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        pf[k] = S.inv_f[4 * i + k]; pb[k] = S.inv_b[4 * i + k];
                        if (S.inv16) p16[k] = S.inv16[4 * i + k];
                        if (S.invh) ph[k] = S.invh[4 * i + k];
                    }
                }
            }
            const float* base = S.slabs + 4 * i;
#pragma unroll 8
            for (int sl = sg; sl < S.nslabs; sl += 32) part += *reinterpret_cast<const f4*>(base + (size_t)sl * S.slab_stride);
        }
        red[sg][pi] = part;
        __syncthreads();
        if (sg != 0) return;
#pragma unroll
        for (int k = 0; k < 32; ++k) extra += red[k][pi];
    }
    if (S.idx) {                  // a masked level: the launch covers the marked voxels only (8 float4 each); unmarked ones are never touched
        if (i >= S.nidx * 8) return;
        i = S.idx[i >> 3] * 8 + (i & 7);
    }
    if (4 * i >= S.n) return;
    f4* g4 = reinterpret_cast<f4*>(S.g) + i;
    const f4 g0 = *g4;
    f4 gg = g0 + extra, mm = pre ? pre_m : reinterpret_cast<f4*>(S.m)[i], vv = pre ? pre_v : reinterpret_cast<f4*>(S.v)[i];
    // a parameter that never received a gradient (g = m = v = 0) does not move under Adam: skip its five memory operations
    if (!S.inv_f && gg[0] == 0.f && gg[1] == 0.f && gg[2] == 0.f && gg[3] == 0.f && mm[0] == 0.f && mm[1] == 0.f && mm[2] == 0.f && mm[3] == 0.f &&
        vv[0] == 0.f && vv[1] == 0.f && vv[2] == 0.f && vv[3] == 0.f) return;
    f4 pp = pre ? pre_p : reinterpret_cast<f4*>(S.p)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        mm[k] = A.b1 * mm[k] + (1.f - A.b1) * gg[k];
        vv[k] = A.b2 * vv[k] + (1.f - A.b2) * gg[k] * gg[k];
        float denom = sqrtf(vv[k]) / S.bc2s + A.eps;
        pp[k] -= S.step_size * (mm[k] / denom);
    }
    reinterpret_cast<f4*>(S.p)[i] = pp; reinterpret_cast<f4*>(S.m)[i] = mm; reinterpret_cast<f4*>(S.v)[i] = vv;
    if (g0[0] != 0.f || g0[1] != 0.f || g0[2] != 0.f || g0[3] != 0.f) *g4 = (f4)(0.f);
    if (S.inv_f) {      // keep the MFMA fragment images of a trainable decoder in step with its parameters
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int fi = pre ? pf[k] : S.inv_f[4 * i + k], bi = pre ? pb[k] : S.inv_b[4 * i + k];
            if (fi >= 0) S.fimg[fi] = pp[k];
            if (bi >= 0) S.bimg[bi] = pp[k];
            if (S.inv16) {
                const int t = pre ? p16[k] : S.inv16[4 * i + k];
                if (t >= 0) store_pieces(S.img16, t, pp[k], S.np16);
                if (fi >= S.tail_off) S.img16_tail[fi - S.tail_off] = pp[k];
            }
            if (S.invh) {
                const int t = pre ? ph[k] : S.invh[4 * i + k];
                if (t >= 0) store_pieces(S.imgh, t, pp[k], 2);
                if (bi >= S.btail_off) S.imgh_tail[bi - S.btail_off] = pp[k];
            }
        }
    }
}

// Mask-compacted gradient exchange (multi-GPU): every rank holds the same optimiser mask, so only the marked voxels' gradients need to
// travel.  k_mask_index lists them in ascending order (one workgroup, deterministic: every rank must build the same list),
// k_grad_gather / k_grad_scatter move their 32-float lines between the dense gradient slab and the packed exchange buffer.
__global__ __launch_bounds__(1024) void k_mask_index(int nvox, const uint8_t* __restrict__ mask, int* __restrict__ idx, int* __restrict__ count)
{
    __shared__ int wsum[16];
    __shared__ int base;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int v0 = 0; v0 < nvox; v0 += 1024) {
        const int v = v0 + threadIdx.x;
        const bool on = v < nvox && mask[v];
        const unsigned long long b = __builtin_amdgcn_ballot_w64(on);
        if (lane == 0) wsum[wave] = __builtin_popcountll(b);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        if (on) idx[off + __builtin_popcountll(b & ((1ull << lane) - 1ull))] = v;
        __syncthreads();
        if (threadIdx.x == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += wsum[w]; base += t; }
        __syncthreads();
    }
    if (threadIdx.x == 0) *count = base;
}
__global__ void k_grad_gather(int n, const int* __restrict__ idx, const float* __restrict__ g, float* __restrict__ packed)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;          // one float4 per thread, 8 per voxel
    if (t >= n * 8) return;
    const int v = idx ? idx[t >> 3] : (t >> 3);
    reinterpret_cast<f4*>(packed)[t] = reinterpret_cast<const f4*>(g)[(size_t)v * 8 + (t & 7)];
}
__global__ void k_grad_scatter(int n, const int* __restrict__ idx, const float* __restrict__ packed, float* __restrict__ g)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * 8) return;
    const int v = idx ? idx[t >> 3] : (t >> 3);
    reinterpret_cast<f4*>(g)[(size_t)v * 8 + (t & 7)] = reinterpret_cast<const f4*>(packed)[t];
}

// the whole packed exchange buffer in ONE launch each way (it was one launch per level plus two copies: eleven small operations per
// step around the all-reduce): segments = grid levels (32 floats per listed voxel), trainable decoders and the loss scalars (plain runs)
struct XSeg { const int* idx; int n4; float* slab; float* buf; int blk_end;       // n4 = float4 count of the segment; idx != nullptr: voxel list, 8 float4 per voxel
              const float* slabs; int nslabs, slab_stride; };                     // gather only: a decoder whose gradient still sits in per-workgroup slabs
struct XArgs { XSeg s[10]; int n; int gather; };
__global__ __launch_bounds__(256) void k_xchg_multi(XArgs A)
{
    int r = 0;
    while (r < A.n - 1 && (int)blockIdx.x >= A.s[r].blk_end) ++r;
    const XSeg& S = A.s[r];
    const int b0 = r ? A.s[r - 1].blk_end : 0;
    if (S.slabs) {          // the pending decoder-gradient slabs are summed on the way into the buffer (the sum k_adam_multi does on one GPU):
        __shared__ f4 red[32][8];                 // 8 float4 per block, 32 thread groups sum 1/32 of the slabs each
        const int pi = threadIdx.x & 7, sg = threadIdx.x >> 3;
        const int i = (blockIdx.x - b0) * 8 + pi;
        f4 part = (f4)(0.f);
        if (i < S.n4) {
            const float* base = S.slabs + 4 * i;
#pragma unroll 8
            for (int sl = sg; sl < S.nslabs; sl += 32) part += *reinterpret_cast<const f4*>(base + (size_t)sl * S.slab_stride);
        }
        red[sg][pi] = part;
        __syncthreads();
        if (sg != 0 || i >= S.n4) return;
        f4 sum = reinterpret_cast<const f4*>(S.slab)[i];          // + what earlier calls accumulated
#pragma unroll
        for (int k = 0; k < 32; ++k) sum += red[k][pi];
        reinterpret_cast<f4*>(S.buf)[i] = sum;
        reinterpret_cast<f4*>(S.slab)[i] = sum;                 // the slab stays what nsk_grid/decoder gradient downloads and a second pack expect
        return;
    }
    const int t = (blockIdx.x - b0) * 256 + threadIdx.x;
    if (t >= S.n4) return;
    const size_t at = S.idx ? (size_t)S.idx[t >> 3] * 8 + (t & 7) : (size_t)t;
    f4* sl = reinterpret_cast<f4*>(S.slab) + at;
    f4* bf = reinterpret_cast<f4*>(S.buf) + t;
    if (A.gather) *bf = *sl; else *sl = *bf;
}

struct PackSeg { float* img; const int* idx; const float* P; int n; int blk_end; };
struct PackArgs { PackSeg s[8]; int n; };
__global__ void k_pack_multi(PackArgs A)
{
    int r = 0;
    while (r < A.n - 1 && (int)blockIdx.x >= A.s[r].blk_end) ++r;
    const PackSeg& S = A.s[r];
    const int b0 = r == 0 ? 0 : A.s[r - 1].blk_end;
    const int i = (blockIdx.x - b0) * blockDim.x + threadIdx.x;
    if (i < S.n) { int k = S.idx[i]; S.img[i] = k >= 0 ? S.P[k] : 0.f; }
}

__global__ void k_adam_scalar(int n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                              float* __restrict__ v, float step_size, float bc2s, float b1, float b2, float eps)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float gg = g[i];
    float mm = b1 * m[i] + (1.f - b1) * gg;
    float vv = b2 * v[i] + (1.f - b2) * gg * gg;
    p[i] -= step_size * (mm / (sqrtf(vv) / bc2s + eps));
    m[i] = mm; v[i] = vv;
}

__global__ void k_sum(int n, const float* __restrict__ x, float* __restrict__ out)
{
    __shared__ float sh[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += x[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { for (int w = 1; w < (int)(blockDim.x >> 6); ++w) s += sh[w]; *out = s; }
}

// Mapper loss, reference src/Mapper.cpp:435-442
__global__ void k_loss_map(int N, const float* depth, const float* rgb, const float* gt_d, const float* gt_c, float w_color,
                           int use_color, float* g_depth, float* g_rgb, float* ray_loss)
{
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float l = 0.f, g = 0.f;
    if (gt_d[n] > 0.f) { float r = gt_d[n] - depth[n]; l += fabsf(r); g = -sgnf(r); }
    g_depth[n] = g;
    for (int k = 0; k < 3; ++k) {
        float r = gt_c[3 * n + k] - rgb[3 * n + k];
        if (use_color) { l += w_color * fabsf(r); g_rgb[3 * n + k] = -w_color * sgnf(r); }
        else g_rgb[3 * n + k] = 0.f;
    }
    ray_loss[n] = l;
}

// Tracker loss, reference src/Tracker.cpp:67-82
__global__ void k_loss_track(int N, const float* depth, const float* rgb, const float* var, const float* gt_d, const float* gt_c,
                             float w_color, int use_color, int handle_dynamic, int detach_var, const float* thr,
                             float* g_depth, float* g_rgb, float* g_var, float* ray_loss)
{
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float r = gt_d[n] - depth[n];
    bool mk = gt_d[n] > 0.f && (!handle_dynamic || fabsf(r) < *thr);
    float l = 0.f, gd = 0.f, gv = 0.f, gc[3] = {0.f, 0.f, 0.f};
    if (mk) {
        float u = sqrtf(var[n] + 1e-10f);
        l = fabsf(r) / u; gd = -sgnf(r) / u;
        if (!detach_var) gv = -fabsf(r) / (2.f * u * u * u);
        if (use_color) for (int k = 0; k < 3; ++k) { float rc = gt_c[3 * n + k] - rgb[3 * n + k]; l += w_color * fabsf(rc); gc[k] = -w_color * sgnf(rc); }
    }
    g_depth[n] = gd; if (g_var) g_var[n] = gv;
    for (int k = 0; k < 3; ++k) g_rgb[3 * n + k] = gc[k];
    ray_loss[n] = l;
}

// 10 * torch.median(|gt - depth|) (lower median) by a single-workgroup bitonic sort in LDS
__global__ __launch_bounds__(1024) void k_median_thr(int N, int P2, const float* gt_d, const float* depth, const uint8_t* keep, float* thr)
{
    extern __shared__ float shm[];
    __shared__ int nvalid;
    if (threadIdx.x == 0) nvalid = 0;
    __syncthreads();
    int cnt = 0;
    for (int i = threadIdx.x; i < P2; i += 1024) {
        const bool ok = i < N && (!keep || keep[i]);
        shm[i] = ok ? fabsf(gt_d[i] - depth[i]) : NSK_INF;       // masked rays sort to the end
        cnt += ok ? 1 : 0;
    }
    if (cnt) atomicAdd(&nvalid, cnt);
    __syncthreads();
    N = max(nvalid, 1);
    for (int k = 2; k <= P2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < P2; i += 1024) {
                int ixj = i ^ j;
                if (ixj > i) {
                    float a = shm[i], b = shm[ixj];
                    bool up = (i & k) == 0;
                    if ((a > b) == up) { shm[i] = b; shm[ixj] = a; }
                }
            }
            __syncthreads();
        }
    if (threadIdx.x == 0) *thr = 10.f * shm[(N - 1) / 2];
}

// raySampler direction part, reference include/torchlib/utils.h:44-52
__global__ void k_rays_from_pixels(int n, const int* pi, const int* pj, float fx, float fy, float cx, float cy, const float* c2w,
                                   int mode, float* ro, float* rd)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    float i = (float)pi[r], j = (float)pj[r];
    float d0 = div_rn(sub_rn(i, cx), fx);
    float d1 = (mode & 1) ? div_rn(sub_rn(i, cy), fy) : -div_rn(sub_rn(j, cy), fy);
    float d2 = -1.f;
    for (int a = 0; a < 3; ++a) {
        rd[3 * r + a] = add_rn(add_rn(mul_rn(d0, c2w[4 * a]), mul_rn(d1, c2w[4 * a + 1])), mul_rn(d2, c2w[4 * a + 2]));
        ro[3 * r + a] = c2w[4 * a + 3];
    }
}

__global__ __launch_bounds__(256) void k_rays_backward(int n, const int* pi, const int* pj, float fx, float fy, float cx, float cy,
                                                       int mode, const float* g_ro, const float* g_rd, float* g_c2w)
{
    __shared__ float sh[4][12];
    float acc[12];
    for (int k = 0; k < 12; ++k) acc[k] = 0.f;
    for (int r = threadIdx.x; r < n; r += 256) {
        float i = (float)pi[r], j = (float)pj[r];
        float dir[3] = {(i - cx) / fx, (mode & 1) ? (i - cy) / fy : -(j - cy) / fy, -1.f};
        for (int a = 0; a < 3; ++a) {
            for (int b = 0; b < 3; ++b) acc[4 * a + b] += g_rd[3 * r + a] * dir[b];
            acc[4 * a + 3] += g_ro[3 * r + a];
        }
    }
    for (int k = 0; k < 12; ++k) { float s = wave_sum(acc[k]); if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][k] = s; }
    __syncthreads();
    if (threadIdx.x < 12) g_c2w[threadIdx.x] = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}

// quad2rotation / get_camera_from_tensor, reference include/torchlib/utils.h:174-210
__device__ __forceinline__ void camera_matrix(const float* cam, float* c2w)
{
    // every product and sum rounded on its own, as the reference's tensor ops do.  The pragma matters: hipcc contracts a * b + c into an
    // FMA across statements by default (and HIP's __fmul_rn / __fadd_rn are plain operators), so without it the matrix depended on the
    // kernel this is inlined into -- k_rays_from_camera and k_prepare_rays differed in the last bit.
#pragma clang fp contract(off)
    const float qr = cam[0], qi = cam[1], qj = cam[2], qk = cam[3];
    const float rr = qr * qr, ii = qi * qi, jj = qj * qj, kk = qk * qk;
    const float ssum = ((rr + ii) + jj) + kk;
    const float two_s = 2.f / ssum;
    const float ij = qi * qj, ik = qi * qk, jk = qj * qk, ir = qi * qr, jr = qj * qr, kr = qk * qr;
    const float s0 = jj + kk, s1 = ii + kk, s2 = ii + jj;
    const float a01 = ij - kr, a02 = ik + jr, a10 = ij + kr, a12 = jk - ir, a20 = ik - jr, a21 = jk + ir;
    const float t0 = two_s * s0, t1 = two_s * s1, t2 = two_s * s2;
    float R[9] = {1.f - t0, two_s * a01, two_s * a02, two_s * a10, 1.f - t1, two_s * a12, two_s * a20, two_s * a21, 1.f - t2};
    for (int a = 0; a < 3; ++a) { for (int b = 0; b < 3; ++b) c2w[4 * a + b] = R[3 * a + b]; c2w[4 * a + 3] = cam[4 + a]; }
}
__global__ void k_camera_from_tensor(const float* cam, float* c2w) { camera_matrix(cam, c2w); }

__device__ __forceinline__ void camera_matrix_backward(const float* cam, const float* g_c2w, float* g_cam)
{
    float q[4] = {cam[0], cam[1], cam[2], cam[3]};
    float n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    float two_s = 2.f / n2;
    float qr = q[0], qi = q[1], qj = q[2], qk = q[3];
    float M[9] = {-(qj * qj + qk * qk), qi * qj - qk * qr, qi * qk + qj * qr, qi * qj + qk * qr, -(qi * qi + qk * qk),
                  qj * qk - qi * qr, qi * qk - qj * qr, qj * qk + qi * qr, -(qi * qi + qj * qj)};
    float dM[9][4] = {{0, 0, -2 * qj, -2 * qk}, {-qk, qj, qi, -qr}, {qj, qk, qr, qi}, {qk, qj, qi, qr}, {0, -2 * qi, 0, -2 * qk},
                      {-qi, -qr, qk, qj}, {-qj, qk, -qr, qi}, {qi, qr, qk, qj}, {0, -2 * qi, -2 * qj, 0}};
    for (int c = 0; c < 4; ++c) {
        float dts = -2.f * two_s * q[c] / n2;
        float s = 0.f;
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) s += g_c2w[4 * a + b] * (dts * M[3 * a + b] + two_s * dM[3 * a + b][c]);
        g_cam[c] = s;
    }
    for (int a = 0; a < 3; ++a) g_cam[4 + a] = g_c2w[4 * a + 3];
}
__global__ void k_camera_backward(const float* cam, const float* g_c2w, float* g_cam) { camera_matrix_backward(cam, g_c2w, g_cam); }

// Fused forms for a device-resident Tracker iteration (each removes launches of ~6 us from a ~70 us, launch-bound iteration):
// pose 7-vector -> c2w -> rays in one kernel; ray gradients -> d c2w -> d pose -> Adam on the pose in one single-block kernel.
__global__ void k_rays_from_camera(int n, const int* pi, const int* pj, float fx, float fy, float cx, float cy, const float* cam, int mode,
                                   float* ro, float* rd, float* c2w_out)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    float c2w[12];
    camera_matrix(cam, c2w);
    if (r == 0 && c2w_out) for (int k = 0; k < 12; ++k) c2w_out[k] = c2w[k];
    if (r >= n) return;
    float i = (float)pi[r], j = (float)pj[r];
    float d0 = div_rn(sub_rn(i, cx), fx);
    float d1 = (mode & 1) ? div_rn(sub_rn(i, cy), fy) : -div_rn(sub_rn(j, cy), fy);
    float d2 = -1.f;
    for (int a = 0; a < 3; ++a) {
        rd[3 * r + a] = add_rn(add_rn(mul_rn(d0, c2w[4 * a]), mul_rn(d1, c2w[4 * a + 1])), mul_rn(d2, c2w[4 * a + 2]));
        ro[3 * r + a] = c2w[4 * a + 3];
    }
}
__global__ __launch_bounds__(256) void k_pose_step(int n, const int* pi, const int* pj, float fx, float fy, float cx, float cy, int mode,
                                                   const float* g_ro, const float* g_rd, float* cam, float* m, float* v, float step_size,
                                                   float bc2s, float b1, float b2, float eps, float* g_cam_out)
{
    __shared__ float sh[4][12];
    __shared__ float g_c2w[12];
    float acc[12];
    for (int k = 0; k < 12; ++k) acc[k] = 0.f;
    for (int r = threadIdx.x; r < n; r += 256) {
        float i = (float)pi[r], j = (float)pj[r];
        float dir[3] = {(i - cx) / fx, (mode & 1) ? (i - cy) / fy : -(j - cy) / fy, -1.f};
        for (int a = 0; a < 3; ++a) {
            for (int b = 0; b < 3; ++b) acc[4 * a + b] += g_rd[3 * r + a] * dir[b];
            acc[4 * a + 3] += g_ro[3 * r + a];
        }
    }
    for (int k = 0; k < 12; ++k) { float s = wave_sum(acc[k]); if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][k] = s; }
    __syncthreads();
    if (threadIdx.x < 12) g_c2w[threadIdx.x] = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        float g[7];
        camera_matrix_backward(cam, g_c2w, g);
        for (int k = 0; k < 7; ++k) {              // k_adam_scalar's update
            float gg = g[k];
            float mm = b1 * m[k] + (1.f - b1) * gg;
            float vv = b2 * v[k] + (1.f - b2) * gg * gg;
            cam[k] -= step_size * (mm / (sqrtf(vv) / bc2s + eps));
            m[k] = mm; v[k] = vv;
            if (g_cam_out) g_cam_out[k] = gg;
        }
    }
}

// The pose kernels of a whole window in ONE launch (bundle adjustment, reference src/Mapper.cpp:305-329,467-489): block f reduces the ray
// gradients of frame f's rays [first, first + count) of the batch to d loss / d c2w, chains through quad2rotation to the 7-vector
// and either steps the pose (adam = 1: k_pose_step's arithmetic, frame by frame) or only stores the gradient (adam = 0: N > 1, the
// gradients of all frames are summed over the ranks first, then nsk_adam_vector steps every pose at once).  Inactive frames (not optimised,
// or no ray of theirs in this shard) get a zero gradient.  One more block counts the shard's kept rays into g_out[8 * nframes + 1].
struct PoseFrames { int first[NSK_MAX_POSE_FRAMES], count[NSK_MAX_POSE_FRAMES]; unsigned char active[NSK_MAX_POSE_FRAMES]; };
__global__ __launch_bounds__(256) void k_pose_multi(PoseFrames F, int nframes, const int* pi, const int* pj, float fx, float fy, float cx, float cy, int mode,
                                                    const float* g_ro, const float* g_rd, float* cams, float* ms, float* vs, float step_size, float bc2s,
                                                    float b1, float b2, float eps, int adam, float* g_out, const uint8_t* keep, int n_keep)
{
    __shared__ float sh[4][12];
    __shared__ float g_c2w[12];
    const int f = blockIdx.x;
    if (f == nframes) {                                  // the count of the rays that take part (keep == nullptr: all of them)
        float cnt = 0.f;
        for (int r = threadIdx.x; r < n_keep; r += 256) cnt += (!keep || keep[r]) ? 1.f : 0.f;
        cnt = wave_sum(cnt);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][0] = cnt;
        __syncthreads();
        if (threadIdx.x == 0 && g_out) g_out[8 * nframes + 1] = sh[0][0] + sh[1][0] + sh[2][0] + sh[3][0];
        return;
    }
    const int n = F.active[f] ? F.count[f] : 0, r0 = F.first[f];
    float acc[12];
    for (int k = 0; k < 12; ++k) acc[k] = 0.f;
    for (int r = r0 + threadIdx.x; r < r0 + n; r += 256) {
        float i = (float)pi[r], j = (float)pj[r];
        float dir[3] = {(i - cx) / fx, (mode & 1) ? (i - cy) / fy : -(j - cy) / fy, -1.f};
        for (int a = 0; a < 3; ++a) {
            for (int b = 0; b < 3; ++b) acc[4 * a + b] += g_rd[3 * r + a] * dir[b];
            acc[4 * a + 3] += g_ro[3 * r + a];
        }
    }
    for (int k = 0; k < 12; ++k) { float s = wave_sum(acc[k]); if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][k] = s; }
    __syncthreads();
    if (threadIdx.x < 12) g_c2w[threadIdx.x] = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        float* cam = cams + 8 * f;
        float g[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (n > 0) camera_matrix_backward(cam, g_c2w, g);
        if (g_out) for (int k = 0; k < 8; ++k) g_out[8 * f + k] = g[k];
        if (adam && F.active[f]) {
            float* m = ms + 8 * f; float* v = vs + 8 * f;
            for (int k = 0; k < 7; ++k) {              // k_adam_scalar's update
                float gg = g[k];
                float mm = b1 * m[k] + (1.f - b1) * gg;
                float vv = b2 * v[k] + (1.f - b2) * gg * gg;
                cam[k] -= step_size * (mm / (sqrtf(vv) / bc2s + eps));
                m[k] = mm; v[k] = vv;
            }
        }
    }
}

__global__ void k_inside_filter(RParams R, int N, const float* ro, const float* rd, const float* gt, uint8_t* keep)
{
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    keep[n] = ray_box_far(R.bound, ro[3 * n], ro[3 * n + 1], ro[3 * n + 2], rd[3 * n], rd[3 * n + 1], rd[3 * n + 2]) >= gt[n];
}

// the whole ray preparation of an iteration in one launch (nsk_prepare_rays): the bodies of k_sample_pixels, k_gather_pixels,
// k_rays_from_pixels / k_rays_from_camera and k_inside_filter, one thread per ray, frames from a table in the kernel arguments
struct PrepFrame { const float* depth; const float* color; const float* pose; int cam7; unsigned long long seed; };
struct PrepArgs {
    PrepFrame f[16];
    int nframes, per, H0, W0, Ww, W, mode;
    unsigned long long total;
    float fx, fy, cx, cy;
    int* pi; int* pj; float* gd; float* gc; float* ro; float* rd; uint8_t* keep;
    RParams R;
};
__global__ __launch_bounds__(256) void k_prepare_rays(PrepArgs A)
{
    const int fi = blockIdx.y, r = blockIdx.x * blockDim.x + threadIdx.x;      // one frame per block row: the table index is wave-uniform
    if (r >= A.per) return;
    const int t = fi * A.per + r;
    const PrepFrame& F = A.f[fi];
    const unsigned long long ind = ((unsigned long long)hash_u32(F.seed, (uint32_t)r, 0x51u) * A.total) >> 32;      // k_sample_pixels
    const int pi = A.W0 + (int)(ind % (unsigned long long)A.Ww), pj = A.H0 + (int)(ind / (unsigned long long)A.Ww);
    A.pi[t] = pi; A.pj[t] = pj;
    const size_t p = (size_t)pj * A.W + pi;                                                                         // k_gather_pixels
    const float gd = F.depth[p];
    A.gd[t] = gd;
    if (F.color && A.gc) { A.gc[3 * t] = F.color[3 * p]; A.gc[3 * t + 1] = F.color[3 * p + 1]; A.gc[3 * t + 2] = F.color[3 * p + 2]; }
    float c2w[12];
    if (F.cam7) camera_matrix(F.pose, c2w);                                                                          // k_rays_from_camera
    else for (int k = 0; k < 12; ++k) c2w[k] = F.pose[k];                                                            // k_rays_from_pixels
    const float i = (float)pi, j = (float)pj;
    const float d0 = div_rn(sub_rn(i, A.cx), A.fx);
    const float d1 = (A.mode & 1) ? div_rn(sub_rn(i, A.cy), A.fy) : -div_rn(sub_rn(j, A.cy), A.fy);
    const float d2 = -1.f;
    float o[3], d[3];
    for (int a = 0; a < 3; ++a) {
        d[a] = add_rn(add_rn(mul_rn(d0, c2w[4 * a]), mul_rn(d1, c2w[4 * a + 1])), mul_rn(d2, c2w[4 * a + 2]));
        o[a] = c2w[4 * a + 3];
        A.rd[3 * t + a] = d[a]; A.ro[3 * t + a] = o[a];
    }
    if (A.keep) A.keep[t] = ray_box_far(A.R.bound, o[0], o[1], o[2], d[0], d[1], d[2]) >= gd;                        // k_inside_filter
}

// in-bound override for eval_points (reference src/Renderer.cpp:26-36)
__global__ void k_eval_finish(int M, int stage, const float* pts, const float* bound6, const float* occ_a, const float* occ_b,
                              const float* rgb4, float* raw)
{
    int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    float px = pts[3 * m], py = pts[3 * m + 1], pz = pts[3 * m + 2];
    bool inb = px < bound6[1] && px > bound6[0] && py < bound6[3] && py > bound6[2] && pz < bound6[5] && pz > bound6[4];
    float occ = occ_a[m];
    if (occ_b) occ = occ_b[m] + occ;
    f4 c = (f4)(0.f);
    if (rgb4) c = *reinterpret_cast<const f4*>(rgb4 + (size_t)m * 4);
    *reinterpret_cast<f4*>(raw + (size_t)m * 4) = (f4){c[0], c[1], c[2], inb ? occ : 100.f};
}

// raw2outputs_nerf_color as a stand-alone op (reference include/torchlib/utils.h:148-172): one wave per ray
__global__ __launch_bounds__(256) void k_raw2outputs(int N, int S, int occupancy, const float* __restrict__ raw, const float* __restrict__ zv,
                                                     const float* __restrict__ rays_d, float* rgb, float* depth, float* var, float* weights)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + wave;
    if (n >= N) return;
    const bool act = lane < S;
    const size_t m = (size_t)n * S + (act ? lane : S - 1);
    const float dx = rays_d[3 * n], dy = rays_d[3 * n + 1], dz = rays_d[3 * n + 2];
    const float z = zv[m];
    const f4 r4 = *reinterpret_cast<const f4*>(raw + m * 4);
    const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
    const float znext = __shfl_down(z, 1);
    const float dist = ((lane + 1 < S) ? (znext - z) : 1e10f) * nrm;
    float alpha = occupancy ? 1.f / (1.f + expf(-10.f * r4[3])) : 1.f - expf(-fmaxf(r4[3], 0.f) * dist);
    if (!act) alpha = 0.f;
    float incl = act ? (1.f - alpha + 1e-10f) : 1.f;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { float up = __shfl_up(incl, o); if (lane >= o) incl *= up; }
    float T = __shfl_up(incl, 1);
    if (lane == 0) T = 1.f;
    const float w = act ? alpha * T : 0.f;
    const float D = wave_sum(w * z);
    const float cr = wave_sum(w * r4[0]), cg = wave_sum(w * r4[1]), cb = wave_sum(w * r4[2]);
    const float V = wave_sum(w * (z - D) * (z - D));
    if (weights && act) weights[m] = w;
    if (lane == 0) { depth[n] = D; var[n] = V; rgb[3 * n] = cr; rgb[3 * n + 1] = cg; rgb[3 * n + 2] = cb; }
}

// Frustum voxel mask (next row N2): Mapper::get_mask_from_c2w, reference src/Mapper.cpp:42-130 (intended semantics, D21/D22):
// pass 1 projects every voxel centre into the frame, samples the depth image bilinearly (cv::remap INTER_LINEAR, zero border)
// and takes the maximum sampled depth; pass 2 applies the in-image / depth / near-camera tests.
__device__ __forceinline__ void voxel_point(const float* b, int Z, int Y, int X, int v, float (&p)[3])
{
    int ix = v % X, iy = (v / X) % Y, iz = v / (X * Y);
    p[0] = add_rn(b[0], mul_rn(sub_rn(b[1], b[0]), linspace01(ix, X)));
    p[1] = add_rn(b[2], mul_rn(sub_rn(b[3], b[2]), linspace01(iy, Y)));
    p[2] = add_rn(b[4], mul_rn(sub_rn(b[5], b[4]), linspace01(iz, Z)));
}
struct FrustumArgs { float bound[6]; float w2c[16]; float cam[3]; int Z, Y, X, H, W; float fx, fy, cx, cy; };
__global__ void k_frustum_pass1(FrustumArgs A, const float* __restrict__ img, float* __restrict__ dep, float* __restrict__ zz,
                                uint8_t* __restrict__ inimg, unsigned int* __restrict__ dmax_bits)
{
    const int n = A.Z * A.Y * A.X;
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    float d = 0.f;
    if (v < n) {
        float p[3], cam[3];
        voxel_point(A.bound, A.Z, A.Y, A.X, v, p);
        for (int a = 0; a < 3; ++a)
            cam[a] = add_rn(add_rn(add_rn(mul_rn(A.w2c[4 * a], p[0]), mul_rn(A.w2c[4 * a + 1], p[1])), mul_rn(A.w2c[4 * a + 2], p[2])), A.w2c[4 * a + 3]);
        cam[0] = -cam[0];
        const float z = add_rn(cam[2], 1e-5f);
        const float u = div_rn(add_rn(mul_rn(A.fx, cam[0]), mul_rn(A.cx, cam[2])), z);
        const float w = div_rn(add_rn(mul_rn(A.fy, cam[1]), mul_rn(A.cy, cam[2])), z);
        const float x0 = floorf(u), y0 = floorf(w);
        const float ax = sub_rn(u, x0), ay = sub_rn(w, y0);
        const int ix = (int)x0, iy = (int)y0;
        float s = 0.f;
        for (int dy = 0; dy < 2; ++dy) for (int dx = 0; dx < 2; ++dx) {
            int x = ix + dx, y = iy + dy;
            if (x < 0 || x >= A.W || y < 0 || y >= A.H) continue;
            s = add_rn(s, mul_rn(mul_rn(dx ? ax : sub_rn(1.f, ax), dy ? ay : sub_rn(1.f, ay)), img[(size_t)y * A.W + x]));
        }
        d = s;
        dep[v] = d; zz[v] = z;
        inimg[v] = (u < (float)A.W && u > 0.f && w < (float)A.H && w > 0.f) ? 1 : 0;
    }
    float m = wave_max(fmaxf(d, 0.f));
    if ((threadIdx.x & 63) == 0) atomicMax(dmax_bits, __float_as_uint(m));     // non-negative floats order like their bit patterns
}
__global__ void k_frustum_pass2(FrustumArgs A, const float* __restrict__ dep, const float* __restrict__ zz, const uint8_t* __restrict__ inimg,
                                const unsigned int* __restrict__ dmax_bits, uint8_t* __restrict__ mask)
{
    const int n = A.Z * A.Y * A.X;
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const float dmax = __uint_as_float(*dmax_bits);
    const float d = dep[v] == 0.f ? dmax : dep[v];
    const float nz = -zz[v];
    int m = inimg[v] && (0.f <= nz) && (nz <= add_rn(d, 0.5f));
    float p[3];
    voxel_point(A.bound, A.Z, A.Y, A.X, v, p);
    const float dx = sub_rn(p[0], A.cam[0]), dy = sub_rn(p[1], A.cam[1]), dz = sub_rn(p[2], A.cam[2]);
    if (add_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)), mul_rn(dz, dz)) < 0.25f) m = 1;
    mask[v] = (uint8_t)m;
}

// Keyframe overlap (next row N3): Mapper::keyframe_selection_overlap, reference src/Mapper.cpp:132-196.  One workgroup per
// keyframe projects the N x ns sample points of the current frame into that keyframe and counts those inside its image.
struct OverlapArgs { int N, ns, H, W, K; float fx, fy, cx, cy; };
__global__ __launch_bounds__(256) void k_keyframe_overlap(OverlapArgs A, const float* __restrict__ ro, const float* __restrict__ rd,
                                                           const float* __restrict__ gd, const float* __restrict__ w2c_all,
                                                           float* __restrict__ percent)
{
    __shared__ int sh[4];
    const float* w2c = w2c_all + 16 * blockIdx.x;
    const int total = A.N * A.ns;
    int count = 0;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        const int n = i / A.ns, s = i - n * A.ns;
        const float nearv = mul_rn(gd[n], 0.8f), farv = add_rn(gd[n], 0.5f);
        const float t = linspace01(s, A.ns);
        const float z = add_rn(mul_rn(nearv, sub_rn(1.f, t)), mul_rn(farv, t));
        float p[3], cam[3];
        for (int a = 0; a < 3; ++a) p[a] = add_rn(ro[3 * n + a], mul_rn(rd[3 * n + a], z));
        for (int a = 0; a < 3; ++a)
            cam[a] = add_rn(add_rn(add_rn(mul_rn(w2c[4 * a], p[0]), mul_rn(w2c[4 * a + 1], p[1])), mul_rn(w2c[4 * a + 2], p[2])), w2c[4 * a + 3]);
        cam[0] = -cam[0];
        const float zc = add_rn(cam[2], 1e-5f);
        const float u = div_rn(add_rn(mul_rn(A.fx, cam[0]), mul_rn(A.cx, cam[2])), zc);
        const float v = div_rn(add_rn(mul_rn(A.fy, cam[1]), mul_rn(A.cy, cam[2])), zc);
        if (u < (float)(A.W - 20) && u > 20.f && v < (float)(A.H - 20) && v > 20.f && zc < 0.f) ++count;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) count += __shfl_xor(count, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = count;
    __syncthreads();
    if (threadIdx.x == 0) percent[blockIdx.x] = div_rn((float)(sh[0] + sh[1] + sh[2] + sh[3]), (float)total);
}

// Pixel draw and ground-truth gather of raySampler (next row N1): reference include/torchlib/utils.h:13-43.  torch::randint's
// stream cannot be reproduced; the draw uses a counter-based hash of (seed, ray index), the same the perturbation of k_sample uses.
__global__ void k_sample_pixels(unsigned long long seed, int n, int H0, int W0, int Ww, unsigned long long total, int* __restrict__ pi,
                                int* __restrict__ pj)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const unsigned long long ind = ((unsigned long long)hash_u32(seed, (uint32_t)r, 0x51u) * total) >> 32;
    pi[r] = W0 + (int)(ind % (unsigned long long)Ww);
    pj[r] = H0 + (int)(ind / (unsigned long long)Ww);
}
__global__ void k_gather_pixels(int n, const int* __restrict__ pi, const int* __restrict__ pj, int W, const float* __restrict__ depth,
                                const float* __restrict__ color, float* __restrict__ gd, float* __restrict__ gc)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const size_t p = (size_t)pj[r] * W + pi[r];
    gd[r] = depth[p];
    if (color && gc) { gc[3 * r] = color[3 * p]; gc[3 * r + 1] = color[3 * p + 1]; gc[3 * r + 2] = color[3 * p + 2]; }
}

// ---------------------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------------------
struct GridState {
    int C = 0, Z = 0, Y = 0, X = 0;
    size_t n = 0;                // floats = nvox*32
    float* v = nullptr; float* m = nullptr; float* s = nullptr;
    uint8_t* mask = nullptr;
    size_t g_off = 0;            // offset in the gradient slab
    int* midx = nullptr; int nmask = 0; bool midx_dirty = true;      // ascending list of the marked voxels (packed gradient exchange)
};
struct DecState {
    int n = 0;
    float* p = nullptr; float* m = nullptr; float* s = nullptr;
    float* fimg = nullptr; float* bimg = nullptr;
    int* fidx = nullptr; int* bidx = nullptr;
    int* finv = nullptr; int* binv = nullptr;     // inverse of fidx / bidx: image position of each canonical parameter
    int* inv16 = nullptr;                         // inverse of fidx16
    int* invh = nullptr;                          // inverse of bidx16 (fp16 backward image)
    float* bimg16 = nullptr; int* bidx16 = nullptr; int bfrag16_n = 0; bool bimg16_dirty = true;   // bf16 3-piece backward image (frozen chain), repacked lazily
    float* fimg16 = nullptr; int* fidx16 = nullptr; int frag16_n = 0, fimg16_f = 0, tail_off = 0, tail16_off = 0, np16 = 3, cq16 = 2;   // forward image in pieces: 3 bf16 (mode 1) or 2 fp16 (mode 2)
    int fimg_n = 0, bimg_n = 0;
    int trainable = 0;
    bool loaded = false;
    size_t g_off = 0;
};
struct Workspace {
    int capM = 0, capN = 0;
    float* z = nullptr; float* occ[3] = {nullptr, nullptr, nullptr}; float* rgb4 = nullptr;
    unsigned long long* masks[4] = {nullptr, nullptr, nullptr, nullptr};
    f4* hsave[4] = {nullptr, nullptr, nullptr, nullptr};     // block outputs of trainable decoders saved by the forward (save_h)
    size_t hcap[4] = {0, 0, 0, 0};                              // capacity in tiles
    int hsave_M[4] = {0, 0, 0, 0};                              // sample count of the forward that filled hsave (0 = stale)
    float* g_raw = nullptr; float* ray_loss = nullptr;
    float* tmp_rgb = nullptr; float* tmp_depth = nullptr; float* tmp_var = nullptr;
    float* dec_slabs = nullptr;      // per-workgroup partial decoder gradients [num_cu][20920]
    // cell sort of the samples (k_sample keys -> k_sort_scan -> k_sort_place): perm lists the samples cell by cell
    int* perm = nullptr; int* skey = nullptr; int* srank = nullptr;     // [capM]
    int* hist = nullptr; int* offs = nullptr; size_t hist_cap = 0;       // [bins of the key level]; hist is zero between steps
    // second set of the sampling outputs: nsk_map_prepare fills it on the side stream while the current step runs; a step that finds its
    // batch prepared swaps the pointers above with these
    float* z_alt = nullptr; int* perm_alt = nullptr; int* skey_alt = nullptr; int* srank_alt = nullptr; int* offs_alt = nullptr;
    int alt_capM = 0; size_t alt_bins = 0;
};
struct nsk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cu = 256;
    RParams R;
    float* d_bound = nullptr;
    GridState grid[4];
    DecState dec[4];
    float* slab = nullptr; size_t slab_n = 0;
    float* xbuf = nullptr; size_t xbuf_cap = 0; size_t xbuf_n = 0;     // packed exchange buffer (nsk_grad_pack) and its current length
    bool x_identity = false;                                              // the last pack handed out the slab itself (no masks)
    bool xlevels[4] = {false, false, false, false}; bool xdecs[4] = {false, false, false, false};      // what the last pack holds
    Workspace ws;
    float* scal = nullptr;       // [0] gt max, [1] median threshold, [2] loss, [4] frustum max depth (bits)
    void* fr_tmp = nullptr; size_t fr_cap = 0;      // nsk_frustum_mask scratch
    int adam_step[NSK_NUM_GROUPS] = {0, 0, 0, 0, 0, 0};
    // ---- hipGraph capture of a step (nsk_graph_*): the kernels of the calls made between begin and end are recorded once and
    // replayed with one launch; Adam's bias-correction constants are kernel arguments, so the recorded k_adam_multi node is
    // patched with the current step counts before every replay
    bool capturing = false;
    struct CapAdam { AdamArgs args; int group[8]; float lr[8]; };
    std::vector<CapAdam> cap_adams;             // Adam launches seen during the current capture
    struct CapVec { int n; float* p; const float* g; float* m; float* v; float lr, b1, b2, eps; int step; hipGraphNode_t node; float ss, bc2s; };
    std::vector<CapVec> cap_vecs;               // nsk_adam_vector launches seen during the current capture
    int cap_rollback[NSK_NUM_GROUPS] = {0, 0, 0, 0, 0, 0};
    int ws_flip = 0;                            // parity of the swaps between the workspace's two sampling-output sets (forward_core): a graph records the set that was primary at capture
    struct GraphRec { bool stale = false; int flip = 0; hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; hipGraphNode_t adam_node = nullptr; bool has_adam = false; CapAdam adam; std::vector<CapVec> vecs; };
    std::vector<GraphRec> graphs;
    bool touched[NSK_NUM_GROUPS] = {false, false, false, false, false, false};
    bool deterministic = false;             // debug: bit-reproducible gradients (see nsk_set_tuning in include/nsk.h)
    bool roctx = false;                     // roctx ranges around every profiled launch group (libroctx64, loaded on demand)
    int tune_fwd_occ_cost = 0, tune_no_occ_role = 0;         // the merged middle + fine role of the forward: its cost against the colour role's (0 = 460); 1 = never merged, 2 = always
    int tune_fwd_fine_cost = 0, tune_fwd_color_cost = 0;     // experiments: forward role costs (nsk_set_tuning "fwd_fine_cost" / "fwd_color_cost")
    int tune_skew = 0;                      // start offset of the upper four waves of a decoder workgroup, x 1024 cycles (wave_skew, nsk_device.h)
    bool median_fused_pending = false;
    int tune_no_deferred_median = 0;        // 1: the threshold inside the compositing launch (grid barrier) even where the backward could find it (round 3's form)
    int tune_no_fused_median = 0;           // 1: the Tracker's median threshold in its own launch even where the fused form applies (experiments, tests)
    int tune_frozen_cost = 0;               // > 0: overrides the frozen-role cost of the backward's workgroup split (nsk_set_tuning; experiments)
    int tune_frozen_cost_rays = 0;          // > 0: the same for launches with ray gradients (bundle adjustment: the frozen roles also carry g_e and d/dp)
    int tune_frozen_mid_pct = 100;          // the middle decoder's frozen tile against the fine one's, in percent (its level has 8x the samples per voxel: more same-line atomics)
    int tune_no_frozen_kernel = 0;          // 1: launches without a trainable role also go through k_decode_bwd_multi (experiments, tests)
    const uint8_t* ray_mask = nullptr;      // nsk_set_ray_mask
    // nsk_set_depth_max_batch: the batch whose max(gt_depth) the sampling uses when gt_depth_max < 0 (a ray shard of a larger batch: N > 1)
    struct DMax { const float* gt = nullptr; const uint8_t* keep = nullptr; int n = 0; } dmax;
    float* xextra = nullptr; size_t xextra_n = 0;      // nsk_grad_extra: a caller-owned vector that travels with the packed exchange
    int sort_mode = -1;                     // -1 automatic (sort_pays), 0 never, 1 always (nsk_set_sort_mode; tests)
    bool sorted = false;                    // the current step's decoder launches walk the samples in cell-sorted order (ws.perm)
    // nsk_map_prepare: the sampling (+ cell sort) of the NEXT batch rides in the launches of the current step (composite + sample, backward + scan,
    // Adam + place): `req` is a registered batch nothing has been launched for yet, `prep` the batch whose outputs sit (or are being built) in the
    // workspace's second set; done: bit 0 sampled, 1 offsets scanned, 2 placed
    struct Prep { bool valid = false; int stage = 0, N = 0, S = 0; const float* ro = nullptr; const float* rd = nullptr; const float* gt = nullptr;
                  float gtmax = 0.f; const uint8_t* mask = nullptr; bool sorted = false; int done = 0; RParams R; DMax dmax; } prep, req;
    int tune_no_piggyback = 0;              // 1: a prepared batch is sampled by launches of its own at the start of its step (experiments, tests)
    int pend_w = -1, pend_nb = 0;           // decoder whose per-workgroup gradient slabs are not yet summed into the slab (flush_pending)
    int dbg_M = 0, dbg_S = 0;               // sample count / samples per ray of the last forward_core (nsk_debug_relu_bits, nsk_debug_preact)
    int bwd_mode = 2;                       // decoder backward chains: 2 fp16 2-piece split (default), 0 fp32 MFMA (nsk_set_backward_mode: a measuring stick)
    int matmul_mode = 2;                    // decoder forward: 0 fp32 MFMA, 1 bf16 3-piece split, 2 fp16 2-piece split (nsk_bf16.h)
    double last_bytes = 0, last_flops = 0; int last_samples = 0;
    // optional per-kernel timing with HIP events on the context's stream (nsk_profile_begin / _end)
    bool prof = false;
    struct ProfRec { const char* name; hipEvent_t a, b; };
    std::vector<ProfRec> prof_recs;
};

// roctx ranges (rocprofv3 --marker-trace shows them around the kernels of each launch group); libroctx64 is loaded on first use
typedef int (*roctx_push_t)(const char*);
typedef int (*roctx_pop_t)(void);
static roctx_push_t g_roctx_push = nullptr;
static roctx_pop_t g_roctx_pop = nullptr;
static bool roctx_ready()
{
    static int state = 0;       // 0 untried, 1 ok, -1 unavailable
    if (state == 0) {
        void* h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
        if (h) { g_roctx_push = (roctx_push_t)dlsym(h, "roctxRangePushA"); g_roctx_pop = (roctx_pop_t)dlsym(h, "roctxRangePop"); }
        state = (g_roctx_push && g_roctx_pop) ? 1 : -1;
    }
    return state == 1;
}

struct ProfScope {      // records start/stop events around the launches issued while it is alive
    nsk_ctx* c; hipEvent_t a = nullptr, b = nullptr; const char* name; bool ranged = false;
    ProfScope(nsk_ctx* c_, const char* n, bool on = true) : c(c_), name(n)
    {
        if (!on) return;                                    // launches on the side stream are not timed (the events sit on c->stream)
        if (c->roctx && roctx_ready()) { g_roctx_push(n); ranged = true; }
        if (!c->prof) return;
        hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a, c->stream);
    }
    ~ProfScope()
    {
        if (ranged) g_roctx_pop();
        if (!c->prof || !a) return;
        hipEventRecord(b, c->stream);
        c->prof_recs.push_back({name, a, b});
    }
};

// Captured graphs hold raw pointers into the workspace, the gradient slab and the grids: whenever one of those buffers is
// freed or reallocated every recorded graph becomes unusable and nsk_graph_launch must say so instead of replaying onto freed memory.
static void invalidate_graphs(nsk_ctx* c)
{
    for (auto& R : c->graphs) {
        if (R.stale) continue;
        if (R.exec) { hipGraphExecDestroy(R.exec); R.exec = nullptr; }
        if (R.graph) { hipGraphDestroy(R.graph); R.graph = nullptr; }
        R.stale = true;
    }
}

extern "C" const char* nsk_last_error(void) { return g_err.c_str(); }
extern "C" int nsk_version(void) { return NSK_VERSION; }

static int which_ok(int w) { return w >= 0 && w < 4; }

template <typename K>
static int set_lds(K kern, size_t bytes)
{
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}

static size_t fwd_img_floats(int w) { return w == 0 ? CoarseFwdImg::TOTAL : (w == 2 ? MlpFwdImg<4>::TOTAL : MlpFwdImg<2>::TOTAL); }
static size_t bwd_img_floats(int w) { return w == 0 ? CoarseBwdImg::TOTAL : MlpBwdImg::TOTAL; }
static size_t bwd_lds_bytes(int w, bool train)
{
    size_t scratch = 8 * 3840;                               // 8 waves x (NSK_SCRATCH_FLOATS <= 960) floats
    if (!train) return bwd_img_floats(w) * 4 + scratch;
    if (w == 1 || w == 3) return PM_LDS_BYTES;               // merged-phase body (nsk_train.h): 18 fragment groups + tail + 224 panel rows
    size_t img = w == 2 ? 0 : bwd_img_floats(w);          // saved-activation form: only the backward image sits in LDS
    return (img + PN_FLOATS(w == 2 ? 4 : 2)) * 4;            // the per-wave scatter scratch lives in panel rows 0..63
}

extern "C" int nsk_ctx_create(int device, void* hip_stream, nsk_ctx** out)
{
    if (!out) return fail("nsk_ctx_create: out is NULL");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return fail("nsk_ctx_create: no HIP device (the MI355X kernels have no CPU fallback)");
    if (device < 0 || device >= ndev) return fail("nsk_ctx_create: device %d out of range (%d devices)", device, ndev);
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return fail("nsk_ctx_create: device arch %s, this library is built for gfx950 only", prop.gcnArchName);
    nsk_ctx* c = new nsk_ctx();
    c->device = device;
    c->num_cu = prop.multiProcessorCount;
    if (hip_stream) { c->stream = (hipStream_t)hip_stream; c->own_stream = false; }
    else { HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
    const float b[6] = {-4.5f, 3.82f, -1.5f, 2.02f, -3.0f, 2.76f};     // reference src/Renderer.cpp:15
    memcpy(c->R.bound, b, sizeof(b));
    c->R.n_samples = 32; c->R.n_surface = 16; c->R.lindisp = 0; c->R.occupancy = 0; c->R.perturb = 0.f; c->R.seed = 0;
    HIPCHK(hipMalloc(&c->d_bound, 6 * sizeof(float)));
    HIPCHK(hipMemcpy(c->d_bound, b, sizeof(b), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(&c->scal, 16 * sizeof(float)));
    HIPCHK(hipMemset(c->scal, 0, 16 * sizeof(float)));
    // dynamic LDS limits
    CHK(set_lds(k_decode_fwd<0>, fwd_img_floats(0) * 4)); CHK(set_lds(k_decode_fwd<1>, fwd_img_floats(1) * 4));
    CHK(set_lds(k_decode_fwd<2>, fwd_img_floats(2) * 4)); CHK(set_lds(k_decode_fwd<3>, fwd_img_floats(3) * 4));
#define SETB(W) \
    CHK(set_lds(k_decode_bwd<W, false>, bwd_lds_bytes(W, false))); CHK(set_lds(k_decode_bwd<W, true>, bwd_lds_bytes(W, false))); \
    CHK(set_lds(k_decode_bwd_train<W, false>, bwd_lds_bytes(W, true))); CHK(set_lds(k_decode_bwd_train<W, true>, bwd_lds_bytes(W, true)));
    SETB(0) SETB(1) SETB(2) SETB(3)
#undef SETB
    CHK(set_lds(k_median_thr, 16384 * 4));
    CHK(set_lds(k_decode_fwd_multi, 160 * 1024)); CHK(set_lds(k_decode_fwd_multi_bf16<8>, 160 * 1024)); CHK(set_lds(k_decode_fwd_multi_bf16<8, 2>, 160 * 1024)); CHK(set_lds(k_decode_fwd_multi_occ<8>, 160 * 1024));
    CHK(set_lds(k_decode_bwd_multi<false>, 160 * 1024 - 256)); CHK(set_lds(k_decode_bwd_multi<true>, 160 * 1024));      // (<false>, frozen: the scan role keeps a few words of static LDS)
    CHK(set_lds(k_decode_bwd_frozen<false>, 160 * 1024 - 256));
    CHK(set_lds(k_decode_bwd_track, 160 * 1024)); CHK(set_lds(k_decode_bwd_multi_full<false>, 160 * 1024)); CHK(set_lds(k_decode_bwd_multi_full<true>, 160 * 1024));
    CHK(set_lds(k_decode_fwd_dump<0>, 160 * 1024)); CHK(set_lds(k_decode_fwd_dump<1>, 160 * 1024)); CHK(set_lds(k_decode_fwd_dump<2>, 160 * 1024));
    *out = c;
    return 0;
}

static void free_ws(Workspace& w)
{
    hipFree(w.z); for (int i = 0; i < 3; ++i) hipFree(w.occ[i]); hipFree(w.rgb4);
    for (int i = 0; i < 4; ++i) hipFree(w.masks[i]);
    for (int i = 0; i < 4; ++i) { hipFree(w.hsave[i]); w.hsave[i] = nullptr; w.hcap[i] = 0; w.hsave_M[i] = 0; }
    hipFree(w.g_raw); hipFree(w.ray_loss); hipFree(w.tmp_rgb); hipFree(w.tmp_depth); hipFree(w.tmp_var); hipFree(w.dec_slabs);
    hipFree(w.perm); hipFree(w.skey); hipFree(w.srank); if (w.hist) hipFree(w.hist - 16); hipFree(w.offs);
    hipFree(w.z_alt); hipFree(w.perm_alt); hipFree(w.skey_alt); hipFree(w.srank_alt); hipFree(w.offs_alt);
    w = Workspace();
}

extern "C" int nsk_ctx_destroy(nsk_ctx* c)
{
    if (!c) return 0;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    for (int i = 0; i < 4; ++i) {
        hipFree(c->grid[i].v); hipFree(c->grid[i].m); hipFree(c->grid[i].s); hipFree(c->grid[i].mask); hipFree(c->grid[i].midx);
        hipFree(c->dec[i].p); hipFree(c->dec[i].m); hipFree(c->dec[i].s); hipFree(c->dec[i].fimg); hipFree(c->dec[i].bimg);
        hipFree(c->dec[i].fidx); hipFree(c->dec[i].bidx); hipFree(c->dec[i].finv); hipFree(c->dec[i].binv); hipFree(c->dec[i].inv16); hipFree(c->dec[i].invh); hipFree(c->dec[i].bimg16); hipFree(c->dec[i].bidx16);
        hipFree(c->dec[i].fimg16); hipFree(c->dec[i].fidx16);
    }
    hipFree(c->xbuf); hipFree(c->slab); hipFree(c->d_bound); hipFree(c->scal); hipFree(c->fr_tmp);
    free_ws(c->ws);
    if (c->own_stream) hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

extern "C" int nsk_sync(nsk_ctx* c)
{
    if (!c) return fail("null ctx");
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->median_fused_pending) {          // k_composite mode 4 ran since the last sync: did its grid barrier time out?
        c->median_fused_pending = false;
        unsigned bar[3] = {0, 0, 0};
        HIPCHK(hipMemcpy(bar, c->scal + 5, sizeof(bar), hipMemcpyDeviceToHost));
        if (bar[2]) {
            HIPCHK(hipMemset(c->scal + 5, 0, sizeof(bar)));
            return fail("nsk_track_step: the grid barrier of the fused median timed out (threshold was infinite for that step); nsk_set_tuning(\"no_fused_median\", 1) selects the three-launch form");
        }
    }
    return 0;
}
extern "C" void* nsk_stream(nsk_ctx* c) { return c ? (void*)c->stream : nullptr; }

static int repack16(nsk_ctx* c, int w);
extern "C" int nsk_set_matmul_mode(nsk_ctx* c, int mode)
{
    if (!c) return fail("null ctx");
    if (mode < 0 || mode > 2) return fail("nsk_set_matmul_mode: mode must be 0 (fp32 MFMA), 1 (bf16 3-piece split) or 2 (fp16 2-piece split)");
    if (c->capturing) return fail("nsk_set_matmul_mode: not while a graph is being captured");
    const bool relayout = (mode == 2) != (c->matmul_mode == 2);
    c->matmul_mode = mode;
    if (relayout) {              // the forward images change their piece format: rebuild those of the loaded decoders
        HIPCHK(hipSetDevice(c->device));
        for (int w = 1; w < 4; ++w) if (c->dec[w].loaded && c->dec[w].fimg16) CHK(repack16(c, w));
        invalidate_graphs(c);
    }
    return 0;
}

extern "C" int nsk_set_backward_mode(nsk_ctx* c, int mode)
{
    if (!c) return fail("null ctx");
    if (mode != 0 && mode != 2) return fail("nsk_set_backward_mode: 0 (fp32 MFMA chains) or 2 (two fp16 pieces, default)");
    c->bwd_mode = mode;
    return 0;
}

extern "C" int nsk_set_tuning(nsk_ctx* c, const char* key, int value)
{
    if (!c || !key) return fail("nsk_set_tuning: null argument");
    if (!strcmp(key, "frozen_cost")) { c->tune_frozen_cost = value; return 0; }
    if (!strcmp(key, "frozen_cost_rays")) { c->tune_frozen_cost_rays = value; return 0; }
    if (!strcmp(key, "no_frozen_kernel")) { c->tune_no_frozen_kernel = value; return 0; }
    if (!strcmp(key, "no_piggyback")) { c->tune_no_piggyback = value; return 0; }
    if (!strcmp(key, "frozen_mid_pct")) { if (value < 10 || value > 1000) return fail("nsk_set_tuning: frozen_mid_pct out of range"); c->tune_frozen_mid_pct = value; return 0; }
    if (!strcmp(key, "no_fused_median")) { c->tune_no_fused_median = value; return 0; }
    if (!strcmp(key, "no_deferred_median")) { c->tune_no_deferred_median = value; return 0; }
    if (!strcmp(key, "fwd_fine_cost")) { c->tune_fwd_fine_cost = value; return 0; }
    if (!strcmp(key, "fwd_occ_cost")) { c->tune_fwd_occ_cost = value; return 0; }
    if (!strcmp(key, "no_occ_role")) { c->tune_no_occ_role = value; return 0; }
    if (!strcmp(key, "fwd_color_cost")) { c->tune_fwd_color_cost = value; return 0; }
    if (!strcmp(key, "skew")) { if (value < 0 || value > 299) return fail("nsk_set_tuning: skew out of range"); c->tune_skew = value; return 0; }
    if (!strcmp(key, "deterministic")) { c->deterministic = value != 0; return 0; }
    if (!strcmp(key, "roctx")) { c->roctx = value != 0; return 0; }
    return fail("nsk_set_tuning: unknown key '%s'", key);
}

extern "C" int nsk_set_ray_mask(nsk_ctx* c, const uint8_t* d_keep)
{
    if (!c) return fail("null ctx");
    c->ray_mask = d_keep;
    return 0;
}

extern "C" int nsk_set_sort_mode(nsk_ctx* c, int mode)
{
    if (!c) return fail("null ctx");
    if (mode < -1 || mode > 1) return fail("nsk_set_sort_mode: mode must be -1 (automatic), 0 (never) or 1 (always)");
    c->sort_mode = mode;
    return 0;
}

extern "C" int nsk_set_bound(nsk_ctx* c, const float h_bound[6])
{
    if (!c || !h_bound) return fail("nsk_set_bound: null argument");
    for (int k = 0; k < 3; ++k) if (!(h_bound[2 * k + 1] > h_bound[2 * k])) return fail("nsk_set_bound: empty axis %d", k);
    memcpy(c->R.bound, h_bound, 6 * sizeof(float));
    HIPCHK(hipMemcpyAsync(c->d_bound, c->R.bound, 6 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int nsk_set_render_opts(nsk_ctx* c, int n_samples, int n_surface, int lindisp, float perturb, int occupancy, uint64_t seed)
{
    if (!c) return fail("null ctx");
    if (n_samples < 1 || n_surface < 0 || n_samples + n_surface > 64) return fail("nsk_set_render_opts: need 1 <= n_samples, n_samples+n_surface <= 64 (got %d+%d)", n_samples, n_surface);
    c->R.n_samples = n_samples; c->R.n_surface = n_surface; c->R.lindisp = lindisp ? 1 : 0; c->R.perturb = perturb;
    c->R.occupancy = occupancy ? 1 : 0; c->R.seed = seed;
    return 0;
}

// gradient slab = [grid0..3 | dec0..3 | 4 floats (loss)]
static int rebuild_slab(nsk_ctx* c)
{
    size_t n = 0;
    for (int i = 0; i < 4; ++i) { c->grid[i].g_off = n; n += c->grid[i].n; }
    for (int i = 0; i < 4; ++i) { c->dec[i].g_off = n; n += (size_t)((c->dec[i].n + 3) & ~3); }
    n += 4;
    if (n != c->slab_n) {
        HIPCHK(hipStreamSynchronize(c->stream));
        invalidate_graphs(c);
        if (c->slab) HIPCHK(hipFree(c->slab));
        HIPCHK(hipMalloc(&c->slab, n * sizeof(float)));
        c->slab_n = n;
    }
    HIPCHK(hipMemsetAsync(c->slab, 0, n * sizeof(float), c->stream));
    return 0;
}

extern "C" int nsk_grid_upload(nsk_ctx* c, int level, const float* h, int C, int Z, int Y, int X)
{
    if (!c || !h) return fail("nsk_grid_upload: null argument");
    if (!which_ok(level)) return fail("nsk_grid_upload: bad level %d", level);
    if (C != 32) return fail("nsk_grid_upload: C must be 32 (got %d)", C);
    if (Z < 1 || Y < 1 || X < 1) return fail("nsk_grid_upload: bad shape");
    if ((size_t)Z * Y * X >= ((size_t)1 << 25)) return fail("nsk_grid_upload: at most 2^25 - 1 voxels per level (32-bit byte offsets in the forward's gather: 128 B per voxel)");
    HIPCHK(hipSetDevice(c->device));
    GridState& G = c->grid[level];
    size_t nvox = (size_t)Z * Y * X, n = nvox * 32;
    bool realloc_ = n != G.n;
    if (realloc_) {
        HIPCHK(hipStreamSynchronize(c->stream));
        invalidate_graphs(c);
        hipFree(G.v); hipFree(G.m); hipFree(G.s); hipFree(G.mask); G.mask = nullptr; hipFree(G.midx); G.midx = nullptr;
        HIPCHK(hipMalloc(&G.v, n * 4)); HIPCHK(hipMalloc(&G.m, n * 4)); HIPCHK(hipMalloc(&G.s, n * 4));
    }
    if (G.Z != Z || G.Y != Y || G.X != X) { if (G.mask) { HIPCHK(hipStreamSynchronize(c->stream)); invalidate_graphs(c); hipFree(G.mask); G.mask = nullptr; } }
    G.C = C; G.Z = Z; G.Y = Y; G.X = X; G.n = n; G.midx_dirty = true;
    std::vector<float> t(n);
    for (int ch = 0; ch < 32; ++ch)
        for (size_t v = 0; v < nvox; ++v) t[v * 32 + ch] = h[(size_t)ch * nvox + v];
    HIPCHK(hipMemcpyAsync(G.v, t.data(), n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemsetAsync(G.m, 0, n * 4, c->stream));
    HIPCHK(hipMemsetAsync(G.s, 0, n * 4, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (realloc_) CHK(rebuild_slab(c));
    return 0;
}

static int grid_fetch(nsk_ctx* c, int level, const float* src, float* h)
{
    GridState& G = c->grid[level];
    if (!G.n) return fail("grid level %d not uploaded", level);
    std::vector<float> t(G.n);
    HIPCHK(hipMemcpyAsync(t.data(), src, G.n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    size_t nvox = G.n / 32;
    for (int ch = 0; ch < 32; ++ch)
        for (size_t v = 0; v < nvox; ++v) h[(size_t)ch * nvox + v] = t[v * 32 + ch];
    return 0;
}
extern "C" int nsk_grid_download(nsk_ctx* c, int level, float* h)
{
    if (!c || !h || !which_ok(level)) return fail("nsk_grid_download: bad argument");
    return grid_fetch(c, level, c->grid[level].v, h);
}
extern "C" int nsk_grid_grad_download(nsk_ctx* c, int level, float* h)
{
    if (!c || !h || !which_ok(level)) return fail("nsk_grid_grad_download: bad argument");
    CHK(grid_fetch(c, level, c->slab + c->grid[level].g_off, h));
    GridState& G = c->grid[level];
    if (G.mask) {                                   // "gradients of unmarked voxels are discarded": what the scatter left there is not a gradient
        const size_t nvox = G.n / 32;
        std::vector<uint8_t> m(nvox);
        HIPCHK(hipMemcpy(m.data(), G.mask, nvox, hipMemcpyDeviceToHost));
        for (int ch = 0; ch < 32; ++ch)
            for (size_t v = 0; v < nvox; ++v) if (!m[v]) h[(size_t)ch * nvox + v] = 0.f;
    }
    return 0;
}

extern "C" int nsk_set_mask(nsk_ctx* c, int level, const uint8_t* h_mask)
{
    if (!c || !which_ok(level)) return fail("nsk_set_mask: bad argument");
    GridState& G = c->grid[level];
    if (!G.n) return fail("nsk_set_mask: grid level %d not uploaded", level);
    if (c->capturing) return fail("nsk_set_mask: not while a graph is being captured");
    HIPCHK(hipStreamSynchronize(c->stream));
    size_t nvox = G.n / 32;
    G.midx_dirty = true;
    // A captured step holds the level's voxel list as kernel arguments (its pointer, its length and the block ranges derived from it:
    // k_adam_multi, k_xchg_multi): new mask CONTENTS make every recorded graph wrong, not only a new allocation.
    invalidate_graphs(c);
    // Unmarked voxels are never visited by the optimiser, so whatever the scatter added to their gradient stays there: a voxel that
    // becomes marked now must not inherit it.  The level's gradient is cleared whenever its mask changes.
    if (c->slab) HIPCHK(hipMemsetAsync(c->slab + G.g_off, 0, G.n * 4, c->stream));
    if (!h_mask) { if (G.mask) { hipFree(G.mask); G.mask = nullptr; } return 0; }
    if (!G.mask) HIPCHK(hipMalloc(&G.mask, nvox));
    HIPCHK(hipMemcpy(G.mask, h_mask, nvox, hipMemcpyHostToDevice));
    return 0;
}

// ---- decoder images -------------------------------------------------------------------------------------
static void seg_fwd(std::vector<int>& idx, int quad0, int KQ, int Wofs, int ld, int col0, int kvalid)
{
    for (int rt = 0; rt < 2; ++rt) for (int q = 0; q < KQ; ++q) for (int lane = 0; lane < 64; ++lane) for (int i = 0; i < 4; ++i) {
        int o = 16 * rt + (lane & 15), k = 16 * q + 4 * (lane >> 4) + i;
        idx[((size_t)(quad0 + rt * KQ + q) * 64 + lane) * 4 + i] = k < kvalid ? Wofs + o * ld + col0 + k : -1;
    }
}
static void seg_bwd(std::vector<int>& idx, int quad0, int RT, int Wofs, int ld, int col0, int rvalid)
{
    const int KQ = 2;
    for (int rt = 0; rt < RT; ++rt) for (int q = 0; q < KQ; ++q) for (int lane = 0; lane < 64; ++lane) for (int i = 0; i < 4; ++i) {
        int r = 16 * rt + (lane & 15), k = 16 * q + 4 * (lane >> 4) + i;
        idx[((size_t)(quad0 + rt * KQ + q) * 64 + lane) * 4 + i] = r < rvalid ? Wofs + k * ld + col0 + r : -1;
    }
}

template <int CQ>
static void build_mlp_fwd_idx(const DecLayout& L, std::vector<int>& idx)
{
    typedef MlpFwdImg<CQ> I;
    idx.assign(I::TOTAL, -1);
    seg_fwd(idx, I::W0E, 6, L.oW[0], NSK_E, 0, NSK_E);
    for (int l = 0; l < 5; ++l) seg_fwd(idx, I::F(l), CQ, L.oFw[l], L.c_dim, 0, L.c_dim);
    seg_fwd(idx, I::W1, 2, L.oW[1], 32, 0, 32);
    seg_fwd(idx, I::W2, 2, L.oW[2], 32, 0, 32);
    seg_fwd(idx, I::W3E, 6, L.oW[3], 125, 0, NSK_E);
    seg_fwd(idx, I::W3H, 2, L.oW[3], 125, NSK_E, 32);
    seg_fwd(idx, I::W4, 2, L.oW[4], 32, 0, 32);
    for (int l = 0; l < 5; ++l) for (int o = 0; o < 32; ++o) { idx[I::P_B + 32 * l + o] = L.ob[l] + o; idx[I::P_BC + 32 * l + o] = L.oFb[l] + o; }
    for (int o = 0; o < L.out_dim; ++o) { for (int k = 0; k < 32; ++k) idx[I::P_WO + 32 * o + k] = L.oWo + 32 * o + k; idx[I::P_BO + o] = L.obo + o; }
    for (int a = 0; a < 3; ++a) for (int k = 0; k < NSK_E; ++k) idx[I::P_BM + 96 * a + k] = L.oB + NSK_E * a + k;
}

static void build_idx(int w, std::vector<int>& fidx, std::vector<int>& bidx)
{
    DecLayout L = nsk_dec_layout(w);
    if (w == 0) {
        typedef CoarseFwdImg I; typedef CoarseBwdImg J;
        fidx.assign(I::TOTAL, -1); bidx.assign(J::TOTAL, -1);
        seg_fwd(fidx, I::W0, 2, L.oW[0], 32, 0, 32); seg_fwd(fidx, I::W1, 2, L.oW[1], 32, 0, 32); seg_fwd(fidx, I::W2, 2, L.oW[2], 32, 0, 32);
        seg_fwd(fidx, I::W3C, 2, L.oW[3], 64, 0, 32); seg_fwd(fidx, I::W3H, 2, L.oW[3], 64, 32, 32); seg_fwd(fidx, I::W4, 2, L.oW[4], 32, 0, 32);
        for (int l = 0; l < 5; ++l) for (int o = 0; o < 32; ++o) fidx[I::P_B + 32 * l + o] = L.ob[l] + o;
        for (int k = 0; k < 32; ++k) fidx[I::P_WO + k] = L.oWo + k;
        fidx[I::P_BO] = L.obo;
        seg_bwd(bidx, J::W0T, 2, L.oW[0], 32, 0, 32); seg_bwd(bidx, J::W1T, 2, L.oW[1], 32, 0, 32); seg_bwd(bidx, J::W2T, 2, L.oW[2], 32, 0, 32);
        seg_bwd(bidx, J::W3CT, 2, L.oW[3], 64, 0, 32); seg_bwd(bidx, J::W3HT, 2, L.oW[3], 64, 32, 32); seg_bwd(bidx, J::W4T, 2, L.oW[4], 32, 0, 32);
        for (int k = 0; k < 32; ++k) bidx[J::P_WO + k] = L.oWo + k;
        return;
    }
    if (w == 2) build_mlp_fwd_idx<4>(L, fidx); else build_mlp_fwd_idx<2>(L, fidx);
    typedef MlpBwdImg J;
    bidx.assign(J::TOTAL, -1);
    for (int l = 0; l < 5; ++l) seg_bwd(bidx, J::FT(l), 2, L.oFw[l], L.c_dim, 0, 32);
    for (int l = 1; l < 5; ++l) seg_bwd(bidx, J::WT(l), 2, L.oW[l], L.in_dim[l], l == 3 ? NSK_E : 0, 32);
    seg_bwd(bidx, J::W0ET, 6, L.oW[0], NSK_E, 0, NSK_E);
    seg_bwd(bidx, J::W3ET, 6, L.oW[3], 125, 0, NSK_E);
    for (int o = 0; o < L.out_dim; ++o) for (int k = 0; k < 32; ++k) bidx[J::P_WO + 32 * o + k] = L.oWo + 32 * o + k;
    for (int a = 0; a < 3; ++a) for (int k = 0; k < NSK_E; ++k) bidx[J::P_BM + 96 * a + k] = L.oB + NSK_E * a + k;
}

static void seg16(std::vector<int>& idx, int blk0, int NB, int Wofs, int ld, int col0, int kvalid)
{
    for (int b = 0; b < NB; ++b) for (int rt = 0; rt < 2; ++rt) for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 8; ++j) {
        int fg = 2 * (blk0 + b) + rt, o = 16 * rt + (lane & 15), k = 32 * b + nsk_bf16_kperm(lane >> 4, j);
        idx[((size_t)fg * 64 + lane) * 8 + j] = k < kvalid ? Wofs + o * ld + col0 + k : -1;
    }
}
template <int CQ>
static void build_idx16(const DecLayout& L, std::vector<int>& idx)
{
    typedef MlpFwdImgB<CQ> I;
    idx.assign((size_t)I::NBLK * 2 * 512, -1);
    seg16(idx, I::W0E, 3, L.oW[0], NSK_E, 0, NSK_E);
    for (int l = 0; l < 5; ++l) seg16(idx, I::F(l), I::CB, L.oFw[l], L.c_dim, 0, L.c_dim);
    seg16(idx, I::W1, 1, L.oW[1], 32, 0, 32); seg16(idx, I::W2, 1, L.oW[2], 32, 0, 32); seg16(idx, I::W4, 1, L.oW[4], 32, 0, 32);
    seg16(idx, I::W3E, 3, L.oW[3], 125, 0, NSK_E); seg16(idx, I::W3H, 1, L.oW[3], 125, NSK_E, 32);
}

static void build_idx16b(const DecLayout& L, std::vector<int>& idx)
{
    typedef MlpBwdImgH J;
    idx.assign((size_t)J::NFG * 512, -1);
    auto seg = [&](int fg0, int nrt, int Wofs, int ld, int col0, int rows) {   // transposed: row o = input column, k = output row of the forward weight
        for (int rt = 0; rt < nrt; ++rt) for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 8; ++j) {
            const int o = 16 * rt + (lane & 15), k = nsk_bf16_kperm(lane >> 4, j);
            idx[((size_t)(fg0 + rt) * 64 + lane) * 8 + j] = o < rows ? Wofs + k * ld + col0 + o : -1;
        }
    };
    for (int l = 0; l < 5; ++l) seg(J::FT(l), 2, L.oFw[l], L.c_dim, 0, 32);
    for (int l = 1; l < 5; ++l) seg(J::WT(l), 2, L.oW[l], L.in_dim[l], l == 3 ? NSK_E : 0, 32);
    seg(J::W0ET, 6, L.oW[0], NSK_E, 0, NSK_E);
    seg(J::W3ET, 6, L.oW[3], 125, 0, NSK_E);
}

// the fp16 backward image (MlpBwdImgH) follows the parameters lazily: marked stale by uploads and by Adam steps on a decoder whose image
// the optimiser launch does not rewrite itself, rebuilt before use
static int ensure_bimg16(nsk_ctx* c, int w)
{
    DecState& D = c->dec[w];
    if (!D.bimg16 || !D.bimg16_dirty) return 0;
    ProfScope ps(c, "pack_bwd_bf16");
    k_pack_bf16<<<(D.bfrag16_n + 255) / 256, 256, 0, c->stream>>>(reinterpret_cast<unsigned short*>(D.bimg16), D.bidx16, D.p, D.bfrag16_n, 2);
    k_pack<<<2, 256, 0, c->stream>>>(D.bimg16 + MlpBwdImgH::P_WO, D.bidx + MlpBwdImg::P_WO, D.p, 128 + 288);     // fp32 tail: Wo, B
    HIPCHK(hipGetLastError());
    D.bimg16_dirty = false;
    return 0;
}

// the forward image holds 3 bf16 pieces (matmul modes 0 / 1) or 2 fp16 pieces (mode 2) per weight; the fp32 tail follows the fragments
static void set_np16(DecState& D, int np)
{
    D.np16 = np;
    if (D.cq16 == 4) { D.fimg16_f = np == 2 ? MlpFwdImgB<4, 2>::TOTAL_F : MlpFwdImgB<4, 3>::TOTAL_F; D.tail16_off = np == 2 ? MlpFwdImgB<4, 2>::P_F32 : MlpFwdImgB<4, 3>::P_F32; }
    else { D.fimg16_f = np == 2 ? MlpFwdImgB<2, 2>::TOTAL_F : MlpFwdImgB<2, 3>::TOTAL_F; D.tail16_off = np == 2 ? MlpFwdImgB<2, 2>::P_F32 : MlpFwdImgB<2, 3>::P_F32; }
}

static int repack16(nsk_ctx* c, int w)
{
    DecState& D = c->dec[w];
    if (!D.fimg16) return 0;
    set_np16(D, c->matmul_mode == 2 ? 2 : 3);
    k_pack_bf16<<<(D.frag16_n + 255) / 256, 256, 0, c->stream>>>(reinterpret_cast<unsigned short*>(D.fimg16), D.fidx16, D.p, D.frag16_n, D.np16);
    k_pack<<<(740 + 255) / 256, 256, 0, c->stream>>>(D.fimg16 + D.tail16_off, D.fidx + D.tail_off, D.p, 740);     // fp32 tail: biases, Wo, bo, B
    HIPCHK(hipGetLastError());
    return 0;
}

static int repack(nsk_ctx* c, int w)
{
    DecState& D = c->dec[w];
    ProfScope ps(c, "pack_images");
    k_pack<<<(D.fimg_n + 255) / 256, 256, 0, c->stream>>>(D.fimg, D.fidx, D.p, D.fimg_n);
    k_pack<<<(D.bimg_n + 255) / 256, 256, 0, c->stream>>>(D.bimg, D.bidx, D.p, D.bimg_n);
    HIPCHK(hipGetLastError());
    CHK(repack16(c, w));
    D.bimg16_dirty = true;
    return 0;
}

static int flush_pending(nsk_ctx* c);
static int ensure_midx(nsk_ctx* c, int l);
static void adam_consts(float lr, float b1, float b2, int step, float& step_size, float& bc2s);

extern "C" size_t nsk_decoder_param_count(int which) { return which_ok(which) ? (size_t)nsk_dec_layout(which).total : 0; }

extern "C" int nsk_decoder_upload(nsk_ctx* c, int w, const float* h, size_t n)
{
    if (!c || !h || !which_ok(w)) return fail("nsk_decoder_upload: bad argument");
    if (n != nsk_decoder_param_count(w)) return fail("nsk_decoder_upload: decoder %d expects %zu parameters, got %zu", w, nsk_decoder_param_count(w), n);
    HIPCHK(hipSetDevice(c->device));
    DecState& D = c->dec[w];
    if (!D.p) {
        D.n = (int)n;
        size_t n4 = (n + 3) & ~(size_t)3;
        HIPCHK(hipMalloc(&D.p, n4 * 4)); HIPCHK(hipMalloc(&D.m, n4 * 4)); HIPCHK(hipMalloc(&D.s, n4 * 4));
        HIPCHK(hipMemset(D.p, 0, n4 * 4));
        std::vector<int> fi, bi;
        build_idx(w, fi, bi);
        D.fimg_n = (int)fi.size(); D.bimg_n = (int)bi.size();
        HIPCHK(hipMalloc(&D.fimg, fi.size() * 4)); HIPCHK(hipMalloc(&D.bimg, bi.size() * 4));
        HIPCHK(hipMalloc(&D.fidx, fi.size() * 4)); HIPCHK(hipMalloc(&D.bidx, bi.size() * 4));
        HIPCHK(hipMemcpy(D.fidx, fi.data(), fi.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(D.bidx, bi.data(), bi.size() * 4, hipMemcpyHostToDevice));
        if (w != 0) {
            std::vector<int> i16;
            if (w == 2) { build_idx16<4>(nsk_dec_layout(w), i16); D.cq16 = 4; D.tail_off = MlpFwdImg<4>::P_B; }
            else { build_idx16<2>(nsk_dec_layout(w), i16); D.cq16 = 2; D.tail_off = MlpFwdImg<2>::P_B; }
            set_np16(D, 3);          // allocate for the larger (3-piece) form
            D.frag16_n = (int)i16.size();
            HIPCHK(hipMalloc(&D.fimg16, (size_t)D.fimg16_f * 4)); HIPCHK(hipMalloc(&D.fidx16, i16.size() * 4));
            HIPCHK(hipMemset(D.fimg16, 0, (size_t)D.fimg16_f * 4));
            HIPCHK(hipMemcpy(D.fidx16, i16.data(), i16.size() * 4, hipMemcpyHostToDevice));
            std::vector<int> ib;
            build_idx16b(nsk_dec_layout(w), ib);
            D.bfrag16_n = (int)ib.size();
            HIPCHK(hipMalloc(&D.bimg16, (size_t)MlpBwdImgH::TOTAL_F * 4)); HIPCHK(hipMalloc(&D.bidx16, ib.size() * 4));
            HIPCHK(hipMemset(D.bimg16, 0, (size_t)MlpBwdImgH::TOTAL_F * 4));
            HIPCHK(hipMemcpy(D.bidx16, ib.data(), ib.size() * 4, hipMemcpyHostToDevice));
            std::vector<int> invh(n4, -1);
            for (size_t k = 0; k < ib.size(); ++k) if (ib[k] >= 0) { if (invh[ib[k]] != -1) return fail("decoder %d: parameter %d appears twice in the fp16 backward image", w, ib[k]); invh[ib[k]] = (int)k; }
            HIPCHK(hipMalloc(&D.invh, n4 * 4));
            HIPCHK(hipMemcpy(D.invh, invh.data(), n4 * 4, hipMemcpyHostToDevice));
            std::vector<int> inv16(n4, -1);
            for (size_t k = 0; k < i16.size(); ++k) if (i16[k] >= 0) { if (inv16[i16[k]] != -1) return fail("decoder %d: parameter %d appears twice in the bf16 image", w, i16[k]); inv16[i16[k]] = (int)k; }
            HIPCHK(hipMalloc(&D.inv16, n4 * 4));
            HIPCHK(hipMemcpy(D.inv16, inv16.data(), n4 * 4, hipMemcpyHostToDevice));
        }
        {
            std::vector<int> finv(n4, -1), binv(n4, -1);
            for (size_t k = 0; k < fi.size(); ++k) if (fi[k] >= 0) { if (finv[fi[k]] != -1) return fail("decoder %d: parameter %d appears twice in the forward image", w, fi[k]); finv[fi[k]] = (int)k; }
            for (size_t k = 0; k < bi.size(); ++k) if (bi[k] >= 0) { if (binv[bi[k]] != -1) return fail("decoder %d: parameter %d appears twice in the backward image", w, bi[k]); binv[bi[k]] = (int)k; }
            HIPCHK(hipMalloc(&D.finv, n4 * 4)); HIPCHK(hipMalloc(&D.binv, n4 * 4));
            HIPCHK(hipMemcpy(D.finv, finv.data(), n4 * 4, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(D.binv, binv.data(), n4 * 4, hipMemcpyHostToDevice));
        }
        CHK(rebuild_slab(c));
    }
    HIPCHK(hipMemcpyAsync(D.p, h, n * 4, hipMemcpyHostToDevice, c->stream));
    size_t n4 = (n + 3) & ~(size_t)3;
    HIPCHK(hipMemsetAsync(D.m, 0, n4 * 4, c->stream));
    HIPCHK(hipMemsetAsync(D.s, 0, n4 * 4, c->stream));
    CHK(repack(c, w));
    HIPCHK(hipStreamSynchronize(c->stream));
    D.loaded = true;
    return 0;
}

static int dec_fetch(nsk_ctx* c, int w, const float* src, float* h, size_t n)
{
    if (!c->dec[w].loaded) return fail("decoder %d not uploaded", w);
    if (n != (size_t)c->dec[w].n) return fail("decoder %d has %d parameters, buffer has %zu", w, c->dec[w].n, n);
    HIPCHK(hipMemcpyAsync(h, src, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}
extern "C" int nsk_decoder_download(nsk_ctx* c, int w, float* h, size_t n)
{
    if (!c || !h || !which_ok(w)) return fail("nsk_decoder_download: bad argument");
    return dec_fetch(c, w, c->dec[w].p, h, n);
}
extern "C" int nsk_decoder_grad_download(nsk_ctx* c, int w, float* h, size_t n)
{
    if (!c || !h || !which_ok(w)) return fail("nsk_decoder_grad_download: bad argument");
    CHK(flush_pending(c));
    return dec_fetch(c, w, c->slab + c->dec[w].g_off, h, n);
}
extern "C" int nsk_decoder_set_trainable(nsk_ctx* c, int w, int t)
{
    if (!c || !which_ok(w)) return fail("nsk_decoder_set_trainable: bad argument");
    c->dec[w].trainable = t ? 1 : 0;
    return 0;
}

// ---- workspace --------------------------------------------------------------------------------------------
static int prep_drop(nsk_ctx* c);
static int ensure_ws(nsk_ctx* c, int N, int M)
{
    Workspace& w = c->ws;
    if (M <= w.capM && N <= w.capN) return 0;
    if (c->capturing) return fail("graph capture: the workspace must grow (run the same step once before nsk_graph_begin)");
    HIPCHK(hipStreamSynchronize(c->stream));
    CHK(prep_drop(c));
    invalidate_graphs(c);
    int capM = std::max(M, w.capM), capN = std::max(N, w.capN);
    free_ws(w);
    size_t m = (size_t)capM + 64;
    HIPCHK(hipMalloc(&w.z, m * 4));
    for (int i = 0; i < 3; ++i) HIPCHK(hipMalloc(&w.occ[i], m * 4));
    HIPCHK(hipMalloc(&w.rgb4, m * 16));
    for (int i = 0; i < 4; ++i) HIPCHK(hipMalloc(&w.masks[i], m * 32));
    HIPCHK(hipMalloc(&w.g_raw, m * 16));
    size_t n = (size_t)capN + 64;
    HIPCHK(hipMalloc(&w.ray_loss, n * 4));
    HIPCHK(hipMalloc(&w.tmp_rgb, n * 12)); HIPCHK(hipMalloc(&w.tmp_depth, n * 4)); HIPCHK(hipMalloc(&w.tmp_var, n * 4));
    HIPCHK(hipMalloc(&w.dec_slabs, (size_t)c->num_cu * 20920 * 4));
    HIPCHK(hipMemsetAsync(w.dec_slabs, 0, (size_t)c->num_cu * 20920 * 4, c->stream));
    HIPCHK(hipMalloc(&w.perm, m * 4)); HIPCHK(hipMalloc(&w.skey, m * 4)); HIPCHK(hipMalloc(&w.srank, m * 4));
    w.capM = capM; w.capN = capN;
    return 0;
}

// histogram / offsets of the cell sort: one bin per cell of the key level
static int ensure_hist(nsk_ctx* c, size_t bins)
{
    Workspace& w = c->ws;
    if (bins <= w.hist_cap) return 0;
    if (c->capturing) return fail("graph capture: the workspace must grow (run the same step once before nsk_graph_begin)");
    HIPCHK(hipStreamSynchronize(c->stream));
    c->prep.valid = false; c->req.valid = false;       // (the histogram is replaced: nothing to clean)
    invalidate_graphs(c);
    if (w.hist) hipFree(w.hist - 16);
    hipFree(w.offs);
    const size_t slots = ((bins / 8 + 1) / 2) * 16 + 16;    // hist_slot() layout (bins = 8 keys per cell) + one 64-byte line in front (hist[-1] = k_sort_scan's cursor)
    int* raw = nullptr;
    HIPCHK(hipMalloc(&raw, slots * 4)); HIPCHK(hipMalloc(&w.offs, bins * 4));
    HIPCHK(hipMemsetAsync(raw, 0, slots * 4, c->stream));
    w.hist = raw + 16;
    w.hist_cap = bins;
    return 0;
}

// the forward of a trainable decoder (all but the fine one, see nsk_train.h) also stores its block outputs for the backward
static bool saves_h(nsk_ctx* c, int w, bool save_masks) { return save_masks && w != 2 && c->dec[w].trainable; }
static int ensure_hsave(nsk_ctx* c, int w, int M)
{
    Workspace& ws = c->ws;
    const size_t tiles = (size_t)(M + 15) / 16 + 1;
    if (tiles > ws.hcap[w]) {
        if (c->capturing) return fail("graph capture: the workspace must grow (run the same step once before nsk_graph_begin)");
        HIPCHK(hipStreamSynchronize(c->stream));
        invalidate_graphs(c);
        hipFree(ws.hsave[w]);
        HIPCHK(hipMalloc(&ws.hsave[w], tiles * 10 * 64 * sizeof(f4)));
        ws.hcap[w] = tiles;
    }
    return 0;
}

static const int STAGE_DEC[4][3] = {{0, -1, -1}, {1, -1, -1}, {1, 2, -1}, {1, 2, 3}};

static GridD grid_dev(nsk_ctx* c, int level, bool with_grad)
{
    GridD g;
    GridState& G = c->grid[level];
    g.v = G.v; g.g = with_grad ? c->slab + G.g_off : nullptr; g.mask = G.mask; g.Z = G.Z; g.Y = G.Y; g.X = G.X;
    return g;
}

static int check_stage(nsk_ctx* c, int stage)
{
    if (stage < 0 || stage > 3) return fail("bad stage %d", stage);
    for (int q = 0; q < 3; ++q) {
        int w = STAGE_DEC[stage][q];
        if (w < 0) break;
        if (!c->grid[w].n) return fail("stage %d needs grid level %d (nsk_grid_upload)", stage, w);
        if (!c->dec[w].loaded) return fail("stage %d needs decoder %d (nsk_decoder_upload)", stage, w);
    }
    return 0;
}

static void fill_args(nsk_ctx* c, DecArgs& A, int w, int M, int S, const float* ro, const float* rd, const float* pts)
{
    memset(&A, 0, sizeof(A));
    A.rays_o = ro; A.rays_d = rd; A.z = c->ws.z; A.pts = pts; A.M = M; A.S = S; A.S_magic = S > 1 ? (unsigned)((0x100000000ull + (unsigned)S - 1) / (unsigned)S) : 0u;
    A.perm = (c->sorted && !pts) ? c->ws.perm : nullptr;
    memcpy(A.bound, c->R.bound, sizeof(A.bound));
    A.grid = grid_dev(c, w, false);
    if (w == 2) A.grid_mid = grid_dev(c, 1, false);
    A.img = reinterpret_cast<const f4*>(c->dec[w].fimg);
    A.bimg = reinterpret_cast<const f4*>(c->dec[w].bimg);
    A.img_f4 = c->dec[w].fimg_n / 4;
    A.img16 = c->dec[w].fimg16;
    A.bimg16 = c->dec[w].bimg16;
    A.out = w == 3 ? c->ws.rgb4 : c->ws.occ[w];
    A.skew = c->tune_skew;
}

static int launch_decode_fwd(nsk_ctx* c, int w, int M, int S, const float* ro, const float* rd, const float* pts, bool save_masks)
{
    DecArgs A;
    fill_args(c, A, w, M, S, ro, rd, pts);
    A.masks = save_masks ? c->ws.masks[w] : nullptr;
    if (save_masks) c->ws.hsave_M[w] = 0;
    if (saves_h(c, w, save_masks)) { CHK(ensure_hsave(c, w, M)); A.hsave = c->ws.hsave[w]; c->ws.hsave_M[w] = M; }
    int ntasks = (M + 15) / 16;
    size_t lds = fwd_img_floats(w) * 4;
    int maxwg = c->num_cu;        // persistent: one workgroup per CU balances the 16-sample tasks over SIMDs
    int grid = std::max(1, std::min((ntasks + 7) / 8, maxwg));
    static const char* names[4] = {"decode_fwd_coarse", "decode_fwd_middle", "decode_fwd_fine", "decode_fwd_color"};
    ProfScope ps(c, names[w]);
    switch (w) {
    case 0: k_decode_fwd<0><<<grid, 512, lds, c->stream>>>(A); break;
    case 1: k_decode_fwd<1><<<grid, 512, lds, c->stream>>>(A); break;
    case 2: k_decode_fwd<2><<<grid, 512, lds, c->stream>>>(A); break;
    default: k_decode_fwd<3><<<grid, 512, lds, c->stream>>>(A); break;
    }
    HIPCHK(hipGetLastError());
    return 0;
}

static int launch_decode_bwd(nsk_ctx* c, int w, int M, int S, const float* ro, const float* rd, bool train, bool rays, unsigned flags,
                             float* g_ro, float* g_rd)
{
    DecArgs A;
    fill_args(c, A, w, M, S, ro, rd, nullptr);
    A.grid = grid_dev(c, w, (flags & NSK_GRAD_GRIDS) != 0);
    A.masks = c->ws.masks[w];
    A.g_raw = c->ws.g_raw;
    A.g_rays_o = g_ro; A.g_rays_d = g_rd;
    A.g_dec = train ? c->ws.dec_slabs : c->slab + c->dec[w].g_off;
    A.flags = flags;
    if (w != 0 && (!train || w != 2)) CHK(ensure_bimg16(c, w));      // the chains on fp16 pieces: every frozen MLP decoder, trainable middle / colour
    if (train && w != 2) {
        if (c->ws.hsave_M[w] != M) return fail("backward of trainable decoder %d: its forward must run with the decoder already trainable (block outputs not saved)", w);
        A.hsave = c->ws.hsave[w];
    }
    int ntasks = (M + 15) / 16;
    size_t lds = bwd_lds_bytes(w, train);
    int grid = std::max(1, std::min((ntasks + 7) / 8, c->num_cu));
    if (c->deterministic) { grid = 1; A.flags |= 0x8000u; }      // one workgroup, its waves scatter one after the other
    static const char* names[8] = {"decode_bwd_coarse", "decode_bwd_middle", "decode_bwd_fine", "decode_bwd_color",
                                   "decode_bwd_coarse_train", "decode_bwd_middle_train", "decode_bwd_fine_train", "decode_bwd_color_train"};
    {
    ProfScope ps(c, names[w + (train ? 4 : 0)]);
#define LB(W) \
    if (train) { if (rays) k_decode_bwd_train<W, true><<<grid, 512, lds, c->stream>>>(A); else k_decode_bwd_train<W, false><<<grid, 512, lds, c->stream>>>(A); } \
    else { if (rays) k_decode_bwd<W, true><<<grid, 512, lds, c->stream>>>(A); else k_decode_bwd<W, false><<<grid, 512, lds, c->stream>>>(A); }
    switch (w) {
    case 0: LB(0) break;
    case 1: LB(1) break;
    case 2: LB(2) break;
    default: LB(3) break;
    }
#undef LB
    }
    HIPCHK(hipGetLastError());
    if (train) {
        int n = c->dec[w].n, n4 = (n + 3) & ~3;
        ProfScope ps2(c, "dec_grad_reduce");
        k_dec_grad_reduce<<<dim3((n + 255) / 256, 8), 256, 0, c->stream>>>(n, n4, grid, c->ws.dec_slabs, c->slab + c->dec[w].g_off);
        HIPCHK(hipGetLastError());
    }
    return 0;
}

// split num_cu workgroups over roles in proportion to their cost per task (every role gets at least one)
static void split_wgs(int num_cu, int ntasks, int n, const int* cost, int* wg_end, int waves = 8)
{
    int cap = std::max(1, (ntasks + waves - 1) / waves), tot = 0, used = 0;
    for (int r = 0; r < n; ++r) tot += cost[r];
    for (int r = 0; r < n; ++r) {
        int k = std::max(1, (int)((double)num_cu * cost[r] / tot));
        k = std::min(k, cap);
        used += k;
        wg_end[r] = used;
    }
}

// Cost-proportional shares, then single workgroups moved from the role that would suffer least to the role that finishes last while
// the modelled makespan (whole tiles per wave x cost) falls: a role's time is a step function of its workgroups, and the
// proportional split alone left the forward 3-5 % behind the best split whenever a role sat just past a step (K3, K4 shard).
static void split_wgs_balanced(int num_cu, int ntasks, int n, const int* cost, int* wg_end, int waves = 8)
{
    split_wgs(num_cu, ntasks, n, cost, wg_end, waves);
    int w[4];
    for (int r = 0; r < n; ++r) w[r] = wg_end[r] - (r ? wg_end[r - 1] : 0);
    int used = wg_end[n - 1];
    const int cap = std::max(1, (ntasks + waves - 1) / waves);
    auto t_of = [&](int r, int wr) { return (long)((ntasks + waves * wr - 1) / (waves * wr)) * cost[r]; };
    for (int r = 0; used < num_cu && r < 8 * n; ++r) {      // hand out what the rounding left, to whoever finishes last
        int worst = 0;
        for (int q = 1; q < n; ++q) if (t_of(q, w[q]) > t_of(worst, w[worst])) worst = q;
        if (w[worst] >= cap) break;
        ++w[worst]; ++used;
    }
    for (int it = 0; it < 64; ++it) {
        int worst = 0;
        for (int q = 1; q < n; ++q) if (t_of(q, w[q]) > t_of(worst, w[worst])) worst = q;
        const long cur = t_of(worst, w[worst]);
        if (w[worst] >= cap) break;
        int donor = -1; long best = cur;
        for (int q = 0; q < n; ++q) {
            if (q == worst || w[q] <= 1) continue;
            long m = std::max(t_of(q, w[q] - 1), t_of(worst, w[worst] + 1));
            for (int o = 0; o < n; ++o) if (o != q && o != worst) m = std::max(m, t_of(o, w[o]));
            if (m < best) { best = m; donor = q; }
        }
        if (donor < 0) break;
        --w[donor]; ++w[worst];
    }
    int acc = 0;
    for (int r = 0; r < n; ++r) { acc += w[r]; wg_end[r] = acc; }
}

// Backward with one trainable role: that role advances in whole iterations (8 tasks per workgroup, all its workgroups in
// lockstep), so its time is ceil(groups / workgroups) iterations -- a step function -- while a frozen role's time falls
// smoothly with its workgroups.  Pick the trainable role's share by minimising the modelled makespan instead of in
// proportion to cost (1024 rays: 3 iterations with the proportional 191 workgroups, 2 with 192).
static void split_wgs_train(int num_cu, int ntasks, int n, const int* cost, int train_role, int* wg_end)
{
    const int groups = std::max(1, (ntasks + 7) / 8);
    int fsum = 0;
    for (int r = 0; r < n; ++r) if (r != train_role) fsum += cost[r];
    // For every iteration count the role could run, give it the FEWEST workgroups that reach it: more would not shorten it (its time is
    // a step function) and would starve the frozen roles (1250 rays: 3 iterations need 157 workgroups; the 190 a cost-proportional
    // split hands it left the frozen roles as the kernel's tail, 120 us against 94 us).
    long best = -1; int best_wt = 1;
    const int wt_max = std::min(groups, num_cu - (n - 1));
    for (int iters = (groups + wt_max - 1) / wt_max; iters <= groups; ++iters) {
        const int wt = (groups + iters - 1) / iters;
        if (wt > wt_max) continue;
        long t = (long)iters * cost[train_role];
        const int rest = num_cu - wt;
        for (int r = 0; r < n; ++r) {
            if (r == train_role) continue;
            const int wr = std::max(1, (int)((long)rest * cost[r] / std::max(1, fsum)));
            t = std::max(t, (long)((ntasks + 8 * wr - 1) / (8 * wr)) * cost[r]);
        }
        if (best < 0 || t < best) { best = t; best_wt = wt; }
        if ((long)iters * cost[train_role] > best) break;          // more iterations only get slower from here
    }
    const int rest = num_cu - best_wt;
    int used = 0;
    for (int r = 0; r < n; ++r) {
        int k = r == train_role ? best_wt : std::max(1, (int)((long)rest * cost[r] / std::max(1, fsum)));
        k = std::min(k, std::max(1, (ntasks + 7) / 8));
        used += k;
        wg_end[r] = used;
    }
}

// predicted time of a split in the cost units of its roles: every wave of a role walks ceil(tiles / waves) tiles
static long split_makespan(int ntasks, int n, const int* cost, const int* wg_end, int waves = 8)
{
    long t = 0;
    for (int r = 0; r < n; ++r) {
        const int w = std::max(1, wg_end[r] - (r ? wg_end[r - 1] : 0));
        t = std::max(t, (long)((ntasks + waves * w - 1) / (waves * w)) * cost[r]);
    }
    return t;
}

// all decoders of the stage in ONE launch (workgroup roles), or a plain launch when the stage has one decoder
static int launch_decode_fwd_stage(nsk_ctx* c, int stage, int M, int S, const float* ro, const float* rd, bool save_masks)
{
    int n = 0;
    for (int q = 0; q < 3; ++q) if (STAGE_DEC[stage][q] >= 0) ++n;
    if (n == 1) return launch_decode_fwd(c, STAGE_DEC[stage][0], M, S, ro, rd, nullptr, save_masks);
    static const int fcost[4] = {96, 240, 292, 248};      // issue cycles per tile of the roles (coarse fp32; middle 7 560, fine 9 380, colour 7 710 + its block-output stores)
    MultiArgs MA;
    memset(&MA, 0, sizeof(MA));
    // middle + fine as ONE role (decode_fwd_occ_body: the middle level looked up once): frozen occupancy decoders on two-piece operands
    if (c->matmul_mode == 2 && c->tune_no_occ_role != 1 && STAGE_DEC[stage][0] == 1 && STAGE_DEC[stage][1] == 2 && !saves_h(c, 1, save_masks) && !saves_h(c, 2, save_masks)) {
        const bool colour = STAGE_DEC[stage][2] == 3;
        for (int r = 0; r < (colour ? 3 : 2); ++r) {
            const int w = STAGE_DEC[stage][r];
            fill_args(c, MA.a[r], w, M, S, ro, rd, nullptr);
            MA.a[r].masks = save_masks ? c->ws.masks[w] : nullptr;
            if (save_masks) c->ws.hsave_M[w] = 0;
            MA.which[r] = w;
        }
        if (colour && saves_h(c, 3, save_masks)) { CHK(ensure_hsave(c, 3, M)); MA.a[2].hsave = c->ws.hsave[3]; c->ws.hsave_M[3] = M; }
        int cost[2] = {c->tune_fwd_occ_cost > 0 ? c->tune_fwd_occ_cost : 460, c->tune_fwd_color_cost > 0 ? c->tune_fwd_color_cost : fcost[3]};
        MA.n = colour ? 2 : 1;
        split_wgs_balanced(c->num_cu, (M + 15) / 16, MA.n, cost, MA.wg_end, 8);
        if (!colour) MA.wg_end[1] = MA.wg_end[0];
        // A merged tile is two decoders long, so small batches quantise worse (K2: 3 000 tiles on 256 workgroups -- forward 37.7 us as three roles,
        // 40.2 as two): take the form whose split predicts the shorter launch.  (tune_no_occ_role 2: always merged)
        bool merged = true;
        if (c->tune_no_occ_role != 2) {
            int cost3[3], wg3[3], n3 = colour ? 3 : 2;
            for (int r = 0; r < n3; ++r) {
                const int w = STAGE_DEC[stage][r];
                cost3[r] = (w == 2 && c->tune_fwd_fine_cost > 0) ? c->tune_fwd_fine_cost : ((w == 3 && c->tune_fwd_color_cost > 0) ? c->tune_fwd_color_cost : fcost[w]);
            }
            split_wgs_balanced(c->num_cu, (M + 15) / 16, n3, cost3, wg3, 8);
            merged = split_makespan((M + 15) / 16, MA.n, cost, MA.wg_end) <= split_makespan((M + 15) / 16, n3, cost3, wg3);
        }
        if (merged) {
            size_t lds_occ = ((size_t)c->dec[1].fimg16_f + (size_t)c->dec[2].fimg16_f) * 4;
            if (colour) lds_occ = std::max(lds_occ, (size_t)c->dec[3].fimg16_f * 4);
            ProfScope ps(c, "decode_fwd_multi");
            k_decode_fwd_multi_occ<8><<<MA.wg_end[MA.n - 1], 512, lds_occ, c->stream>>>(MA);
            HIPCHK(hipGetLastError());
            return 0;
        }
        memset(&MA, 0, sizeof(MA));
    }
    int cost[3]; size_t lds = 0;
    for (int r = 0; r < n; ++r) {
        int w = STAGE_DEC[stage][r];
        fill_args(c, MA.a[r], w, M, S, ro, rd, nullptr);
        MA.a[r].masks = save_masks ? c->ws.masks[w] : nullptr;
        if (save_masks) c->ws.hsave_M[w] = 0;
        if (saves_h(c, w, save_masks)) { CHK(ensure_hsave(c, w, M)); MA.a[r].hsave = c->ws.hsave[w]; c->ws.hsave_M[w] = M; }
        MA.which[r] = w; cost[r] = (w == 2 && c->tune_fwd_fine_cost > 0) ? c->tune_fwd_fine_cost : ((w == 3 && c->tune_fwd_color_cost > 0) ? c->tune_fwd_color_cost : fcost[w]);
        lds = std::max(lds, fwd_img_floats(w) * 4);
    }
    MA.n = n;
    split_wgs_balanced(c->num_cu, (M + 15) / 16, n, cost, MA.wg_end, 8);
    ProfScope ps(c, "decode_fwd_multi");
    if (c->matmul_mode != 0) {
        size_t lds16 = 0;
        for (int r = 0; r < n; ++r) lds16 = std::max(lds16, MA.which[r] == 0 ? fwd_img_floats(0) * 4 : (size_t)c->dec[MA.which[r]].fimg16_f * 4);
        if (c->matmul_mode == 2) k_decode_fwd_multi_bf16<8, 2><<<MA.wg_end[n - 1], 512, lds16, c->stream>>>(MA);
        else k_decode_fwd_multi_bf16<8><<<MA.wg_end[n - 1], 512, lds16, c->stream>>>(MA);       // 12 waves (168 VGPRs) spill: 59 -> 71 us (the two-piece form at 12 waves: 50 spilled registers, K3 forward 139 -> 202 us, round 4)
    } else
    k_decode_fwd_multi<<<MA.wg_end[n - 1], 512, lds, c->stream>>>(MA);
    HIPCHK(hipGetLastError());
    return 0;
}

// algorithmic bytes / flops per sample
static void account(nsk_ctx* c, int stage, int M, int N, bool bwd, unsigned flags)
{
    static const double fwd_b[4] = {1024, 1024, 2048, 3072}, fwd_f[4] = {12.4e3, 31.0e3, 72.2e3, 103.3e3};
    double b = fwd_b[stage] + 5.0, f = fwd_f[stage];
    if (bwd) {
        int lv = stage == 0 ? 1 : stage;      // levels receiving gradient
        if (flags & NSK_GRAD_GRIDS) b += 2048.0 * lv;
        f *= 2.0;
        if (flags & NSK_GRAD_DECODERS) for (int q = 0; q < 3; ++q) { int w = STAGE_DEC[stage][q]; if (w >= 0 && c->dec[w].trainable) f += 2.0 * (w == 0 ? 6176 : (w == 2 ? 20599 : 15500)); }
    }
    c->last_bytes = b * M; c->last_flops = f * M; c->last_samples = M; (void)N;
}

// sampling + decoders of the stage; leaves z / occ / rgb4 (and ReLU bits) in the workspace
// The cell sort pays when the step scatters into the grids and does not need per-ray gradient sums (those rely on a tile's 16
// samples sharing a ray: the Tracker and bundle adjustment keep the ray order)
static bool sort_pays(nsk_ctx* c, int stage, int M, unsigned flags)
{
    if (c->deterministic) return false;          // (the cell sort's ranks come from atomics in arrival order)
    if (c->sort_mode >= 0) return c->sort_mode == 1;
    if (!(flags & NSK_GRAD_GRIDS)) return false;             // (the Tracker: 200 rays, no scatter)
    // Bundle adjustment (grids + rays): in cell order a tile's 16 samples belong to 16 rays, so the ray-gradient sums go lane by lane instead of
    // one add per tile -- and the scatter still wins: per-role stamps at 1000 / 5000 rays (tools/exp_ts_ba.py) 168 -> 136 us / 742 -> 432 us.
    // Measured (tools/exp_sort_threshold.py, host_test): on the reference's grids (41 k fine cells) the two extra launches pay from ~300
    // rays x 48 on (200 rays: 92 us in ray order, 97 us sorted; 1000 rays: 236 against 162 us); on a small grid (792 cells, 9600 samples)
    // many samples share the few cells, ray order serialises their atomics and sorting wins much earlier (118 against 102 us).
    int key_level = 0;
    for (int q = 0; q < 3; ++q) if (STAGE_DEC[stage][q] >= 0) key_level = std::max(key_level, STAGE_DEC[stage][q] == 3 ? 3 : STAGE_DEC[stage][q]);
    const long cells = (long)(c->grid[key_level].n / 32);
    return M >= 14336 || (cells > 0 && (long)M >= 4 * cells);
}

// sorted: the decoders of this step (forward and the backward that follows) walk the samples cell by cell (see k_sample)
static int stage_key_level(int stage)
{
    int key_level = 0;
    for (int q = 0; q < 3; ++q) if (STAGE_DEC[stage][q] >= 0) key_level = STAGE_DEC[stage][q];      // the finest level the stage reads
    return key_level;
}
static void samp_args(nsk_ctx* c, SampArgs& A, const RParams& R, int stage, int N, int S, const float* ro, const float* rd, const float* gt, float gtmax,
                      const float* gmax_dev, const uint8_t* mask, bool sorted, float* z, int* skey, int* srank, const nsk_ctx::DMax& dm)
{
    const GridState& KG = c->grid[stage_key_level(stage)];
    const GridState* PG = stage >= 2 ? &c->grid[1] : nullptr;      // parent level whose cells order the samples inside a key cell (k_sample)
    const size_t bins = KG.n / 32 * 8;
    memset(&A, 0, sizeof(A));
    A.R = R; A.N = N; A.S = S; A.rays_o = ro; A.rays_d = rd; A.gt_depth = gt; A.gtmax_host = gtmax; A.gtmax_dev = gmax_dev; A.keep = mask; A.z_out = z;
    A.kX = KG.X; A.kY = KG.Y; A.kZ = KG.Z; A.pX = PG ? PG->X : 0; A.pY = PG ? PG->Y : 0; A.pZ = PG ? PG->Z : 0; A.ncell2 = (int)((bins / 8 + 1) / 2);
    A.skey = sorted ? skey : nullptr; A.srank = srank; A.hist = c->ws.hist;
    if (dm.n > 0) { A.mx_gt = dm.gt; A.mx_keep = dm.keep; A.mx_n = dm.n; } else { A.mx_gt = gt; A.mx_keep = mask; A.mx_n = N; }
}
// halves: 256-cell chunks per workgroup (1: k_sort_scan, 2: the role inside k_decode_bwd_multi)
static ScanArgs scan_args(nsk_ctx* c, int stage, int* offs, int halves)
{
    const size_t bins = c->grid[stage_key_level(stage)].n / 32 * 8;
    ScanArgs A; A.nkeys = (int)bins; A.ncell2 = (int)((bins / 8 + 1) / 2); A.hist = c->ws.hist; A.offs = offs;
    A.nblocks = (int)((bins + 2048 * (size_t)halves - 1) / (2048 * (size_t)halves));
    return A;
}
static PlaceArgs place_args(int M, const int* skey, const int* srank, const int* offs, int* perm)
{
    PlaceArgs A; A.M = M; A.skey = skey; A.srank = srank; A.offs = offs; A.perm = perm; A.nblocks = (M + 255) / 256;
    return A;
}
// the batch maximum of gt_depth has to come from a launch of its own (k_sample's waves take it themselves for smaller batches)
static bool needs_depth_max(const float* gt, float gtmax, int N, const nsk_ctx::DMax& dm) { return gt && gtmax < 0.f && (dm.n > 0 ? dm.n : N) > 8192; }

// sampling (+ cell sort) of one batch into the given output set by launches of its own; `done`: stages that have already run (nsk_ctx::Prep)
static int launch_sampling(nsk_ctx* c, const RParams& R, int stage, int N, int S, const float* ro, const float* rd, const float* gt, float gtmax,
                           const uint8_t* mask, bool sorted, float* z, int* skey, int* srank, int* offs, int* perm, const nsk_ctx::DMax& dm, int done = 0)
{
    const int M = N * S;
    hipStream_t st = c->stream;
    if (!(done & 1)) {
        const float* gmax_dev = nullptr;
        if (needs_depth_max(gt, gtmax, N, dm)) {
            ProfScope ps(c, "depth_max");
            if (dm.n > 0) k_depth_max<<<1, 1024, 0, st>>>(dm.n, dm.gt, dm.keep, c->scal);      // the maximum of the batch this call's rays are a shard of
            else k_depth_max<<<1, 1024, 0, st>>>(N, gt, mask, c->scal);
            gmax_dev = c->scal;
        }
        ProfScope ps(c, "sample");
        SampArgs A;
        samp_args(c, A, R, stage, N, S, ro, rd, gt, gtmax, gmax_dev, mask, sorted, z, skey, srank, dm);
        k_sample<<<(N + NSK_SAMPLE_RAYS - 1) / NSK_SAMPLE_RAYS, 64 * NSK_SAMPLE_RAYS, 0, st>>>(A);
    }
    if (sorted && (done & 6) != 6) {
        ProfScope ps(c, "cell_sort");
        if (!(done & 2)) { const ScanArgs A = scan_args(c, stage, offs, 1); k_sort_scan<<<A.nblocks, 256, 0, st>>>(A); }
        if (!(done & 4)) { const PlaceArgs A = place_args(M, skey, srank, offs, perm); k_sort_place<<<A.nblocks, 256, 0, st>>>(A); }
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// the stages of the prepared batch that have not run yet, by launches of their own (its step has come, or something else needs the histogram)
static int prep_finish(nsk_ctx* c)
{
    nsk_ctx::Prep& P = c->prep;
    if (!P.valid) return 0;
    const int all = P.sorted ? 7 : 1;
    if ((P.done & all) == all) return 0;
    Workspace& w = c->ws;
    CHK(launch_sampling(c, P.R, P.stage, P.N, P.S, P.ro, P.rd, P.gt, P.gtmax, P.mask, P.sorted, w.z_alt, w.skey_alt, w.srank_alt, w.offs_alt, w.perm_alt, P.dmax, P.done));
    P.done = all;
    return 0;
}
// forget the prepared batch and any registered one; a histogram that holds its counts is cleared by the scan
static int prep_drop(nsk_ctx* c)
{
    nsk_ctx::Prep& P = c->prep;
    if (P.valid && P.sorted && (P.done & 1) && !(P.done & 2)) { const ScanArgs A = scan_args(c, P.stage, c->ws.offs_alt, 1); k_sort_scan<<<A.nblocks, 256, 0, c->stream>>>(A); HIPCHK(hipGetLastError()); }
    P.valid = false; c->req.valid = false;
    return 0;
}

static size_t stage_bins(nsk_ctx* c, int stage) { return c->grid[stage_key_level(stage)].n / 32 * 8; }

static int forward_core(nsk_ctx* c, int stage, int N, int S, const float* ro, const float* rd, const float* gt, float gtmax, bool save_masks,
                        bool sorted = false)
{
    const int M = N * S;
    const uint8_t* mask = save_masks ? c->ray_mask : nullptr;      // (only the steps that form a loss honour it; a plain render shows every ray)
    auto is_this_batch = [&](const nsk_ctx::Prep& X) {
        return X.valid && save_masks && !c->capturing && X.stage == stage && X.N == N && X.S == S && X.ro == ro && X.rd == rd && X.gt == gt && X.gtmax == gtmax &&
               X.mask == mask && X.sorted == sorted && memcmp(&X.R, &c->R, sizeof(RParams)) == 0 &&
               X.dmax.gt == c->dmax.gt && X.dmax.keep == c->dmax.keep && X.dmax.n == c->dmax.n;
    };
    nsk_ctx::Prep& P = c->prep;
    if (!c->capturing) CHK(prep_finish(c));                 // (also when the set is for another batch: the sampling below needs the cell histogram; nsk_graph_begin has settled it before a capture)
    if (is_this_batch(P)) {
        // this batch was sampled during the previous step (nsk_map_prepare): take its outputs
        Workspace& w = c->ws;
        std::swap(w.z, w.z_alt); std::swap(w.perm, w.perm_alt); std::swap(w.skey, w.skey_alt); std::swap(w.srank, w.srank_alt);
        std::swap(w.offs, w.offs_alt);                      // (both sets are kept at the same capacities: ensure_alt)
        c->ws_flip ^= 1;
        P.valid = false;
        c->sorted = sorted;
    } else {
        if (is_this_batch(c->req)) c->req.valid = false;    // registered, but no step came by to carry its sampling: sampled here like any other batch
        c->sorted = sorted;
        if (sorted) CHK(ensure_hist(c, stage_bins(c, stage)));
        CHK(launch_sampling(c, c->R, stage, N, S, ro, rd, gt, gtmax, mask, sorted, c->ws.z, c->ws.skey, c->ws.srank, c->ws.offs, c->ws.perm, c->dmax));
    }
    CHK(launch_decode_fwd_stage(c, stage, M, S, ro, rd, save_masks));
    c->dbg_M = M; c->dbg_S = S;
    return 0;
}

static void comp_args(nsk_ctx* c, CompArgs& A, int stage, int N, int S, const float* ro, const float* rd)
{
    memset(&A, 0, sizeof(A));
    A.R = c->R; A.N = N; A.S = S; A.stage = stage; A.rays_o = ro; A.rays_d = rd; A.z = c->ws.z;
    A.occ_a = stage == 0 ? c->ws.occ[0] : c->ws.occ[1];
    A.occ_b = stage >= 2 ? c->ws.occ[2] : nullptr;
    A.rgb4 = stage == 3 ? c->ws.rgb4 : nullptr;
    A.g_raw = c->ws.g_raw;
    A.keep = c->ray_mask;
}

static int common_checks(nsk_ctx* c, int stage, int N, const float* ro, const float* rd, int* S_out, const float* gt)
{
    if (!c) return fail("null ctx");
    if (N < 1) return fail("N must be >= 1 (got %d)", N);
    if (!ro || !rd) return fail("rays_o / rays_d is NULL");
    CHK(check_stage(c, stage));
    HIPCHK(hipSetDevice(c->device));
    int S = c->R.n_samples + (gt ? c->R.n_surface : 0);
    if ((long long)N * S >= (1LL << 26)) return fail("N*S too large (%lld samples; at most 2^26 - 1 per call)", (long long)N * S);      // (also the range of ray_of)
    CHK(ensure_ws(c, N, N * S));
    *S_out = S;
    return 0;
}

extern "C" int nsk_render_forward(nsk_ctx* c, int stage, int N, const float* ro, const float* rd, const float* gt, float gtmax,
                                  float* rgb, float* depth, float* var, float* weights)
{
    int S;
    CHK(common_checks(c, stage, N, ro, rd, &S, gt));
    CHK(forward_core(c, stage, N, S, ro, rd, gt, gtmax, false));
    CompArgs A;
    comp_args(c, A, stage, N, S, ro, rd);
    A.rgb = rgb; A.depth = depth; A.var = var; A.weights = weights; A.mode = 0;
    { ProfScope ps(c, "composite"); k_composite<<<(N + 3) / 4, 256, 0, c->stream>>>(A); }
    HIPCHK(hipGetLastError());
    account(c, stage, N * S, N, false, 0);
    return 0;
}

extern "C" int nsk_raw2outputs(nsk_ctx* c, int N, int S, const float* raw, const float* z, const float* rd, int occupancy,
                               float* rgb, float* depth, float* var, float* weights)
{
    if (!c || !raw || !z || !rd || !rgb || !depth || !var) return fail("nsk_raw2outputs: null argument");
    if (N < 1 || S < 1 || S > 64) return fail("nsk_raw2outputs: need N >= 1 and 1 <= S <= 64");
    HIPCHK(hipSetDevice(c->device));
    k_raw2outputs<<<(N + 3) / 4, 256, 0, c->stream>>>(N, S, occupancy, raw, z, rd, rgb, depth, var, weights);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int nsk_eval_points(nsk_ctx* c, int stage, int M, const float* pts, float* raw)
{
    if (!c || !pts || !raw) return fail("nsk_eval_points: null argument");
    if (M < 1) return fail("nsk_eval_points: M must be >= 1");
    CHK(check_stage(c, stage));
    HIPCHK(hipSetDevice(c->device));
    CHK(ensure_ws(c, 1, M));
    for (int q = 0; q < 3; ++q) {
        int w = STAGE_DEC[stage][q];
        if (w < 0) break;
        CHK(launch_decode_fwd(c, w, M, 1, nullptr, nullptr, pts, false));
    }
    k_eval_finish<<<(M + 255) / 256, 256, 0, c->stream>>>(M, stage, pts, c->d_bound, stage == 0 ? c->ws.occ[0] : c->ws.occ[1],
                                                          stage >= 2 ? c->ws.occ[2] : nullptr, stage == 3 ? c->ws.rgb4 : nullptr, raw);
    HIPCHK(hipGetLastError());
    account(c, stage, M, 0, false, 0);
    return 0;
}

// decoders' backward after k_composite wrote g_raw: ONE launch for every decoder of the stage that needs it
// sum the per-workgroup decoder-gradient slabs of the last backward into the gradient slab (normally done inside k_adam_multi)
static int flush_pending(nsk_ctx* c)
{
    if (c->pend_w < 0) return 0;
    const int w = c->pend_w, np = c->dec[w].n, n4 = (np + 3) & ~3;
    ProfScope ps(c, "dec_grad_reduce");
    k_dec_grad_reduce<<<dim3((np + 255) / 256, 8), 256, 0, c->stream>>>(np, n4, c->pend_nb, c->ws.dec_slabs, c->slab + c->dec[w].g_off);
    HIPCHK(hipGetLastError());
    c->pend_w = -1;
    return 0;
}

// dyn_resid: the Tracker's deferred median mask (composite mode 5 wrote the residuals there and the seeds to ws.tmp_rgb): k_decode_bwd_track
static int backward_core(nsk_ctx* c, int stage, int N, int S, const float* ro, const float* rd, unsigned flags, float* g_ro, float* g_rd,
                         float* d_loss = nullptr, const float* dyn_resid = nullptr)
{
    const bool rays = (flags & NSK_GRAD_RAYS) != 0;
    const bool grids = (flags & NSK_GRAD_GRIDS) != 0;
    const int M = N * S;
    CHK(flush_pending(c));                      // gradients accumulate across calls: the slabs are about to be overwritten
    MultiArgs MA;
    memset(&MA, 0, sizeof(MA));
    int n = 0, cost[3], train_role = -1; size_t lds = 0;
    for (int q = 2; q >= 0; --q) {
        int w = STAGE_DEC[stage][q];
        if (w < 0) continue;
        bool train = (flags & NSK_GRAD_DECODERS) && c->dec[w].trainable;
        if (!train && !grids && !rays) continue;
        DecArgs& A = MA.a[n];
        fill_args(c, A, w, M, S, ro, rd, nullptr);
        A.grid = grid_dev(c, w, grids);
        A.masks = c->ws.masks[w];
        A.g_raw = c->ws.g_raw;
        A.g_rays_o = g_ro; A.g_rays_d = g_rd;
        A.dyn_resid = dyn_resid; A.dyn_n = N;
        A.g_dec = train ? c->ws.dec_slabs : c->slab + c->dec[w].g_off;
#ifdef NSK_EXPERIMENT
        A.flags = flags & 0xffffu;                                  // experiment builds pass the debug bits 9.. through (tools/exp_bwd.py)
#else
        A.flags = flags & 0xffu;
#endif
        if (w != 0 && (!train || w != 2)) CHK(ensure_bimg16(c, w));
        if (train && w != 2) {
            if (c->ws.hsave_M[w] != M) return fail("backward of trainable decoder %d: its forward must run with the decoder already trainable (block outputs not saved)", w);
            A.hsave = c->ws.hsave[w];
        }
        MA.which[n] = w; MA.train[n] = train ? 1 : 0;
        // relative cost of one tile of a frozen role against one 8-tile iteration of the trainable role (= 1000)
        const int frozen_cost = rays ? (c->tune_frozen_cost_rays > 0 ? c->tune_frozen_cost_rays : 330) : (c->tune_frozen_cost > 0 ? c->tune_frozen_cost : 205);
        cost[n] = train ? 1000 : (w == 1 ? frozen_cost * c->tune_frozen_mid_pct / 100 : frozen_cost);
        lds = std::max(lds, bwd_lds_bytes(w, train));
        if (train) train_role = (train_role == -1 && w != 2) ? n : -2;     // -2: more than one trainable decoder, or the fine one (its
                                                                            // body is not part of k_decode_bwd_multi) -> separate launches below
        if (train) c->touched[NSK_GROUP_DECODERS] = true;
        if (grids) c->touched[NSK_GROUP_COARSE + w] = true;
        ++n;
    }
    if (dyn_resid && (train_role != -1 || !rays || c->deterministic || c->bwd_mode == 0 || n == 0 || N > NSK_MEDIAN_FUSED_MAX))
        return fail("backward_core: the deferred median mask needs the Tracker's launch (frozen decoders, ray gradients)");      // the caller checked the same
    const int separate = c->deterministic ? 1 : 0;      // debug mode: one launch per decoder, in a fixed order
    if ((n == 0 || train_role == -2 || separate) && d_loss) { ProfScope ps(c, "loss_sum"); k_sum<<<1, 1024, 0, c->stream>>>(N, c->ws.ray_loss, d_loss); }
    if (n == 0) return 0;
    if (train_role == -2 || separate) {      // rare configuration (several trainable decoders share one slab buffer): one launch each
        for (int r = 0; r < n; ++r)
            CHK(launch_decode_bwd(c, MA.which[r], M, S, ro, rd, MA.train[r] != 0, rays, flags, g_ro, g_rd));
        return 0;
    }
    MA.n = n;
    // no trainable role and no ray gradients (the fine / middle / coarse stages of the Mapper): the 16-wave frozen kernel (nsk_device.h); with ray
    // gradients the frozen bodies need 160 VGPRs (the Tracker: 600 tiles, latency-bound either way) and stay in k_decode_bwd_multi
    const bool full = c->bwd_mode == 0;                 // every chain on the fp32 MFMA: k_decode_bwd_multi_full, nothing rides
    const bool frozen_only = train_role == -1 && !rays && !c->tune_no_frozen_kernel && !full;
    if (train_role >= 0 && n > 1) split_wgs_train(c->num_cu, (M + 15) / 16, n, cost, train_role, MA.wg_end);
    else split_wgs_balanced(c->num_cu, (M + 15) / 16, n, cost, MA.wg_end, frozen_only ? NSK_FROZEN_NW : 8);
    const int extra = d_loss ? 1 : 0;          // one more workgroup sums the per-ray losses written by k_composite
    if (d_loss) { MA.sum_src = c->ws.ray_loss; MA.sum_dst = d_loss; MA.sum_n = N; }
    // the prepared batch's cell-sort offsets ride behind the roles (nsk_map_prepare): short workgroups that wait for nothing of this launch
    int scan_wgs = 0;
    {
        nsk_ctx::Prep& P = c->prep;
        if (P.valid && P.sorted && (P.done & 3) == 1 && !rays && !full && !c->capturing && !c->tune_no_piggyback) {
            MA.scan = scan_args(c, P.stage, c->ws.offs_alt, frozen_only ? NSK_FROZEN_NW / 4 : 2);
            scan_wgs = MA.scan.nblocks;
            P.done |= 2;
        }
    }
    {
        ProfScope ps(c, "decode_bwd_multi");
        if (dyn_resid) {
            MA.dyn_seed = c->ws.tmp_rgb; MA.dyn_thr_out = c->scal + 1;
            k_decode_bwd_track<<<MA.wg_end[n - 1] + 1, 512, lds + NSK_DYN_LDS_BYTES, c->stream>>>(MA);
        } else if (full) {
            size_t ldsf = 0;
            for (int r = 0; r < n; ++r) {
                const int w = MA.which[r];
                ldsf = std::max(ldsf, MA.train[r] ? (bwd_img_floats(w) + PN_FLOATS(2)) * 4 : bwd_img_floats(w) * 4 + 8 * 3840);
            }
            if (rays) k_decode_bwd_multi_full<true><<<MA.wg_end[n - 1] + extra, 512, ldsf, c->stream>>>(MA);
            else k_decode_bwd_multi_full<false><<<MA.wg_end[n - 1] + extra, 512, ldsf, c->stream>>>(MA);
        } else if (frozen_only) {
            const size_t lds16 = lds - 8 * 3840 + (size_t)NSK_FROZEN_NW * 3840;      // image + one scatter scratch per wave
            k_decode_bwd_frozen<false><<<MA.wg_end[n - 1] + scan_wgs + extra, 64 * NSK_FROZEN_NW, lds16, c->stream>>>(MA);
        } else if (rays) k_decode_bwd_multi<true><<<MA.wg_end[n - 1] + extra, 512, lds, c->stream>>>(MA);
        else k_decode_bwd_multi<false><<<MA.wg_end[n - 1] + scan_wgs + extra, 512, lds, c->stream>>>(MA);
    }
    HIPCHK(hipGetLastError());
    if (train_role >= 0) {
        int w = MA.which[train_role];
        int nb = MA.wg_end[train_role] - (train_role == 0 ? 0 : MA.wg_end[train_role - 1]);
        int np = c->dec[w].n, n4 = (np + 3) & ~3;
        (void)np; (void)n4;
        c->pend_w = w; c->pend_nb = nb;          // summed by k_adam_multi, or by flush_pending when someone reads the slab first
    }
    return 0;
}

extern "C" int nsk_render_backward(nsk_ctx* c, int stage, int N, const float* ro, const float* rd, const float* gt, float gtmax,
                                   const float* g_rgb, const float* g_depth, const float* g_var, unsigned flags, float* g_ro, float* g_rd)
{
    int S;
    CHK(common_checks(c, stage, N, ro, rd, &S, gt));
    if (!g_rgb || !g_depth) return fail("nsk_render_backward: g_rgb / g_depth is NULL");
    if ((flags & NSK_GRAD_RAYS) && (!g_ro || !g_rd)) return fail("nsk_render_backward: NSK_GRAD_RAYS needs g_rays_o and g_rays_d");
    CHK(forward_core(c, stage, N, S, ro, rd, gt, gtmax, true, sort_pays(c, stage, N * S, flags)));
    CompArgs A;
    comp_args(c, A, stage, N, S, ro, rd);
    A.mode = 1; A.g_rgb = g_rgb; A.g_depth = g_depth; A.g_var = g_var;
    if (flags & NSK_GRAD_RAYS) { A.g_rays_o = g_ro; A.g_rays_d = g_rd; }
    { ProfScope ps(c, "composite"); k_composite<<<(N + 3) / 4, 256, 0, c->stream>>>(A); }
    HIPCHK(hipGetLastError());
    CHK(backward_core(c, stage, N, S, ro, rd, flags, g_ro, g_rd));
    account(c, stage, N * S, N, true, flags);
    return 0;
}

// second set of sampling outputs, kept at the capacities of the first
static int ensure_alt(nsk_ctx* c, bool need_offs)
{
    Workspace& w = c->ws;
    if (w.alt_capM < w.capM) {
        HIPCHK(hipStreamSynchronize(c->stream));
        hipFree(w.z_alt); hipFree(w.perm_alt); hipFree(w.skey_alt); hipFree(w.srank_alt);
        const size_t m = (size_t)w.capM + 64;
        HIPCHK(hipMalloc(&w.z_alt, m * 4)); HIPCHK(hipMalloc(&w.perm_alt, m * 4)); HIPCHK(hipMalloc(&w.skey_alt, m * 4)); HIPCHK(hipMalloc(&w.srank_alt, m * 4));
        w.alt_capM = w.capM;
    }
    if (need_offs && w.alt_bins < w.hist_cap) {
        HIPCHK(hipStreamSynchronize(c->stream));
        hipFree(w.offs_alt);
        HIPCHK(hipMalloc(&w.offs_alt, w.hist_cap * 4));
        w.alt_bins = w.hist_cap;
    }
    return 0;
}

// Registers the NEXT batch.  Nothing is launched here: the nsk_map_step that follows carries the batch's sampling in its composite launch, its
// backward launch carries the offsets of the cell sort, the nsk_adam_step after it the placement -- three launches and their dependent round
// trips (30 us at 5000 rays, 20 us at 1000) leave the front of the next step.  Until round 3 the sampling ran on a side stream beside the
// exchange and the optimiser step: that needed two cross-stream waits per step and lost on one GPU (K3 0.511 against 0.500 ms).
// Whatever has not been carried when the batch's own step arrives is launched there, as for an unprepared batch.
extern "C" int nsk_map_prepare(nsk_ctx* c, int stage, int N, const float* ro, const float* rd, const float* gt, float gtmax, unsigned flags)
{
    int S;
    if (!gt) return fail("nsk_map_prepare: gt_depth is NULL");
    if (c && c->capturing) return fail("nsk_map_prepare: not inside a graph capture");
    CHK(common_checks(c, stage, N, ro, rd, &S, gt));
    const bool sorted = sort_pays(c, stage, N * S, flags);
    if (sorted) CHK(ensure_hist(c, stage_bins(c, stage)));
    CHK(ensure_alt(c, sorted));
    nsk_ctx::Prep& P = c->req;
    P.valid = true; P.stage = stage; P.N = N; P.S = S; P.ro = ro; P.rd = rd; P.gt = gt; P.gtmax = gtmax; P.mask = c->ray_mask; P.sorted = sorted; P.done = 0; P.R = c->R; P.dmax = c->dmax;
    return 0;
}

// a registered batch becomes the prepared one (the second set is free: the step that was using it has swapped it out); ride: its sampling may
// go into this step's composite launch
static int prep_promote(nsk_ctx* c, bool* ride)
{
    *ride = false;
    if (!c->req.valid || c->capturing) return 0;
    if (c->prep.valid) { const nsk_ctx::Prep keep = c->req; CHK(prep_drop(c)); c->req = keep; }      // an unclaimed set makes room
    const nsk_ctx::Prep R = c->req;
    if (R.sorted) CHK(ensure_hist(c, stage_bins(c, R.stage)));
    CHK(ensure_alt(c, R.sorted));
    c->prep = R; c->prep.done = 0; c->req.valid = false;
    *ride = !c->tune_no_piggyback && !needs_depth_max(R.gt, R.gtmax, R.N, R.dmax);
    return 0;
}

extern "C" int nsk_map_step(nsk_ctx* c, int stage, int N, const float* ro, const float* rd, const float* gt, const float* gtc,
                            float gtmax, float w_color, int use_color, unsigned flags, float* d_loss, float* rgb, float* depth,
                            float* var, float* g_ro, float* g_rd)
{
    int S;
    if (!gt) return fail("nsk_map_step: gt_depth is NULL");
    if (use_color && !gtc) return fail("nsk_map_step: use_color needs gt_color");
    CHK(common_checks(c, stage, N, ro, rd, &S, gt));
    if ((flags & NSK_GRAD_RAYS) && (!g_ro || !g_rd)) return fail("nsk_map_step: NSK_GRAD_RAYS needs g_rays_o and g_rays_d");
    CHK(forward_core(c, stage, N, S, ro, rd, gt, gtmax, true, sort_pays(c, stage, N * S, flags)));
    CompArgs A;
    comp_args(c, A, stage, N, S, ro, rd);
    A.mode = 2; A.gt_depth = gt; A.gt_color = gtc; A.w_color = w_color; A.use_color = use_color;
    A.rgb = rgb; A.depth = depth; A.var = var; A.loss = c->ws.ray_loss;
    if (flags & NSK_GRAD_RAYS) { A.g_rays_o = g_ro; A.g_rays_d = g_rd; }
    bool ride = false;
    CHK(prep_promote(c, &ride));
    if (ride) {          // the next batch's sampling behind this batch's compositing, one launch (k_composite_sample)
        nsk_ctx::Prep& P = c->prep;
        SampArgs SA;
        samp_args(c, SA, P.R, P.stage, P.N, P.S, P.ro, P.rd, P.gt, P.gtmax, nullptr, P.mask, P.sorted, c->ws.z_alt, c->ws.skey_alt, c->ws.srank_alt, P.dmax);
        const int cb = (N + 7) / 8, sb = (P.N + NSK_SAMPLE_RAYS - 1) / NSK_SAMPLE_RAYS;
        { ProfScope ps(c, "composite"); k_composite_sample<<<cb + sb, 512, 0, c->stream>>>(A, SA, cb); }
        P.done |= 1;
    } else {
        ProfScope ps(c, "composite"); k_composite<<<(N + 3) / 4, 256, 0, c->stream>>>(A);
    }
    HIPCHK(hipGetLastError());
    CHK(backward_core(c, stage, N, S, ro, rd, flags, g_ro, g_rd, d_loss));
    account(c, stage, N * S, N, true, flags);
    return 0;
}

static int median_thr(nsk_ctx* c, int N, const float* gt, const float* depth)
{
    int P2 = 1; while (P2 < N) P2 <<= 1;
    if (P2 > 16384) return fail("handle_dynamic median supports at most 16384 rays (got %d)", N);
    k_median_thr<<<1, 1024, P2 * 4, c->stream>>>(N, P2, gt, depth, c->ray_mask, c->scal + 1);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int nsk_track_step(nsk_ctx* c, int stage, int N, const float* ro, const float* rd, const float* gt, const float* gtc,
                              float gtmax, float w_color, int use_color, int handle_dynamic, int detach_var, unsigned flags,
                              float* d_loss, float* g_ro, float* g_rd)
{
    int S;
    if (!gt) return fail("nsk_track_step: gt_depth is NULL");
    if (use_color && !gtc) return fail("nsk_track_step: use_color needs gt_color");
    CHK(common_checks(c, stage, N, ro, rd, &S, gt));
    if ((flags & NSK_GRAD_RAYS) && (!g_ro || !g_rd)) return fail("nsk_track_step: NSK_GRAD_RAYS needs g_rays_o and g_rays_d");
    CHK(forward_core(c, stage, N, S, ro, rd, gt, gtmax, true, sort_pays(c, stage, N * S, flags)));
    CompArgs A;
    comp_args(c, A, stage, N, S, ro, rd);
    // One launch (k_composite mode 4: residuals -> grid barrier -> median -> loss and backward) while the grid is at most one
    // workgroup per CU, i.e. certainly resident as a whole; three launches otherwise (and in the deterministic debug mode)
    // (round 3 tried ONE 16-wave workgroup for up to 256 rays -- render, select the median in LDS, render again, no grid barrier: 70 us at 200 rays
    // against 21, a wave's rays run one after the other at ~2.5 us each)
    const bool fused_median = handle_dynamic && !c->deterministic && !c->tune_no_fused_median && N <= NSK_MEDIAN_FUSED_MAX && (N + 3) / 4 <= c->num_cu;
    // Round 4: with ray gradients and frozen decoders (the Tracker as the reference runs it) not even that barrier -- the mask is one factor per
    // ray on what the compositing hands on, and the backward's workgroups apply it (composite mode 5, k_decode_bwd_track)
    bool deferred = handle_dynamic && !c->deterministic && !c->tune_no_fused_median && !c->tune_no_deferred_median && N <= NSK_MEDIAN_FUSED_MAX &&
                    (flags & NSK_GRAD_RAYS) && c->bwd_mode != 0;
    for (int q = 0; q < 3 && deferred; ++q) {
        const int w = STAGE_DEC[stage][q];
        if (w >= 0 && (flags & NSK_GRAD_DECODERS) && c->dec[w].trainable) deferred = false;
    }
    if (handle_dynamic && !fused_median && !deferred) {                     // forward pass for the median (Tracker.cpp:69-70)
        A.mode = 0; A.depth = c->ws.tmp_depth;
        k_composite<<<(N + 3) / 4, 256, 0, c->stream>>>(A);
        CHK(median_thr(c, N, gt, c->ws.tmp_depth));
        A.depth = nullptr;
    }
    A.mode = deferred ? 5 : (fused_median ? 4 : 3); A.gt_depth = gt; A.gt_color = gtc; A.w_color = w_color; A.use_color = use_color;
    A.thr = c->scal + 1; A.handle_dynamic = handle_dynamic; A.detach_var = detach_var; A.loss = c->ws.ray_loss;
    if (deferred) { A.resid = c->ws.tmp_depth; A.g_seed = c->ws.tmp_rgb; }
    else if (fused_median) { c->median_fused_pending = true; A.resid = c->ws.tmp_depth; A.bar = reinterpret_cast<unsigned*>(c->scal + 5); A.thr_out = c->scal + 1; }
    if (flags & NSK_GRAD_RAYS) { A.g_rays_o = g_ro; A.g_rays_d = g_rd; }
    { ProfScope ps(c, "composite"); k_composite<<<(N + 3) / 4, 256, 0, c->stream>>>(A); }
    HIPCHK(hipGetLastError());
    CHK(backward_core(c, stage, N, S, ro, rd, flags, g_ro, g_rd, d_loss, deferred ? c->ws.tmp_depth : nullptr));
    account(c, stage, N * S, N, true, flags);
    return 0;
}

extern "C" int nsk_loss_map(nsk_ctx* c, int N, const float* depth, const float* rgb, const float* gt, const float* gtc, float w_color,
                            int use_color, float* g_depth, float* g_rgb, float* d_loss)
{
    if (!c || !depth || !rgb || !gt || !gtc || !g_depth || !g_rgb) return fail("nsk_loss_map: null argument");
    if (N < 1) return fail("nsk_loss_map: N must be >= 1");
    HIPCHK(hipSetDevice(c->device));
    CHK(ensure_ws(c, N, 1));
    k_loss_map<<<(N + 255) / 256, 256, 0, c->stream>>>(N, depth, rgb, gt, gtc, w_color, use_color, g_depth, g_rgb, c->ws.ray_loss);
    if (d_loss) k_sum<<<1, 1024, 0, c->stream>>>(N, c->ws.ray_loss, d_loss);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int nsk_loss_track(nsk_ctx* c, int N, const float* depth, const float* rgb, const float* var, const float* gt,
                              const float* gtc, float w_color, int use_color, int handle_dynamic, int detach_var, float* g_depth,
                              float* g_rgb, float* g_var, float* d_loss)
{
    if (!c || !depth || !rgb || !var || !gt || !gtc || !g_depth || !g_rgb) return fail("nsk_loss_track: null argument");
    if (N < 1) return fail("nsk_loss_track: N must be >= 1");
    HIPCHK(hipSetDevice(c->device));
    CHK(ensure_ws(c, N, 1));
    if (handle_dynamic) CHK(median_thr(c, N, gt, depth));
    k_loss_track<<<(N + 255) / 256, 256, 0, c->stream>>>(N, depth, rgb, var, gt, gtc, w_color, use_color, handle_dynamic, detach_var,
                                                         c->scal + 1, g_depth, g_rgb, g_var, c->ws.ray_loss);
    if (d_loss) k_sum<<<1, 1024, 0, c->stream>>>(N, c->ws.ray_loss, d_loss);
    HIPCHK(hipGetLastError());
    return 0;
}

// ---- rays / pose --------------------------------------------------------------------------------------------
static void intr(int mode, float& fx, float& fy, float& cx, float& cy)
{
    if (mode & 2) { fx = (float)(int)fx; fy = (float)(int)fy; cx = (float)(int)cx; cy = (float)(int)cy; }   // D10
}
static void invert4(const float* m, float* inv)          // Gauss-Jordan with partial pivoting, in double
{
    double a[4][8];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { a[i][j] = m[4 * i + j]; a[i][4 + j] = i == j; }
    for (int c = 0; c < 4; ++c) {
        int p = c;
        for (int r = c + 1; r < 4; ++r) if (fabs(a[r][c]) > fabs(a[p][c])) p = r;
        for (int j = 0; j < 8; ++j) std::swap(a[c][j], a[p][j]);
        double d = a[c][c];
        for (int j = 0; j < 8; ++j) a[c][j] /= d;
        for (int r = 0; r < 4; ++r) if (r != c) { double f = a[r][c]; for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j]; }
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) inv[4 * i + j] = (float)a[i][4 + j];
}

// ---- test aids: what the last step's forward decided at every ReLU, and the numbers it decided on ----------------------------------
// (read-only views of the workspace; nothing in the product path calls them.  tests/test_gpu_relu.py, tools/relu_flips.py)
extern "C" int nsk_debug_relu_bits(nsk_ctx* c, int which, int M, uint8_t* h_bits /*[M][5][32] by sample*/)
{
    if (!c || !h_bits) return fail("nsk_debug_relu_bits: null argument");
    if (which < 1 || which > 3) return fail("nsk_debug_relu_bits: decoder 1..3");
    if (M < 1 || M != c->dbg_M || M > c->ws.capM) return fail("nsk_debug_relu_bits: M = %d is not the last step's sample count (%d)", M, c->dbg_M);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    std::vector<unsigned long long> mk((size_t)M * 4);
    std::vector<int> perm;
    HIPCHK(hipMemcpy(mk.data(), c->ws.masks[which], mk.size() * 8, hipMemcpyDeviceToHost));
    if (c->sorted) { perm.resize(M); HIPCHK(hipMemcpy(perm.data(), c->ws.perm, (size_t)M * 4, hipMemcpyDeviceToHost)); }
    for (int slot = 0; slot < M; ++slot) {
        const int m = c->sorted ? perm[slot] : slot;
        if (m < 0 || m >= M) return fail("nsk_debug_relu_bits: perm[%d] = %d out of range", slot, m);
        uint8_t* row = h_bits + (size_t)m * 160;
        for (int g = 0; g < 4; ++g) {
            const unsigned long long w = mk[(size_t)slot * 4 + g];
            for (int l = 0; l < 5; ++l)
                for (int r = 0; r < 2; ++r)
                    for (int i = 0; i < 4; ++i) row[32 * l + 16 * r + 4 * g + i] = (uint8_t)((w >> (8 * l + 4 * r + i)) & 1ull);
        }
    }
    return 0;
}
// per-sample arrays of the last step's workspace, by sample: what = 0..2 occupancy of decoder 0..2 [M], 3 colour decoder output [M][4], 4 g_raw [M][4], 5 z [M]
extern "C" int nsk_debug_fetch(nsk_ctx* c, int what, int M, float* h_out)
{
    if (!c || !h_out) return fail("nsk_debug_fetch: null argument");
    if (M < 1 || M != c->dbg_M || M > c->ws.capM) return fail("nsk_debug_fetch: M = %d is not the last step's sample count (%d)", M, c->dbg_M);
    const float* src = what >= 0 && what <= 2 ? c->ws.occ[what] : (what == 3 ? c->ws.rgb4 : (what == 4 ? c->ws.g_raw : (what == 5 ? c->ws.z : nullptr)));
    if (!src) return fail("nsk_debug_fetch: what = 0..5");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(h_out, src, (size_t)M * ((what == 3 || what == 4) ? 16 : 4), hipMemcpyDeviceToHost));
    return 0;
}
// the ReLU inputs of decoder `which` over the samples of the last step (same rays), by the forward body of the current matmul mode
extern "C" int nsk_debug_preact(nsk_ctx* c, int which, int N, const float* ro, const float* rd, float* d_out /*[M][5][32] device*/)
{
    if (!c || !ro || !rd || !d_out) return fail("nsk_debug_preact: null argument");
    if (which < 1 || which > 3) return fail("nsk_debug_preact: decoder 1..3");
    const int S = c->dbg_S, M = c->dbg_M;
    if (S < 1 || N * S != M) return fail("nsk_debug_preact: N = %d does not match the last step (%d samples, %d per ray)", N, M, S);
    HIPCHK(hipSetDevice(c->device));
    DecArgs A;
    fill_args(c, A, which, M, S, ro, rd, nullptr);
    A.dump = d_out;
    const int grid = std::max(1, std::min(((M + 15) / 16 + 7) / 8, c->num_cu));
    if (c->matmul_mode == 0) k_decode_fwd_dump<0><<<grid, 512, fwd_img_floats(which) * 4, c->stream>>>(A, which);
    else if (c->matmul_mode == 1) k_decode_fwd_dump<1><<<grid, 512, (size_t)c->dec[which].fimg16_f * 4, c->stream>>>(A, which);
    else k_decode_fwd_dump<2><<<grid, 512, (size_t)c->dec[which].fimg16_f * 4, c->stream>>>(A, which);
    HIPCHK(hipGetLastError());
    return 0;
}

#ifdef NSK_EXPERIMENT
extern "C" int nsk_dbg_set(nsk_ctx* c, int flags)
{
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(nsk_dbg_flags), &flags, sizeof(int)));
    return 0;
}
extern "C" int nsk_dbg_read_ph(nsk_ctx* c, unsigned long long* out)      // [8][8][96]
{
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(nsk_dbg_ph), sizeof(unsigned long long) * 8 * 8 * 96));
    return 0;
}
extern "C" int nsk_dbg_read_oob(nsk_ctx* c, unsigned* out, int clear)      // [8]: out-of-range index counts by site (NSK_IDX, nsk_device.h)
{
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(nsk_dbg_oob), sizeof(unsigned) * 8));
    if (clear) { unsigned z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(nsk_dbg_oob), z, sizeof(z))); }
    return 0;
}
extern "C" int nsk_dbg_read_ts(nsk_ctx* c, unsigned long long* out)      // [2][1024][4]
{
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(nsk_dbg_ts), sizeof(unsigned long long) * 2 * 1024 * 4));
    return 0;
}
#endif

extern "C" int nsk_frustum_mask(nsk_ctx* c, int level, const float* d_depth, int H, int W, float fx, float fy, float cx, float cy,
                                const float h_c2w[16], uint8_t* h_mask_out)
{
    if (!c || !which_ok(level) || !d_depth || !h_c2w) return fail("nsk_frustum_mask: bad argument");
    GridState& G = c->grid[level];
    if (!G.n) return fail("nsk_frustum_mask: grid level %d not uploaded", level);
    HIPCHK(hipSetDevice(c->device));
    const size_t nvox = G.n / 32;
    G.midx_dirty = true;
    if (c->capturing) return fail("nsk_frustum_mask: not while a graph is being captured");
    bool live = false;
    for (auto& R : c->graphs) live = live || !R.stale;
    if (live) { HIPCHK(hipStreamSynchronize(c->stream)); invalidate_graphs(c); }      // new mask contents: see nsk_set_mask
    if (c->slab) HIPCHK(hipMemsetAsync(c->slab + G.g_off, 0, G.n * 4, c->stream));       // see nsk_set_mask
    if (!G.mask) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipMalloc(&G.mask, nvox)); }
    if (level == NSK_COARSE) {                                   // src/Mapper.cpp:54-59
        HIPCHK(hipMemsetAsync(G.mask, 1, nvox, c->stream));
    } else {
        if (nvox > c->fr_cap) {
            HIPCHK(hipStreamSynchronize(c->stream));
            hipFree(c->fr_tmp);
            HIPCHK(hipMalloc(&c->fr_tmp, nvox * 9 + 16));
            c->fr_cap = nvox;
        }
        float* dep = reinterpret_cast<float*>(c->fr_tmp); float* zz = dep + nvox;
        uint8_t* inimg = reinterpret_cast<uint8_t*>(zz + nvox);
        unsigned int* dmax = reinterpret_cast<unsigned int*>(c->scal + 4);
        FrustumArgs A;
        memcpy(A.bound, c->R.bound, sizeof(A.bound));
        invert4(h_c2w, A.w2c);
        A.cam[0] = h_c2w[3]; A.cam[1] = h_c2w[7]; A.cam[2] = h_c2w[11];
        A.Z = G.Z; A.Y = G.Y; A.X = G.X; A.H = H; A.W = W; A.fx = fx; A.fy = fy; A.cx = cx; A.cy = cy;
        HIPCHK(hipMemsetAsync(dmax, 0, 4, c->stream));
        int blocks = (int)((nvox + 255) / 256);
        k_frustum_pass1<<<blocks, 256, 0, c->stream>>>(A, d_depth, dep, zz, inimg, dmax);
        k_frustum_pass2<<<blocks, 256, 0, c->stream>>>(A, dep, zz, inimg, dmax, G.mask);
        HIPCHK(hipGetLastError());
    }
    if (h_mask_out) {
        HIPCHK(hipMemcpyAsync(h_mask_out, G.mask, nvox, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return 0;
}

extern "C" int nsk_keyframe_overlap(nsk_ctx* c, int N, const float* d_ro, const float* d_rd, const float* d_gt_depth, int n_samples,
                                    int H, int W, float fx, float fy, float cx, float cy, int K, const float* h_c2w, float* h_percent)
{
    if (!c || !d_ro || !d_rd || !d_gt_depth || !h_c2w || !h_percent) return fail("nsk_keyframe_overlap: null argument");
    if (N <= 0 || n_samples <= 0 || K < 0) return fail("nsk_keyframe_overlap: bad sizes");
    if (K == 0) return 0;
    HIPCHK(hipSetDevice(c->device));
    std::vector<float> w2c((size_t)K * 16);
    for (int k = 0; k < K; ++k) invert4(h_c2w + 16 * k, w2c.data() + 16 * k);
    const size_t need = (size_t)K * 17 * 4;
    if (need > c->fr_cap * 9 + 16 || !c->fr_tmp) {
        HIPCHK(hipStreamSynchronize(c->stream));
        hipFree(c->fr_tmp);
        c->fr_cap = (need + 8) / 9;
        HIPCHK(hipMalloc(&c->fr_tmp, c->fr_cap * 9 + 16));
    }
    float* d_w2c = reinterpret_cast<float*>(c->fr_tmp); float* d_pct = d_w2c + (size_t)K * 16;
    HIPCHK(hipMemcpyAsync(d_w2c, w2c.data(), (size_t)K * 64, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));         // w2c is a stack-lifetime host buffer
    OverlapArgs A; A.N = N; A.ns = n_samples; A.H = H; A.W = W; A.K = K; A.fx = fx; A.fy = fy; A.cx = cx; A.cy = cy;
    k_keyframe_overlap<<<K, 256, 0, c->stream>>>(A, d_ro, d_rd, d_gt_depth, d_w2c, d_pct);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(h_percent, d_pct, (size_t)K * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int nsk_sample_pixels(nsk_ctx* c, unsigned long long seed, int n, int H0, int H1, int W0, int W1, int32_t* d_pix_i, int32_t* d_pix_j)
{
    if (!c || !d_pix_i || !d_pix_j) return fail("nsk_sample_pixels: null argument");
    if (n < 0 || H1 <= H0 || W1 <= W0) return fail("nsk_sample_pixels: empty window [%d,%d) x [%d,%d)", H0, H1, W0, W1);
    if (n == 0) return 0;
    HIPCHK(hipSetDevice(c->device));
    k_sample_pixels<<<(n + 255) / 256, 256, 0, c->stream>>>(seed, n, H0, W0, W1 - W0, (unsigned long long)(H1 - H0) * (unsigned long long)(W1 - W0), d_pix_i, d_pix_j);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int nsk_gather_pixels(nsk_ctx* c, int n, const int32_t* d_pix_i, const int32_t* d_pix_j, int H, int W, const float* d_depth,
                                 const float* d_color, float* d_gt_depth, float* d_gt_color)
{
    if (!c || !d_pix_i || !d_pix_j || !d_depth || !d_gt_depth) return fail("nsk_gather_pixels: null argument");
    if (n < 0 || H <= 0 || W <= 0) return fail("nsk_gather_pixels: bad sizes");
    if (n == 0) return 0;
    HIPCHK(hipSetDevice(c->device));
    k_gather_pixels<<<(n + 255) / 256, 256, 0, c->stream>>>(n, d_pix_i, d_pix_j, W, d_depth, d_color, d_gt_depth, d_gt_color);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int nsk_rays_from_pixels(nsk_ctx* c, int n, const int32_t* pi, const int32_t* pj, float fx, float fy, float cx, float cy,
                                    const float* c2w, int mode, float* ro, float* rd)
{
    if (!c || !pi || !pj || !c2w || !ro || !rd || n < 1) return fail("nsk_rays_from_pixels: bad argument");
    intr(mode, fx, fy, cx, cy);
    k_rays_from_pixels<<<(n + 255) / 256, 256, 0, c->stream>>>(n, pi, pj, fx, fy, cx, cy, c2w, mode, ro, rd);
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int nsk_rays_backward(nsk_ctx* c, int n, const int32_t* pi, const int32_t* pj, float fx, float fy, float cx, float cy, int mode,
                                 const float* g_ro, const float* g_rd, float* g_c2w)
{
    if (!c || !pi || !pj || !g_ro || !g_rd || !g_c2w || n < 1) return fail("nsk_rays_backward: bad argument");
    intr(mode, fx, fy, cx, cy);
    k_rays_backward<<<1, 256, 0, c->stream>>>(n, pi, pj, fx, fy, cx, cy, mode, g_ro, g_rd, g_c2w);
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int nsk_camera_from_tensor(nsk_ctx* c, const float* cam, float* c2w)
{
    if (!c || !cam || !c2w) return fail("nsk_camera_from_tensor: null argument");
    k_camera_from_tensor<<<1, 1, 0, c->stream>>>(cam, c2w);
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int nsk_camera_backward(nsk_ctx* c, const float* cam, const float* g_c2w, float* g_cam)
{
    if (!c || !cam || !g_c2w || !g_cam) return fail("nsk_camera_backward: null argument");
    k_camera_backward<<<1, 1, 0, c->stream>>>(cam, g_c2w, g_cam);
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int nsk_rays_from_camera(nsk_ctx* c, int n, const int32_t* pi, const int32_t* pj, float fx, float fy, float cx, float cy,
                                    const float* d_cam, int mode, float* ro, float* rd, float* d_c2w_out)
{
    if (!c || !pi || !pj || !d_cam || !ro || !rd || n < 1) return fail("nsk_rays_from_camera: bad argument");
    intr(mode, fx, fy, cx, cy);
    k_rays_from_camera<<<(n + 255) / 256, 256, 0, c->stream>>>(n, pi, pj, fx, fy, cx, cy, d_cam, mode, ro, rd, d_c2w_out);
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int nsk_pose_step(nsk_ctx* c, int n, const int32_t* pi, const int32_t* pj, float fx, float fy, float cx, float cy, int mode,
                             const float* g_ro, const float* g_rd, float* d_cam, float* d_m, float* d_v, float lr, float b1, float b2, float eps,
                             int step, float* d_g_cam_out)
{
    if (!c || !pi || !pj || !g_ro || !g_rd || !d_cam || !d_m || !d_v || n < 1 || step < 1) return fail("nsk_pose_step: bad argument");
    intr(mode, fx, fy, cx, cy);
    float ss, bc2s; adam_consts(lr, b1, b2, step, ss, bc2s);
    k_pose_step<<<1, 256, 0, c->stream>>>(n, pi, pj, fx, fy, cx, cy, mode, g_ro, g_rd, d_cam, d_m, d_v, ss, bc2s, b1, b2, eps, d_g_cam_out);
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int nsk_pose_step_multi(nsk_ctx* c, int nframes, const int* h_first, const int* h_count, const uint8_t* h_active, const int32_t* pi, const int32_t* pj,
                                   float fx, float fy, float cx, float cy, int mode, const float* g_ro, const float* g_rd, float* d_cams, float* d_m, float* d_v,
                                   float lr, float b1, float b2, float eps, int step, float* d_g_cams, const uint8_t* d_keep, int n_keep)
{
    if (!c || !h_first || !h_count || !h_active || !pi || !pj || !g_ro || !g_rd || !d_cams) return fail("nsk_pose_step_multi: null argument");
    if (nframes < 1 || nframes > NSK_MAX_POSE_FRAMES) return fail("nsk_pose_step_multi: 1 <= nframes <= %d (got %d)", NSK_MAX_POSE_FRAMES, nframes);
    if (step < 0 || (step > 0 && (!d_m || !d_v))) return fail("nsk_pose_step_multi: step >= 1 needs the Adam moments (step 0 = gradients only)");
    if (step == 0 && !d_g_cams) return fail("nsk_pose_step_multi: step 0 (gradients only) needs d_g_cams");
    HIPCHK(hipSetDevice(c->device));
    PoseFrames F; memset(&F, 0, sizeof(F));
    for (int f = 0; f < nframes; ++f) {
        if (h_first[f] < 0 || h_count[f] < 0) return fail("nsk_pose_step_multi: frame %d has a negative ray range", f);
        F.first[f] = h_first[f]; F.count[f] = h_count[f]; F.active[f] = h_active[f] ? 1 : 0;
    }
    intr(mode, fx, fy, cx, cy);
    float ss = 0.f, bc2s = 1.f;
    if (step > 0) adam_consts(lr, b1, b2, step, ss, bc2s);
    const int count_block = (d_g_cams && n_keep > 0) ? 1 : 0;
    ProfScope ps(c, "pose_multi");
    k_pose_multi<<<nframes + count_block, 256, 0, c->stream>>>(F, nframes, pi, pj, fx, fy, cx, cy, mode, g_ro, g_rd, d_cams, d_m, d_v, ss, bc2s, b1, b2, eps,
                                                             step > 0 ? 1 : 0, d_g_cams, d_keep, n_keep);
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int nsk_set_depth_max_batch(nsk_ctx* c, const float* d_gt_depth, const uint8_t* d_keep, int n)
{
    if (!c) return fail("null ctx");
    if (n < 0 || (n > 0 && !d_gt_depth)) return fail("nsk_set_depth_max_batch: bad argument");
    c->dmax.gt = n > 0 ? d_gt_depth : nullptr; c->dmax.keep = n > 0 ? d_keep : nullptr; c->dmax.n = n;
    return 0;
}
extern "C" int nsk_grad_extra(nsk_ctx* c, float* d_buf, size_t n_floats)
{
    if (!c) return fail("null ctx");
    if (n_floats % 4 != 0 || (n_floats > 0 && !d_buf)) return fail("nsk_grad_extra: a device buffer of a multiple of 4 floats (or 0 floats to remove it)");
    if (((uintptr_t)d_buf & 15) != 0) return fail("nsk_grad_extra: the buffer must be 16-byte aligned");
    c->xextra = n_floats ? d_buf : nullptr; c->xextra_n = n_floats;
    return 0;
}
extern "C" int nsk_prepare_rays(nsk_ctx* c, int nframes, const nsk_frame_rays* fr, int per, int H0, int H1, int W0, int W1, int H, int W,
                                float fx, float fy, float cx, float cy, int mode, int32_t* pi, int32_t* pj, float* gd, float* gc,
                                float* ro, float* rd, uint8_t* keep)
{
    if (!c || !fr || !pi || !pj || !gd || !ro || !rd) return fail("nsk_prepare_rays: null argument");
    if (nframes < 1 || nframes > 64 || per < 1) return fail("nsk_prepare_rays: nframes must be 1..64 and rays_per_frame >= 1");
    if (H1 <= H0 || W1 <= W0 || H0 < 0 || W0 < 0 || H1 > H || W1 > W) return fail("nsk_prepare_rays: window [%d,%d) x [%d,%d) outside the %d x %d image", H0, H1, W0, W1, H, W);
    HIPCHK(hipSetDevice(c->device));
    intr(mode, fx, fy, cx, cy);
    for (int f0 = 0; f0 < nframes; f0 += 16) {
        PrepArgs A;
        memset(&A, 0, sizeof(A));
        A.nframes = std::min(16, nframes - f0);
        for (int i = 0; i < A.nframes; ++i) {
            const nsk_frame_rays& s = fr[f0 + i];
            if (!s.d_depth || !s.d_pose) return fail("nsk_prepare_rays: frame %d has no depth image or pose", f0 + i);
            A.f[i].depth = s.d_depth; A.f[i].color = s.d_color; A.f[i].pose = s.d_pose; A.f[i].cam7 = s.pose_is_cam7; A.f[i].seed = s.seed;
        }
        const size_t o = (size_t)f0 * per;
        A.per = per; A.H0 = H0; A.W0 = W0; A.Ww = W1 - W0; A.W = W; A.mode = mode;
        A.total = (unsigned long long)(H1 - H0) * (unsigned long long)(W1 - W0);
        A.fx = fx; A.fy = fy; A.cx = cx; A.cy = cy;
        A.pi = pi + o; A.pj = pj + o; A.gd = gd + o; A.gc = gc ? gc + 3 * o : nullptr; A.ro = ro + 3 * o; A.rd = rd + 3 * o; A.keep = keep ? keep + o : nullptr;
        A.R = c->R;
        { ProfScope ps(c, "prepare_rays"); k_prepare_rays<<<dim3((per + 255) / 256, A.nframes), 256, 0, c->stream>>>(A); }
        HIPCHK(hipGetLastError());
    }
    return 0;
}

extern "C" int nsk_inside_filter(nsk_ctx* c, int N, const float* ro, const float* rd, const float* gt, uint8_t* keep)
{
    if (!c || !ro || !rd || !gt || !keep || N < 1) return fail("nsk_inside_filter: bad argument");
    k_inside_filter<<<(N + 255) / 256, 256, 0, c->stream>>>(c->R, N, ro, rd, gt, keep);
    HIPCHK(hipGetLastError());
    return 0;
}

static void adam_consts(float lr, float b1, float b2, int step, float& step_size, float& bc2s)
{
    float bc1 = 1.f - (float)pow((double)b1, step);
    float bc2 = 1.f - (float)pow((double)b2, step);
    step_size = lr / bc1; bc2s = sqrtf(bc2);
}

extern "C" int nsk_adam_vector(nsk_ctx* c, int n, float* p, const float* g, float* m, float* v, float lr, float b1, float b2, float eps, int step)
{
    if (!c || !p || !g || !m || !v || n < 1 || step < 1) return fail("nsk_adam_vector: bad argument");
    float ss, bc2s; adam_consts(lr, b1, b2, step, ss, bc2s);
    if (c->capturing) c->cap_vecs.push_back(nsk_ctx::CapVec{n, p, g, m, v, lr, b1, b2, eps, step, nullptr, ss, bc2s});
    k_adam_scalar<<<(n + 255) / 256, 256, 0, c->stream>>>(n, p, g, m, v, ss, bc2s, b1, b2, eps);
    HIPCHK(hipGetLastError());
    return 0;
}

// ---- optimiser ------------------------------------------------------------------------------------------------
extern "C" int nsk_adam_step(nsk_ctx* c, const float lr[NSK_NUM_GROUPS], float b1, float b2, float eps)
{
    if (!c || !lr) return fail("nsk_adam_step: null argument");
    HIPCHK(hipSetDevice(c->device));
    AdamArgs AA; memset(&AA, 0, sizeof(AA));
    PackArgs PA; memset(&PA, 0, sizeof(PA));
    AA.b1 = b1; AA.b2 = b2; AA.eps = eps;
    int blocks = 0, pblocks = 0;
    int seg_group[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int lv = 0; lv < 4; ++lv) {
        int grp = NSK_GROUP_COARSE + lv;
        if (!c->touched[grp] || !c->grid[lv].n) continue;
        int step = ++c->adam_step[grp];
        if (c->capturing) ++c->cap_rollback[grp];
        GridState& G = c->grid[lv];
        seg_group[AA.n] = grp;
        AdamSeg& S = AA.s[AA.n++];
        adam_consts(lr[grp], b1, b2, step, S.step_size, S.bc2s);
        CHK(ensure_midx(c, lv));
        S.p = G.v; S.g = c->slab + G.g_off; S.m = G.m; S.v = G.s; S.n = (int)G.n;
        S.idx = G.mask ? G.midx : nullptr; S.nidx = G.mask ? G.nmask : 0;
        blocks += G.mask ? std::max(1, (G.nmask * 8 + 255) / 256) : ((int)G.n / 4 + 255) / 256; S.blk_end = blocks;
        c->touched[grp] = false;
    }
    if (c->touched[NSK_GROUP_DECODERS]) {
        int step = ++c->adam_step[NSK_GROUP_DECODERS];
        if (c->capturing) ++c->cap_rollback[NSK_GROUP_DECODERS];
        for (int w = 0; w < 4; ++w) {
            DecState& D = c->dec[w];
            if (!D.trainable || !D.loaded) continue;
            int n4 = (D.n + 3) & ~3;
            seg_group[AA.n] = NSK_GROUP_DECODERS;
            AdamSeg& S = AA.s[AA.n++];
            adam_consts(lr[NSK_GROUP_DECODERS], b1, b2, step, S.step_size, S.bc2s);
            S.p = D.p; S.g = c->slab + D.g_off; S.m = D.m; S.v = D.s; S.idx = nullptr; S.nidx = 0; S.n = n4;
            S.inv_f = D.finv; S.inv_b = D.binv; S.fimg = D.fimg; S.bimg = D.bimg;
            if (D.bimg16) { S.invh = D.invh; S.imgh = reinterpret_cast<unsigned short*>(D.bimg16); S.imgh_tail = D.bimg16 + MlpBwdImgH::P_WO; S.btail_off = MlpBwdImg::P_WO; }
            else D.bimg16_dirty = true;
            if (c->pend_w == w) {
                S.slabs = c->ws.dec_slabs; S.nslabs = c->pend_nb; S.slab_stride = n4; c->pend_w = -1;
                blocks += (n4 / 4 + 7) / 8 - (n4 / 4 + 255) / 256;         // 8 float4 per block (see k_adam_multi)
            }
            if (D.fimg16) { S.inv16 = D.inv16; S.img16 = reinterpret_cast<unsigned short*>(D.fimg16); S.img16_tail = D.fimg16 + D.tail16_off; S.tail_off = D.tail_off; S.np16 = D.np16; }
            blocks += (n4 / 4 + 255) / 256; S.blk_end = blocks;
        }
        c->touched[NSK_GROUP_DECODERS] = false;
    }
    if (AA.n && c->capturing) {
        nsk_ctx::CapAdam ca; ca.args = AA;
        for (int i = 0; i < AA.n; ++i) { ca.group[i] = seg_group[i]; ca.lr[i] = lr[seg_group[i]]; }
        c->cap_adams.push_back(ca);
    }
    int place_wgs = 0;
    {   // the prepared batch's cell-sort placement rides behind the segments (nsk_map_prepare)
        nsk_ctx::Prep& P = c->prep;
        if (AA.n && P.valid && P.sorted && (P.done & 7) == 3 && !c->capturing && !c->tune_no_piggyback) {
            AA.place = place_args(P.N * P.S, c->ws.skey_alt, c->ws.srank_alt, c->ws.offs_alt, c->ws.perm_alt);
            AA.adam_blocks = blocks; place_wgs = AA.place.nblocks;
            P.done |= 4;
        }
    }
    if (AA.n) { ProfScope ps(c, "adam_multi"); k_adam_multi<<<blocks + place_wgs, 256, 0, c->stream>>>(AA); }
    if (PA.n) { ProfScope ps(c, "pack_images"); k_pack_multi<<<pblocks, 256, 0, c->stream>>>(PA); }
    HIPCHK(hipGetLastError());
    return 0;
}

// ---- hipGraph capture (see nsk_ctx::GraphRec) ---------------------------------------------------------------------
extern "C" int nsk_graph_begin(nsk_ctx* c)
{
    if (!c) return fail("null ctx");
    if (c->capturing) return fail("nsk_graph_begin: a capture is already open");
    HIPCHK(hipSetDevice(c->device));
    CHK(flush_pending(c));
    // A batch registered with nsk_map_prepare is settled BEFORE the capture opens: whatever of its sampling / cell sort has not run yet runs now,
    // eagerly (inside the capture forward_core's prep_finish would have recorded those launches into the graph instead of running them, and the
    // batch's own step would then have swapped in buffers nobody filled); the batch stays prepared for its own eager step.
    CHK(prep_finish(c));
    HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    c->capturing = true;
    c->cap_adams.clear(); c->cap_vecs.clear();
    for (int g = 0; g < NSK_NUM_GROUPS; ++g) c->cap_rollback[g] = 0;
    return 0;
}

extern "C" int nsk_graph_end(nsk_ctx* c, int* graph_id)
{
    if (!c || !graph_id) return fail("nsk_graph_end: null argument");
    if (!c->capturing) return fail("nsk_graph_end: no capture is open");
    c->capturing = false;
    nsk_ctx::GraphRec R;
    hipError_t e = hipStreamEndCapture(c->stream, &R.graph);
    for (int g = 0; g < NSK_NUM_GROUPS; ++g) c->adam_step[g] -= c->cap_rollback[g];      // nothing ran during the capture
    c->pend_w = -1;
    if (e != hipSuccess || !R.graph) return fail("nsk_graph_end: hipStreamEndCapture: %s", hipGetErrorString(e));
    if (c->cap_adams.size() > 1) { hipGraphDestroy(R.graph); return fail("nsk_graph_end: at most one nsk_adam_step per captured graph"); }
    R.vecs = c->cap_vecs;
    if (!c->cap_adams.empty()) { R.has_adam = true; R.adam = c->cap_adams[0]; }
    if (R.has_adam || !R.vecs.empty()) {
        size_t nn = 0;
        HIPCHK(hipGraphGetNodes(R.graph, nullptr, &nn));
        std::vector<hipGraphNode_t> nodes(nn);
        HIPCHK(hipGraphGetNodes(R.graph, nodes.data(), &nn));
        for (hipGraphNode_t nd : nodes) {
            hipGraphNodeType t;
            HIPCHK(hipGraphNodeGetType(nd, &t));
            if (t != hipGraphNodeTypeKernel) continue;
            hipKernelNodeParams kp;
            HIPCHK(hipGraphKernelNodeGetParams(nd, &kp));
            if (kp.func == reinterpret_cast<void*>(k_adam_multi)) R.adam_node = nd;
            if (kp.func == reinterpret_cast<void*>(k_adam_scalar) && kp.kernelParams)
                for (auto& V : R.vecs) if (!V.node && *reinterpret_cast<float**>(kp.kernelParams[1]) == V.p) { V.node = nd; break; }
        }
        if (R.has_adam && !R.adam_node) { hipGraphDestroy(R.graph); return fail("nsk_graph_end: the Adam kernel node was not found in the captured graph"); }
        for (auto& V : R.vecs) if (!V.node) { hipGraphDestroy(R.graph); return fail("nsk_graph_end: an nsk_adam_vector node was not found in the captured graph"); }
    }
    HIPCHK(hipGraphInstantiate(&R.exec, R.graph, nullptr, nullptr, 0));
    R.flip = c->ws_flip;
    c->graphs.push_back(R);
    *graph_id = (int)c->graphs.size() - 1;
    return 0;
}

extern "C" int nsk_graph_launch(nsk_ctx* c, int id)
{
    if (!c || id < 0 || id >= (int)c->graphs.size()) return fail("nsk_graph_launch: bad graph id %d", id);
    if (c->graphs[id].stale) return fail("nsk_graph_launch: graph %d is stale -- a workspace, grid or gradient buffer it recorded was reallocated after the capture (larger batch, new grid shape or mask); capture it again", id);
    if (!c->graphs[id].exec) return fail("nsk_graph_launch: graph %d was destroyed", id);
    HIPCHK(hipSetDevice(c->device));
    nsk_ctx::GraphRec& R = c->graphs[id];
    // A prepared batch (nsk_map_prepare) and a replay share the cell histogram and, after an odd number of set swaps since the capture, the very
    // buffers the graph writes: finish the batch's pending stages first (a histogram still holding its counts would be added onto by the
    // replayed sampling), and where the graph's recorded set is the one the batch lives in, drop the batch -- its own step samples it again.
    CHK(prep_finish(c));
    if (c->prep.valid && R.flip != c->ws_flip) CHK(prep_drop(c));
    if (R.has_adam) {                            // this replay is one more Adam step for the groups the graph updates
        AdamArgs& A = R.adam.args;
        int stepped[NSK_NUM_GROUPS] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < A.n; ++i) {
            const int g = R.adam.group[i];
            if (!stepped[g]) { ++c->adam_step[g]; stepped[g] = 1; }
            adam_consts(R.adam.lr[i], A.b1, A.b2, c->adam_step[g], A.s[i].step_size, A.s[i].bc2s);
        }
        hipKernelNodeParams kp;
        HIPCHK(hipGraphKernelNodeGetParams(R.adam_node, &kp));
        void* args[1] = {&A};
        kp.kernelParams = args; kp.extra = nullptr;
        HIPCHK(hipGraphExecKernelNodeSetParams(R.exec, R.adam_node, &kp));
    }
    for (auto& V : R.vecs) {                      // the pose (or any plain vector) Adam: one more step per replay
        adam_consts(V.lr, V.b1, V.b2, V.step, V.ss, V.bc2s);
        ++V.step;
        hipKernelNodeParams kp;
        HIPCHK(hipGraphKernelNodeGetParams(V.node, &kp));
        void* args[10] = {&V.n, &V.p, &V.g, &V.m, &V.v, &V.ss, &V.bc2s, &V.b1, &V.b2, &V.eps};
        kp.kernelParams = args; kp.extra = nullptr;
        HIPCHK(hipGraphExecKernelNodeSetParams(R.exec, V.node, &kp));
    }
    HIPCHK(hipGraphLaunch(R.exec, c->stream));
    return 0;
}

extern "C" int nsk_graph_destroy(nsk_ctx* c, int id)
{
    if (!c || id < 0 || id >= (int)c->graphs.size()) return fail("nsk_graph_destroy: bad graph id %d", id);
    nsk_ctx::GraphRec& R = c->graphs[id];
    if (R.exec) { HIPCHK(hipStreamSynchronize(c->stream)); hipGraphExecDestroy(R.exec); hipGraphDestroy(R.graph); R.exec = nullptr; R.graph = nullptr; }
    return 0;
}

extern "C" int nsk_adam_reset(nsk_ctx* c)
{
    if (!c) return fail("null ctx");
    for (int i = 0; i < 4; ++i) {
        if (c->grid[i].n) { HIPCHK(hipMemsetAsync(c->grid[i].m, 0, c->grid[i].n * 4, c->stream)); HIPCHK(hipMemsetAsync(c->grid[i].s, 0, c->grid[i].n * 4, c->stream)); }
        if (c->dec[i].p) { size_t n4 = (c->dec[i].n + 3) & ~3; HIPCHK(hipMemsetAsync(c->dec[i].m, 0, n4 * 4, c->stream)); HIPCHK(hipMemsetAsync(c->dec[i].s, 0, n4 * 4, c->stream)); }
    }
    for (int g = 0; g < NSK_NUM_GROUPS; ++g) { c->adam_step[g] = 0; c->touched[g] = false; }
    return 0;
}

extern "C" int nsk_zero_grads(nsk_ctx* c)
{
    if (!c) return fail("null ctx");
    c->pend_w = -1;
    if (c->slab) HIPCHK(hipMemsetAsync(c->slab, 0, c->slab_n * 4, c->stream));
    for (int g = 0; g < NSK_NUM_GROUPS; ++g) c->touched[g] = false;
    return 0;
}

// ---- multi-GPU ------------------------------------------------------------------------------------------------
extern "C" int nsk_grad_slab(nsk_ctx* c, float** p, size_t* n)
{
    if (!c || !p || !n) return fail("nsk_grad_slab: null argument");
    CHK(flush_pending(c));
    *p = c->slab; *n = c->slab_n;
    return 0;
}

// ascending list of a level's marked voxels (used by the optimiser launch and by the packed exchange); rebuilt when the mask changed
static int ensure_midx(nsk_ctx* c, int l)
{
    GridState& G = c->grid[l];
    if (!G.mask || !G.midx_dirty) return 0;
    if (c->capturing) return fail("graph capture: a mask changed since the last eager step (run the step once after installing masks, then capture)");
    const int nvox = (int)(G.n / 32);
    if (!G.midx) HIPCHK(hipMalloc(&G.midx, (size_t)nvox * 4));
    int* d_count = reinterpret_cast<int*>(c->scal + 8);
    k_mask_index<<<1, 1024, 0, c->stream>>>(nvox, G.mask, G.midx, d_count);
    HIPCHK(hipMemcpyAsync(&G.nmask, d_count, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));                 // once per mask, not per step
    G.midx_dirty = false;
    return 0;
}

// Which parts of the slab a step's exchange must carry: the grid levels and trainable decoders that received gradients since the
// last optimiser step (the same on every rank: it follows from the stage and the flags) and the loss scalars.
static int pack_layout(nsk_ctx* c, size_t* total)
{
    size_t n = 0;
    for (int l = 0; l < 4; ++l) {
        GridState& G = c->grid[l];
        c->xlevels[l] = G.n && c->touched[NSK_GROUP_COARSE + l];
        if (!c->xlevels[l]) continue;
        CHK(ensure_midx(c, l));
        n += (size_t)(G.mask ? G.nmask : (int)(G.n / 32)) * 32;
    }
    for (int w = 0; w < 4; ++w) {
        c->xdecs[w] = c->dec[w].loaded && c->dec[w].trainable && c->touched[NSK_GROUP_DECODERS];
        if (c->xdecs[w]) n += (size_t)((c->dec[w].n + 3) & ~3);
    }
    n += 4;
    n += c->xextra_n;
    *total = n;
    return 0;
}

static int pack_move(nsk_ctx* c, bool gather)
{
    XArgs A;
    memset(&A, 0, sizeof(A));
    A.gather = gather ? 1 : 0;
    size_t o = 0;
    int blocks = 0;
    auto add = [&](const int* idx, size_t n4, float* slab, int pend_w = -1) {
        if (n4 == 0) return;
        XSeg& S = A.s[A.n++];
        S.idx = idx; S.n4 = (int)n4; S.slab = slab; S.buf = c->xbuf + o;
        if (gather && pend_w >= 0 && c->pend_w == pend_w) {      // its gradient is still in the backward's per-workgroup slabs: summed by this launch
            S.slabs = c->ws.dec_slabs; S.nslabs = c->pend_nb; S.slab_stride = (int)(n4 * 4);
            blocks += (int)((n4 + 7) / 8);
            c->pend_w = -1;
        } else blocks += (int)((n4 + 255) / 256);
        S.blk_end = blocks;
        o += n4 * 4;
    };
    for (int l = 0; l < 4; ++l) {
        if (!c->xlevels[l]) continue;
        GridState& G = c->grid[l];
        const int nv = G.mask ? G.nmask : (int)(G.n / 32);
        add(G.mask ? G.midx : nullptr, (size_t)nv * 8, c->slab + G.g_off);
    }
    for (int w = 0; w < 4; ++w) {
        if (!c->xdecs[w]) continue;
        add(nullptr, (size_t)((c->dec[w].n + 3) & ~3) / 4, c->slab + c->dec[w].g_off, w);
    }
    add(nullptr, 1, c->slab + c->slab_n - 4);                 // loss scalars
    if (c->xextra_n) add(nullptr, c->xextra_n / 4, c->xextra); // the caller's vector (nsk_grad_extra: pose gradients of a bundle-adjustment step)
    if (blocks > 0) k_xchg_multi<<<blocks, 256, 0, c->stream>>>(A);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int nsk_grad_pack(nsk_ctx* c, float** p, size_t* n)
{
    if (!c || !p || !n) return fail("nsk_grad_pack: null argument");
    HIPCHK(hipSetDevice(c->device));
    size_t total = 0;
    CHK(pack_layout(c, &total));
    bool any_mask = false;
    for (int l = 0; l < 4; ++l) any_mask = any_mask || (c->xlevels[l] && c->grid[l].mask);
    if (!any_mask && !c->xextra_n) {                   // nothing to compact: exchange the slab in place (no copies); unpack is then a no-op
        CHK(flush_pending(c));
        c->xbuf_n = 0; c->x_identity = true;
        *p = c->slab; *n = c->slab_n;
        return 0;
    }
    c->x_identity = false;
    if (total > c->xbuf_cap) {
        HIPCHK(hipStreamSynchronize(c->stream));
        invalidate_graphs(c);
        hipFree(c->xbuf);
        HIPCHK(hipMalloc(&c->xbuf, total * 4));
        c->xbuf_cap = total;
    }
    c->xbuf_n = total;
    { ProfScope ps(c, "grad_pack"); CHK(pack_move(c, true)); }
    *p = c->xbuf; *n = total;
    return 0;
}

extern "C" int nsk_grad_unpack(nsk_ctx* c)
{
    if (!c) return fail("null ctx");
    if (c->x_identity) return 0;
    if (!c->xbuf || !c->xbuf_n) return fail("nsk_grad_unpack: nothing was packed (call nsk_grad_pack after nsk_map_step)");
    HIPCHK(hipSetDevice(c->device));
    ProfScope ps(c, "grad_unpack");
    return pack_move(c, false);
}

extern "C" int nsk_allreduce_grads(nsk_ctx* c, void* comm)
{
    if (!c || !comm) return fail("nsk_allreduce_grads: null argument");
    typedef int (*allreduce_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
    static allreduce_t fn = nullptr;
    if (!fn) {
        void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return fail("nsk_allreduce_grads: cannot load librccl.so: %s", dlerror());
        fn = (allreduce_t)dlsym(h, "ncclAllReduce");
        if (!fn) return fail("nsk_allreduce_grads: ncclAllReduce not found");
    }
    float* buf = nullptr; size_t n = 0;
    CHK(nsk_grad_pack(c, &buf, &n));
    const int ncclFloat32 = 7, ncclSum = 0;
    int r;
    { ProfScope ps(c, "allreduce"); r = fn(buf, buf, n, ncclFloat32, ncclSum, comm, c->stream); }      // HIP events on the context's stream: the collective alone
    if (r != 0) return fail("ncclAllReduce failed with %d", r);
    return nsk_grad_unpack(c);
}

extern "C" int nsk_profile_begin(nsk_ctx* c)
{
    if (!c) return fail("null ctx");
    for (auto& r : c->prof_recs) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
    c->prof_recs.clear();
    c->prof = true;
    return 0;
}

extern "C" int nsk_profile_end(nsk_ctx* c, char* buf, size_t n)
{
    if (!c || !buf || n < 2) return fail("nsk_profile_end: bad argument");
    c->prof = false;
    HIPCHK(hipStreamSynchronize(c->stream));
    std::vector<std::string> names; std::vector<double> tot; std::vector<int> cnt;
    for (auto& r : c->prof_recs) {
        float ms = 0.f;
        hipEventElapsedTime(&ms, r.a, r.b);
        size_t k = 0;
        for (; k < names.size(); ++k) if (names[k] == r.name) break;
        if (k == names.size()) { names.push_back(r.name); tot.push_back(0); cnt.push_back(0); }
        tot[k] += ms; cnt[k] += 1;
        hipEventDestroy(r.a); hipEventDestroy(r.b);
    }
    c->prof_recs.clear();
    std::string out;
    for (size_t k = 0; k < names.size(); ++k) { char line[160]; snprintf(line, sizeof(line), "%s %d %.6f\n", names[k].c_str(), cnt[k], tot[k]); out += line; }
    if (out.size() + 1 > n) return fail("nsk_profile_end: buffer too small (%zu needed)", out.size() + 1);
    memcpy(buf, out.c_str(), out.size() + 1);
    return 0;
}

extern "C" int nsk_last_call_stats(nsk_ctx* c, double* b, double* f, int* s)
{
    if (!c) return fail("null ctx");
    if (b) *b = c->last_bytes; if (f) *f = c->last_flops; if (s) *s = c->last_samples;
    return 0;
}
