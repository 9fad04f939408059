// nsk_device.h -- gfx950 device code of the render / map / track kernel family.
// Wave = 64 lanes; MFMA = v_mfma_f32_16x16x4_f32 (exact fp32, SURVEY.md 7.2: the 1e-4 contract needs fp32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "nsk_layout.h"

typedef float f4 __attribute__((ext_vector_type(4)));

struct RParams {            // Renderer constants, reference src/Renderer.cpp:5-15
    float bound[6];
    int n_samples, n_surface, lindisp, occupancy;
    float perturb;
    unsigned long long seed;
};

struct GridD {              // one feature grid level, voxel-major [Z][Y][X][32]
    const float* v;
    float* g;               // gradient (same layout) or nullptr
    const uint8_t* mask;    // per-voxel optimiser mask or nullptr
    int Z, Y, X;
};

#define NSK_INF __builtin_huge_valf()
#ifdef NSK_EXPERIMENT
__device__ int nsk_dbg_flags;
// range audit of every index that addresses a per-sample array in the decoder bodies (experiment builds; tools/exp_idx.py): counts of
// indices outside [0, n) by site -- 0 perm entry, 1 sample index, 2 slot (ReLU bits), 3 tile (saved block outputs), 4 ray
__device__ unsigned nsk_dbg_oob[8];
#define NSK_IDX(site, i, n) do { if ((long long)(i) < 0 || (long long)(i) >= (long long)(n)) atomicAdd(&nsk_dbg_oob[site], 1u); } while (0)
#else
#define NSK_IDX(site, i, n)
#endif

__device__ __forceinline__ f4 mfma4(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// A workgroup's copy of a fragment image into LDS.  Written as the obvious loop the compiler emits load, s_waitcnt vmcnt(0), ds_write per
// iteration: eleven L2 round trips one after the other (~4 us) at the start of every decoder launch.  Eight loads in flight per thread.
template <int NT>
__device__ __forceinline__ void copy_image_to_lds(f4* __restrict__ dst, const f4* __restrict__ src, int n4)
{
    for (int i0 = 0; i0 < n4; i0 += 8 * NT) {
        f4 r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + u * NT + (int)threadIdx.x; r[u] = i < n4 ? src[i] : (f4)(0.f); }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + u * NT + (int)threadIdx.x; if (i < n4) dst[i] = r[u]; }
    }
}

// ------------------------------------------------------------------------------------------------------
// scalar helpers that spell out the operation sequence of the reference's libtorch ops for the sampling geometry (it feeds sin(25*x) and
// ReLU kinks): every libtorch tensor op rounds on its own, so `rays_o + rays_d * z` (src/Renderer.cpp:121) is a rounded product and a
// rounded sum.  hipcc contracts a * b + c into an FMA by default, across statements and inlined calls, and decides so per instantiation:
// until round 4 the SAME source gave different p = o + d z in different kernels (two instantiations of one forward body disagreed on
// ~130 of 38 M ReLU branches at K3, as many as either disagreed with the CPU oracle: tools/relu_flips.py).  mul_rn
// therefore hides its product from the optimiser behind an empty asm (no instruction, the value just becomes opaque): a product formed by
// mul_rn is never fused into a following add.  Where the reference itself fuses -- torch::matmul's K = 3 dot product in GaussianFFT is an
// FMA chain on the CPU (tests/test_oracle.py::test_aten_matmul_k3_is_an_fma_chain) -- the code says so with an explicit fmaf (embed()).
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float mul_rn(float a, float b) { float r = a * b; asm("" : "+v"(r)); return r; }
__device__ __forceinline__ float add_rn(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float sub_rn(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ float div_rn(float a, float b) { return __fdiv_rn(a, b); }
// at::linspace CPU kernel (reference src/Renderer.cpp:86,101)
__device__ __forceinline__ float linspace01(int i, int steps)
{
    if (steps == 1) return 0.f;
    float step = div_rn(1.f, (float)(steps - 1));
    return i < steps / 2 ? mul_rn(step, (float)i) : sub_rn(1.f, mul_rn(step, (float)(steps - 1 - i)));
}

// reference src/Renderer.cpp:66-73, src/Mapper.cpp:417-421: min_axis max_side (bound - o)/d
__device__ __forceinline__ float ray_box_far(const float* bound, float ox, float oy, float oz, float dx, float dy, float dz)
{
    float o[3] = {ox, oy, oz}, d[3] = {dx, dy, dz};
    float far = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float t0 = div_rn(sub_rn(bound[2 * k], o[k]), d[k]);
        float t1 = div_rn(sub_rn(bound[2 * k + 1], o[k]), d[k]);
        float m = t0 > t1 ? t0 : t1;
        if (k == 0 || m < far) far = m;
    }
    return far;
}

__device__ __forceinline__ uint32_t hash_u32(unsigned long long seed, uint32_t a, uint32_t b)
{
    unsigned long long x = seed ^ (0x9E3779B97F4A7C15ull * ((unsigned long long)a + 1)) ^
                           (0xC2B2AE3D27D4EB4Full * ((unsigned long long)b + 1));
    x ^= x >> 33; x *= 0xFF51AFD7ED558CCDull; x ^= x >> 33; x *= 0xC4CEB9FE1A85EC53ull; x ^= x >> 33;
    return (uint32_t)(x >> 32);
}

// wave-wide helpers (64 lanes)
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ bool wave_all(bool p) { return __builtin_amdgcn_ballot_w64(p) == ~0ull; }   // every lane active
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// ------------------------------------------------------------------------------------------------------
// K0: batch maximum of gt_depth (reference src/Renderer.cpp:76,93 torch::max(gt_depth)) -> *out
// ------------------------------------------------------------------------------------------------------
__global__ void k_depth_max(int N, const float* __restrict__ gt, const uint8_t* __restrict__ keep, float* __restrict__ out)
{
    __shared__ float sh[16];
    float m = keep ? 0.f : -NSK_INF;          // with a ray mask an empty selection is possible: its maximum is 0 (depths are >= 0), not -inf
    for (int i = threadIdx.x; i < N; i += blockDim.x) if (!keep || keep[i]) m = fmaxf(m, gt[i]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = fmaxf(m, sh[w]);
        *out = m;
    }
}

// histogram slot of a sort key (= cell * 8 + sub-bin): the 8 sub-bins of a cell stay together (32 bytes), but consecutive cells go
// to different 64-byte lines (a line holds cells c and c + ncell2): same-line atomics serialise at the memory side and
// neighbouring cells are hot together
__device__ __forceinline__ int hist_slot(int key, int ncell2) { const int cell = key >> 3; return (((cell % ncell2) * 2 + cell / ncell2) << 3) + (key & 7); }

// index of the grid cell (lower corner voxel, clamped into the grid) that holds world point p: the i0 of tri_setup below
__device__ __forceinline__ int cell_index(int GX, int GY, int GZ, const float* bound, float px, float py, float pz)
{
    const int dims[3] = {GX, GY, GZ};
    const float p[3] = {px, py, pz};
    int i0[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float lo = bound[2 * k], hi = bound[2 * k + 1];
        float u = sub_rn(mul_rn(div_rn(sub_rn(p[k], lo), sub_rn(hi, lo)), 2.f), 1.f);
        float x = mul_rn(div_rn(add_rn(u, 1.f), 2.f), (float)(dims[k] - 1));
        float mx = (float)(dims[k] - 1);
        if (x <= 0.f) x = 0.f;
        else if (x >= mx) x = mx;
        i0[k] = max(0, min((int)floorf(x), dims[k] - 1));
    }
    return (i0[2] * GY + i0[1]) * GX + i0[0];
}

// ------------------------------------------------------------------------------------------------------
// K1: per-ray z sampling (reference src/Renderer.cpp:44-119).  One wave per ray, lane = sample slot.
// z_out [N][S] sorted ascending.
// Cell sort (optional, skey != nullptr): every sample also gets the index of the grid cell it falls in (grid kX x kY x kZ =
// the finest level the stage reads) and its arrival rank inside that cell's histogram bin; k_sort_scan / k_sort_place turn
// these into the permutation `perm` that lists the samples cell by cell.  The decoders then walk their 16-sample tiles
// in that order, so that the samples of a tile share cells: the backward's scatter into the grid gradient sums them in
// registers and issues one atomic flush per cell run instead of one per sample and cell (global float atomics run at
// ~1.3 TB/s chip-wide and bounded the backward: DESIGN.md section 4).
// ------------------------------------------------------------------------------------------------------
#ifndef NSK_SAMPLE_RAYS
#define NSK_SAMPLE_RAYS 8           // rays (waves) per workgroup of k_sample (16: 19.6 us at 5000 rays, 8: 18.1, 4: 20.9; 1000 rays: 11.5 / 10.6 / 12.3)
#endif
#define NSK_SAMPLE_TABLE 2048       // slots of its cell table (>= 2 x NSK_SAMPLE_RAYS x 64 keeps probing short)
#ifdef NSK_EXPERIMENT
extern __device__ unsigned long long nsk_dbg_ts[2][1024][4];
// stamps of one workgroup of the sampling (tools/exp_sample.py): slot k of the unused tail of the forward's stamp array
#define NSK_SS(k) do { if (bid == 300 && threadIdx.x == 0) nsk_dbg_ts[0][1000 + (k)][0] = __builtin_readcyclecounter(); } while (0)
#else
#define NSK_SS(k)
#endif
struct SampArgs {
    RParams R; int N, S;
    const float* rays_o; const float* rays_d; const float* gt_depth; float gtmax_host; const float* gtmax_dev; const uint8_t* keep;
    float* z_out; int kX, kY, kZ, pX, pY, pZ, ncell2; int* skey; int* srank; int* hist;
    const float* mx_gt; const uint8_t* mx_keep; int mx_n;      // what the batch maximum of gt_depth runs over: the call's own rays, or the whole batch a shard belongs to (nsk_set_depth_max_batch)
};
// (a body, so that the sampling of the NEXT batch can ride in the composite launch of the current step: k_composite_sample, nsk_map_prepare)
__device__ __forceinline__ void sample_body(const SampArgs& P, int bid)
{
    const RParams& R = P.R;
    const int N = P.N, S = P.S, kX = P.kX, kY = P.kY, kZ = P.kZ, pX = P.pX, pY = P.pY, pZ = P.pZ, ncell2 = P.ncell2;
    const float* __restrict__ rays_o = P.rays_o; const float* __restrict__ rays_d = P.rays_d; const float* __restrict__ gt_depth = P.gt_depth;
    const float gtmax_host = P.gtmax_host; const float* __restrict__ gtmax_dev = P.gtmax_dev; const uint8_t* __restrict__ keep = P.keep;
    float* __restrict__ z_out = P.z_out; int* __restrict__ skey = P.skey; int* __restrict__ srank = P.srank; int* __restrict__ hist = P.hist;
    __shared__ float sh[NSK_SAMPLE_RAYS][64];
    __shared__ float sh2[NSK_SAMPLE_RAYS][64];
    __shared__ int tkey[NSK_SAMPLE_TABLE], tcnt[NSK_SAMPLE_TABLE], tbase[NSK_SAMPLE_TABLE];
    __shared__ int tlist[64 * NSK_SAMPLE_RAYS], nlist;      // the occupied slots of the table, in order of arrival
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = bid * NSK_SAMPLE_RAYS + wave;
    const bool active = n < N;                          // whole waves; inactive ones only take part in the barriers below
    const int nc = active ? n : N - 1;
    NSK_SS(0);
    const bool has_gt = gt_depth != nullptr;
    const int ns = R.n_samples;
    const int nsurf = S - ns;
    const float gt = has_gt ? gt_depth[nc] : 0.f;
    float gmax = gtmax_dev ? *gtmax_dev : gtmax_host;
    if (has_gt && !gtmax_dev && gtmax_host < 0.f) {         // batch maximum computed by every wave itself (small batches: one launch fewer)
        // (four independent (depth, keep) load pairs per round: written as `if (keep[i]) mx = max(mx, gt[i])` the loop was two dependent
        // L2 round trips per 64 rays -- half of this kernel's 19 us at 5000 rays)
        const float* __restrict__ mgt = P.mx_gt; const uint8_t* __restrict__ mkeep = P.mx_keep; const int MN = P.mx_n;
        const float mx0 = mkeep ? 0.f : -NSK_INF;           // (see k_depth_max)
        float mx = mx0;
        for (int i0 = 0; i0 < MN; i0 += 256) {
            float v[4]; uint8_t kp[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = min(i0 + 64 * u + lane, MN - 1);         // (clamped: a repeated element does not change a maximum)
                v[u] = mgt[i];
                kp[u] = mkeep ? mkeep[i] : (uint8_t)1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) mx = fmaxf(mx, kp[u] ? v[u] : mx0);
        }
        gmax = wave_max(mx);
    }
    const float ox = rays_o[3 * nc], oy = rays_o[3 * nc + 1], oz = rays_o[3 * nc + 2];
    const float dx = rays_d[3 * nc], dy = rays_d[3 * nc + 1], dz = rays_d[3 * nc + 2];
    float near = has_gt ? mul_rn(gt, 0.01f) : 0.01f;                        // :57,:63
    float far = add_rn(ray_box_far(R.bound, ox, oy, oz, dx, dy, dz), 0.01f);   // :69-73
    if (has_gt) {                                                           // :76
        float hi = mul_rn(gmax, 1.2f);
        if (far < 0.f) far = 0.f;
        if (far > hi) far = hi;
    }
    float z = NSK_INF;
    if (lane < ns) {                                                        // :101-108
        float t = linspace01(lane, ns);
        if (!R.lindisp) z = add_rn(mul_rn(near, sub_rn(1.f, t)), mul_rn(far, t));
        else z = div_rn(1.f, add_rn(mul_rn(div_rn(1.f, near), sub_rn(1.f, t)), mul_rn(div_rn(1.f, far), t)));
    }
    if (R.perturb > 0.f) {                                                  // :110-117
        sh[wave][lane] = z;
        lds_fence();
        if (lane < ns) {
            float zl = lane > 0 ? sh[wave][lane - 1] : z;
            float zu = lane < ns - 1 ? sh[wave][lane + 1] : z;
            float lo = lane == 0 ? z : mul_rn(0.5f, add_rn(z, zl));
            float up = lane == ns - 1 ? z : mul_rn(0.5f, add_rn(zu, z));
            float u = (float)(hash_u32(R.seed, (uint32_t)nc, (uint32_t)lane) >> 8) * (1.0f / 16777216.0f);
            z = add_rn(lo, mul_rn(sub_rn(up, lo), u));
        }
        lds_fence();
    }
    if (lane >= ns && lane < S) {                                           // :80-99
        float t = linspace01(lane - ns, nsurf);
        if (gt > 0.f) z = add_rn(mul_rn(mul_rn(0.95f, gt), sub_rn(1.f, t)), mul_rn(mul_rn(1.05f, gt), t));
        else z = add_rn(mul_rn(0.001f, sub_rn(1.f, t)), mul_rn(gmax, t));
    }
    if (nsurf > 0) {                                                        // :119 sort(cat) as a rank sort
        sh[wave][lane] = z;
        lds_fence();
        int rank = 0;
        for (int k = 0; k < S; ++k) {
            float zk = sh[wave][k];
            rank += (zk < z || (zk == z && k < lane)) ? 1 : 0;
        }
        if (lane < S) sh2[wave][rank] = z;
        lds_fence();
        z = sh2[wave][lane];
    }
    if (active && lane < S) z_out[(size_t)n * S + lane] = z;
    NSK_SS(1);
    if (!skey) return;                                                      // uniform over the launch
    // ---- cell keys and ranks for the cell sort -----------------------------------------------------------------------------
    if (bid == 0 && threadIdx.x == 0) hist[-1] = 0;                 // the bump cursor of k_sort_scan (one int in front of the histogram)
    for (int i = threadIdx.x; i < NSK_SAMPLE_TABLE; i += 64 * NSK_SAMPLE_RAYS) { tkey[i] = -1; tcnt[i] = 0; }
    if (threadIdx.x == 0) nlist = 0;
    NSK_SS(2);
    int cell = -1;
    if (active && lane < S) {
        const float px = add_rn(ox, mul_rn(dx, z)), py = add_rn(oy, mul_rn(dy, z)), pz = add_rn(oz, mul_rn(dz, z));   // = sample_finish
        cell = cell_index(kX, kY, kZ, R.bound, px, py, pz) * 8;
        // The coarser level read beside the key level (grid_middle under grid_fine / grid_color) does not nest in it: grid_sample's
        // align_corners scaling puts its cell faces INSIDE key cells, so the samples of one key cell fall into up to 2 x 2 x 2 parent
        // cells.  Three parity bits of the parent cell order them inside the key cell; without them they alternate at random and
        // the parent level's scatter finds a new run at every other sample.
        if (pX > 0) {
            const int pc = cell_index(pX, pY, pZ, R.bound, px, py, pz);
            const int ix = pc % pX, iy = (pc / pX) % pY, iz = pc / (pX * pY);
            cell += (ix & 1) | ((iy & 1) << 1) | ((iz & 1) << 2);
        }
    }
    // A ray crosses a cell once, so equal cells are consecutive lanes: one histogram add per run, made by its first lane.  The
    // runs of the workgroup's rays are first merged in an LDS table (rays of one frame all start in the cells around the camera,
    // and a thousand adds on one 64-byte line take 25 us: same-line atomics serialise at the memory side), then every
    // distinct cell of the workgroup makes ONE returning add on the global histogram.
    NSK_SS(3);
    const int prev = __shfl_up(cell, 1);
    const bool leader = active && lane < S && (lane == 0 || prev != cell);
    const unsigned long long lead = __builtin_amdgcn_ballot_w64(leader);
    const unsigned long long upto = (2ull << lane) - 1ull;                       // bits 0..lane
    const int start = 63 - __builtin_clzll((lead & upto) | 1ull);
    const unsigned long long above = lead & ~upto;
    const int next = above ? __builtin_ctzll(above) : S;
    __syncthreads();
    int slot = 0, off = 0;
    if (leader) {
        unsigned h = ((unsigned)cell * 2654435761u) >> 21;                        // 11 bits
        for (;;) {
            const int old = atomicCAS(&tkey[h], -1, cell);
            if (old == -1) { tlist[atomicAdd(&nlist, 1)] = (int)h; break; }       // first run of this cell in the workgroup
            if (old == cell) break;
            h = (h + 1) & (NSK_SAMPLE_TABLE - 1);
        }
        slot = (int)h;
        off = atomicAdd(&tcnt[h], next - lane);
    }
    NSK_SS(4);
    __syncthreads();
    NSK_SS(5);
    // one returning add per distinct cell, ONE per thread (at most 64 x rays distinct cells): walking the table itself, a thread met up to
    // four occupied slots and made their adds one after the other -- two to three round trips of the 18 us this launch took at 5000 rays
    if ((int)threadIdx.x < nlist) { const int i = tlist[threadIdx.x]; tbase[i] = atomicAdd(hist + hist_slot(tkey[i], ncell2), tcnt[i]); }
    __syncthreads();
    NSK_SS(6);
    int base = leader ? tbase[slot] + off : 0;
    base = __shfl(base, start);
    if (active && lane < S) { skey[(size_t)n * S + lane] = cell; srank[(size_t)n * S + lane] = base + (lane - start); }
    NSK_SS(7);
}
__global__ __launch_bounds__(64 * NSK_SAMPLE_RAYS) void k_sample(SampArgs P) { sample_body(P, blockIdx.x); }

// Offsets of the cell sort.  Each workgroup scans a chunk of 256 consecutive cells (2048 keys; one cell = 8 keys per thread) and
// takes the chunk's place in the output with ONE returning add on a cursor: chunks land in arrival order (keys stay sorted
// inside a chunk, which is all the tiles need), and no workgroup waits for another.  The histogram is cleared for the next step.
// (a body over 256 threads = one chunk; HALVES chunks per workgroup, so that the scan can ride in a 512-thread launch: k_decode_bwd_multi)
struct ScanArgs { int nkeys, ncell2; int* hist; int* offs; int nblocks; };
template <int HALVES>
__device__ __forceinline__ void sort_scan_body(const ScanArgs& P, int bid)
{
    __shared__ int wsum_[HALVES][4];
    __shared__ int sbase_[HALVES];
    const int half = HALVES > 1 ? (int)(threadIdx.x >> 8) : 0, tid = threadIdx.x & 255;
    int* wsum = wsum_[half]; int& sbase = sbase_[half];
    const int nkeys = P.nkeys, ncell2 = P.ncell2; int* __restrict__ hist = P.hist; int* __restrict__ offs = P.offs;
    const int c0 = ((bid * HALVES + half) * 256 + tid) * 8;
    int v[8], s = 0;
    if (c0 < nkeys) {
        const int4* src = reinterpret_cast<const int4*>(hist + hist_slot(c0, ncell2));
        const int4 a = src[0], b = src[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    int incl = s;
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int up = __shfl_up(incl, o); if (lane >= o) incl += up; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    if (tid == 0) sbase = atomicAdd(hist - 1, wsum[0] + wsum[1] + wsum[2] + wsum[3]);
    __syncthreads();
    int run = sbase + incl - s;
    for (int w = 0; w < wave; ++w) run += wsum[w];
    if (c0 < nkeys) {
        int o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { o[i] = run; run += v[i]; }
        int4* dst = reinterpret_cast<int4*>(offs + c0);
        dst[0] = make_int4(o[0], o[1], o[2], o[3]); dst[1] = make_int4(o[4], o[5], o[6], o[7]);
        if (s) { int4* h = reinterpret_cast<int4*>(hist + hist_slot(c0, ncell2)); h[0] = make_int4(0, 0, 0, 0); h[1] = make_int4(0, 0, 0, 0); }
    }
}
__global__ __launch_bounds__(256) void k_sort_scan(ScanArgs P) { sort_scan_body<1>(P, blockIdx.x); }

struct PlaceArgs { int M; const int* skey; const int* srank; const int* offs; int* perm; int nblocks; };
__device__ __forceinline__ void sort_place_body(const PlaceArgs& P, int bid)
{
    const int m = bid * blockDim.x + threadIdx.x;
    if (m < P.M) P.perm[P.offs[P.skey[m]] + P.srank[m]] = m;
}
__global__ void k_sort_place(PlaceArgs P) { sort_place_body(P, blockIdx.x); }

// ------------------------------------------------------------------------------------------------------
// trilinear lookup = F::grid_sample(bilinear, border, align_corners=true) (reference src/models/MLP.cpp:51-63,
// normalize_3d_coordinate include/torchlib/utils.h:132-139; ATen GridSampler.h:27-83)
// ------------------------------------------------------------------------------------------------------
struct Tri {
    int vox[8];      // voxel index of corner c = dz*4 + dy*2 + dx (clamped into the grid)
    float w[8];      // trilinear weight (0 for corners outside the grid)
    float t[3];      // fractional coordinates
    float gmul[3];   // d(index)/d(world), 0 where the coordinate was clipped
};

__device__ __forceinline__ void tri_setup(const GridD& G, const float* bound, float px, float py, float pz, Tri& T)
{
    const int dims[3] = {G.X, G.Y, G.Z};
    const float p[3] = {px, py, pz};
    int i0[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float lo = bound[2 * k], hi = bound[2 * k + 1];
        float u = sub_rn(mul_rn(div_rn(sub_rn(p[k], lo), sub_rn(hi, lo)), 2.f), 1.f);
        float x = mul_rn(div_rn(add_rn(u, 1.f), 2.f), (float)(dims[k] - 1));
        float mul = mul_rn(div_rn((float)(dims[k] - 1), 2.f), div_rn(2.f, sub_rn(hi, lo)));
        float mx = (float)(dims[k] - 1);
        if (x <= 0.f) { x = 0.f; mul = 0.f; }
        else if (x >= mx) { x = mx; mul = 0.f; }
        float f = floorf(x);
        i0[k] = max(0, min((int)f, dims[k] - 1));     // NaN / inf coordinates stay inside the grid
        T.t[k] = sub_rn(x, f); T.gmul[k] = mul;
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        int dx = c & 1, dy = (c >> 1) & 1, dz = c >> 2;
        int ix = i0[0] + dx, iy = i0[1] + dy, iz = i0[2] + dz;
        bool in = ix < G.X && iy < G.Y && iz < G.Z;
        float w = mul_rn(mul_rn(dx ? T.t[0] : sub_rn(1.f, T.t[0]), dy ? T.t[1] : sub_rn(1.f, T.t[1])),
                         dz ? T.t[2] : sub_rn(1.f, T.t[2]));
        ix = min(ix, G.X - 1); iy = min(iy, G.Y - 1); iz = min(iz, G.Z - 1);
        T.vox[c] = (iz * G.Y + iy) * G.X + ix;
        T.w[c] = in ? w : 0.f;
    }
}

// address of channels 4g.. of voxel `vox`: the level's base pointer is wave-uniform (kernel argument) and the byte offset fits 32 bits (the host keeps
// a level below 2^25 voxels: nsk_grid_upload), so the load can take the scalar base + 32-bit vector offset form instead of a 64-bit vector address
// (a shift, a 64-bit add and a carry chain per corner: ~6 % of the forward's vector instructions were 64-bit address arithmetic)
// O32 = false keeps the 64-bit form: in the backward bodies (at the register limit) the 32-bit form measured +2 % (K3 205.5 -> 209.8 us, two more
// spilled registers), in the forward -2 % (140.5 -> 137.6 us), same bits either way (tools/ab_outputs.py)
template <bool O32>
__device__ __forceinline__ const f4* voxel_ptr(const GridD& G, int vox, int g)
{
    if constexpr (O32) {
        const unsigned off = ((unsigned)vox * 32u + 4u * (unsigned)g) * 4u;
        return reinterpret_cast<const f4*>(reinterpret_cast<const char*>(G.v) + off);
    } else return reinterpret_cast<const f4*>(G.v + (size_t)vox * 32 + 4 * g);
}

// gather the 8 channels {4g..4g+3, 16+4g..16+4g+3} of this lane's quarter into two D-layout quads
template <bool O32 = false>
__device__ __forceinline__ void tri_gather(const GridD& G, const Tri& T, int g, f4& c0, f4& c1)
{
    c0 = (f4)(0.f); c1 = (f4)(0.f);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const f4* vp = voxel_ptr<O32>(G, T.vox[c], g);
        f4 a = vp[0], b = vp[4];
        c0 += T.w[c] * a; c1 += T.w[c] * b;
    }
}

// ---- bf16 3-piece split helpers (the scheme is described in nsk_bf16.h) -------------------------------
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

// split fp32 -> three bf16 pieces (round to nearest even each; the residual of one piece feeds the next)
struct B3 { bf8 h, m, l; };
__device__ __forceinline__ B3 split_block(f4 q0, f4 q1)
{
    B3 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = j < 4 ? q0[j] : q1[j - 4];
        const __bf16 h = (__bf16)x;
        const float r1 = x - (float)h;
        const __bf16 m = (__bf16)r1;
        const float r2 = r1 - (float)m;
        r.h[j] = h; r.m[j] = m; r.l[j] = (__bf16)r2;
    }
    return r;
}

struct Frag3 { bf8 h, m, l; };
__device__ __forceinline__ Frag3 load_frag(const bf8* __restrict__ img, int fg, int lane)
{
    const bf8* b = img + (size_t)fg * 3 * 64 + lane;
    Frag3 f; f.h = b[0]; f.m = b[64]; f.l = b[128];
    return f;
}
__device__ __forceinline__ f4 mfma_b(bf8 a, bf8 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// acc[rt] += W[16rt.., block] x  (six bf16 products per output tile)
__device__ __forceinline__ void mac_block(const Frag3& a0, const Frag3& a1, const B3& x, f4 (&acc)[2])
{
    acc[0] = mfma_b(a0.h, x.h, acc[0]); acc[1] = mfma_b(a1.h, x.h, acc[1]);
    acc[0] = mfma_b(a0.h, x.m, acc[0]); acc[1] = mfma_b(a1.h, x.m, acc[1]);
    acc[0] = mfma_b(a0.m, x.h, acc[0]); acc[1] = mfma_b(a1.m, x.h, acc[1]);
    acc[0] = mfma_b(a0.h, x.l, acc[0]); acc[1] = mfma_b(a1.h, x.l, acc[1]);
    acc[0] = mfma_b(a0.l, x.h, acc[0]); acc[1] = mfma_b(a1.l, x.h, acc[1]);
    acc[0] = mfma_b(a0.m, x.m, acc[0]); acc[1] = mfma_b(a1.m, x.m, acc[1]);
}

// ---- fp16 2-piece split helpers (matmul mode 2; the scheme is described in nsk_bf16.h) ------------------
// x = h + l / 2048 with h = fp16(x) (round to nearest even, v_cvt_pk_f16_f32) and l = fp16(2048 (x - h)): 22 significant bits.
// The low pieces are kept scaled by 2^11 so that they never reach fp16's subnormal range before x itself does; the two
// cross products therefore accumulate in their own registers (accL) and join the main sum with one fma per element.
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define NSK_H16_SCALE 2048.f
struct H2 { h8 h, l; };
// Two values at a time: h pair by v_cvt_pk_f16_f32 (round to nearest even); the residual x - h straight from the packed half by v_fma_mix_f32
// (the fp16 operand is widened inside the instruction: exact, as the separate v_cvt_f32_f16 + v_sub_f32 were) and 2048 (x - h) rounded into
// the low / high half of the l pair by v_fma_mixlo_f16 / v_fma_mixhi_f16 (one rounding to nearest even, as v_mul_f32 by a power of two +
// v_cvt_pk_f16_f32 were): 2.5 vector instructions per value instead of 4, the same bits (tools/ab_outputs.py).  hipcc does not form these
// from C (it emits cvt / sub / mul / cvt); inline asm is opaque to its hazard recogniser, hence the explicit s_nop behind the last writer:
// a vector-ALU result needs two wait states before a matrix instruction may read it.
__device__ __forceinline__ void split_pair_h(float x0, float x1, unsigned& hp, unsigned& lp)
{
    typedef _Float16 hh2 __attribute__((ext_vector_type(2)));
    typedef float ff2 __attribute__((ext_vector_type(2)));
    hp = __builtin_bit_cast(unsigned, __builtin_convertvector((ff2){x0, x1}, hh2));
    float t0, t1;
    const float k2048 = NSK_H16_SCALE;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(t0) : "v"(hp), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(t1) : "v"(hp), "v"(x1));
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(lp) : "v"(t0), "s"(k2048));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(lp) : "v"(t1), "s"(k2048));
}
__device__ __forceinline__ H2 split_block_h(f4 q0, f4 q1)
{
    typedef unsigned int u4s __attribute__((ext_vector_type(4)));
    unsigned h0, h1, h2, h3, l0, l1, l2, l3;
    split_pair_h(q0[0], q0[1], h0, l0);
    split_pair_h(q0[2], q0[3], h1, l1);
    split_pair_h(q1[0], q1[1], h2, l2);
    split_pair_h(q1[2], q1[3], h3, l3);
    asm volatile("s_nop 1" : "+v"(l0), "+v"(l1), "+v"(l2), "+v"(l3));
    H2 r;
    r.h = __builtin_bit_cast(h8, (u4s){h0, h1, h2, h3}); r.l = __builtin_bit_cast(h8, (u4s){l0, l1, l2, l3});
    return r;
}
struct FragH { h8 h, l; };
__device__ __forceinline__ FragH load_frag_h(const h8* __restrict__ img, int fg, int lane)
{
    const h8* b = img + (size_t)fg * 2 * 64 + lane;
    FragH f; f.h = b[0]; f.l = b[64];
    return f;
}
__device__ __forceinline__ f4 mfma_h(h8 a, h8 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
// accH[rt] += Wh x_h;  accL[rt] += Wh x_l + Wl x_h  (three fp16 products per output tile, two independent chains)
__device__ __forceinline__ void mac_block_h(const FragH& a0, const FragH& a1, const H2& x, f4 (&accH)[2], f4 (&accL)[2])
{
    accH[0] = mfma_h(a0.h, x.h, accH[0]); accH[1] = mfma_h(a1.h, x.h, accH[1]);
    accL[0] = mfma_h(a0.h, x.l, accL[0]); accL[1] = mfma_h(a1.h, x.l, accL[1]);
    accL[0] = mfma_h(a0.l, x.h, accL[0]); accL[1] = mfma_h(a1.l, x.h, accL[1]);
}
// the pieces of (bit ? x : 0) from the pieces of x: the ReLU backward masks a gradient the chain has just split for the fc[l]^T product,
// and the pieces of a zero are zeros, so eight ANDs replace a second split.  bits: bit j <-> element j of the block
__device__ __forceinline__ H2 mask_block_h(const H2& x, unsigned bits)
{
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    u4 h = __builtin_bit_cast(u4, x.h), l = __builtin_bit_cast(u4, x.l);
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const unsigned lo = 0u - ((bits >> (2 * d)) & 1u), hi = 0u - ((bits >> (2 * d + 1)) & 1u);
        const unsigned k = (lo & 0xffffu) | (hi & 0xffff0000u);
        h[d] &= k; l[d] &= k;
    }
    H2 r;
    r.h = __builtin_bit_cast(h8, h); r.l = __builtin_bit_cast(h8, l);
    return r;
}

// Backward (transposed) image of an MLP decoder in fp16 pieces (two per weight, the low one 2048-fold), K = 32 output features each:
// fragment group FT(l) + rt: fc[l]^T rows 16rt.. (grid features 0..31); WT(l) + rt: pts_linear[l]^T (h part); W0ET / W3ET + rt: the
// e parts of pts_linear[0] / [3] transposed (96 rows, rt < 6).  Used by the frozen decoders' chain (the first 18 groups) and by the
// trainable decoder's chain (all of it).  The fp32 tail is the one of MlpBwdImg (output weight, embedding matrix): same total size.
struct MlpBwdImgH {
    static constexpr int NFG = 30;
    static constexpr int FRAG_BYTES = NFG * 2 * 1024;
    static constexpr int P_WO = FRAG_BYTES / 4;      // float offset of the output weight [4][32]
    static constexpr int P_BM = P_WO + 128;          // embedding B [3][96]
    static constexpr int TOTAL_F = P_BM + 288;
    static constexpr int W0ET = 18, W3ET = 24;
    __host__ __device__ static constexpr int FT(int l) { return 2 * l; }
    __host__ __device__ static constexpr int WT(int l) { return 10 + 2 * (l - 1); }
};
// accH[rt] / accL[rt] += (fragment groups fg0, fg0+1) x   (three fp16 MFMAs per row tile)
__device__ __forceinline__ void gemm_h(const h8* __restrict__ img, int fg0, int lane, const H2& x, f4 (&accH)[2], f4 (&accL)[2])
{
    const FragH a0 = load_frag_h(img, fg0, lane), a1 = load_frag_h(img, fg0 + 1, lane);
    mac_block_h(a0, a1, x, accH, accL);
}
// [96 x 32] transposed e-part product: acc[0..5] += W?ET x, as three 32-row slices
__device__ __forceinline__ void gemm_e_h(const h8* __restrict__ img, int fg0, int lane, const H2& x, f4 (&acc)[6])
{
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        f4 tH[2] = {acc[2 * a], acc[2 * a + 1]}, tL[2] = {(f4)(0.f), (f4)(0.f)};
        gemm_h(img, fg0 + 2 * a, lane, x, tH, tL);
        acc[2 * a] = tH[0] + tL[0] * (1.f / NSK_H16_SCALE); acc[2 * a + 1] = tH[1] + tL[1] * (1.f / NSK_H16_SCALE);
    }
}
// A chain on fp16 pieces runs on a per-sample power-of-two multiple of the upstream gradient (largest component in [2^-4, 2^-3)):
// a sample's gradients span many decades between samples (transmittance), fp16 does not; the chain is linear per sample, so the
// scale is exact and comes off again where a result leaves the chain.  Returns the unscale factor, scales g in place.
template <int OD>
__device__ __forceinline__ float chain_scale(float (&g)[OD])
{
    float m = 0.f;
#pragma unroll
    for (int o = 0; o < OD; ++o) m = fmaxf(m, fabsf(g[o]));
    int e = __builtin_amdgcn_frexp_expf(m);
    e = max(-100, min(100, e));
    const float sc = __builtin_ldexpf(1.f, -e - 3);
#pragma unroll
    for (int o = 0; o < OD; ++o) g[o] *= sc;
    return __builtin_ldexpf(1.f, e + 3);
}

// the same in two steps (loads first, weighting later) so that independent work can sit between them
struct GatherRaw { f4 a[8], b[8]; };
template <bool O32 = false>
__device__ __forceinline__ void tri_gather_issue(const GridD& G, const Tri& T, int g, GatherRaw& R)
{
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const f4* vp = voxel_ptr<O32>(G, T.vox[c], g);
        R.a[c] = vp[0]; R.b[c] = vp[4];
    }
}
__device__ __forceinline__ void tri_gather_reduce(const Tri& T, const GatherRaw& R, f4& c0, f4& c1)
{
    c0 = (f4)(0.f); c1 = (f4)(0.f);
#pragma unroll
    for (int c = 0; c < 8; ++c) { c0 += T.w[c] * R.a[c]; c1 += T.w[c] * R.b[c]; }
}

// ------------------------------------------------------------------------------------------------------
// A-fragment GEMM: acc[rt] += W[16rt.., :] x  for an input of KQ quads (D layout) -- see nsk_layout.h
// ------------------------------------------------------------------------------------------------------
template <int RT, int KQ>
__device__ __forceinline__ void gemm(const f4* __restrict__ img, int quad0, int lane, const f4 (&x)[KQ], f4 (&acc)[RT])
{
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
        f4 a[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) a[r] = img[(quad0 + r * KQ + q) * 64 + lane];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < RT; ++r) acc[r] = mfma4(a[r][i], x[q][i], acc[r]);
    }
}

// [96 x 32] transposed e-part product as three 32-row slices (bounds the fragment registers in flight)
__device__ __forceinline__ void gemm_e(const f4* __restrict__ img, int quad0, int lane, const f4 (&x)[2], f4 (&acc)[6])
{
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        f4 t[2] = {acc[2 * a], acc[2 * a + 1]};
        gemm<2, 2>(img, quad0 + 4 * a, lane, x, t);
        acc[2 * a] = t[0]; acc[2 * a + 1] = t[1];
    }
}

// bias vector [32] in D layout: lane holds features 4g+i and 16+4g+i
__device__ __forceinline__ void load_bias(const float* __restrict__ b, int g, f4 (&acc)[2])
{
    acc[0] = *reinterpret_cast<const f4*>(b + 4 * g);
    acc[1] = *reinterpret_cast<const f4*>(b + 16 + 4 * g);
}

// ReLU in place, returns the 8 "input was > 0" bits (bit r*4+i).  All in integer arithmetic on the bit patterns, three vector
// instructions per element: max_i32(bits, 0) is the ReLU (negative floats, -0 and negative NaNs are negative integers), and the result
// is a non-negative integer that is zero exactly when the ReLU's derivative is, so min_u32(result, 1) is the bit.  (The float form,
// fmaxf + a compare, costs a canonicalising v_max on MFMA outputs, a v_cmp, a v_cndmask and wait states between them.)
__device__ __forceinline__ uint32_t relu_mask(f4 (&a)[2])
{
    uint32_t m = 0;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int y = max(__float_as_int(a[r][i]), 0);
            a[r][i] = __int_as_float(y);
            // m |= min(y, 1) << bit, as exactly two instructions (left to itself the compiler turns the min into a compare and a select,
            // 3.5 instructions per element; their input is the v_max_i32 above, a vector-ALU result: no matrix-pipe hazard inside the asm)
            unsigned t;
            asm("v_min_u32 %0, %1, 1" : "=v"(t) : "v"(y));
            asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(m) : "v"(t), "n"(r * 4 + i), "v"(m));
        }
    return m;
}

// test aid (nsk_debug_preact): the ReLU inputs of layer l as the chain holds them, row = sample, [5][32] floats
__device__ __forceinline__ void dump_preact(float* __restrict__ row, int l, int g, const f4 (&acc)[2])
{
    if (!row) return;
    *reinterpret_cast<f4*>(row + 32 * l + 4 * g) = acc[0];
    *reinterpret_cast<f4*>(row + 32 * l + 16 + 4 * g) = acc[1];
}

template <int CQ>
struct Act {               // activations of one 16-sample tile, D layout
    f4 xe[6];              // sin(pB), k = 16q+4g+i (k>=93: 0)
    f4 xc[CQ];             // grid features (fine: quads 2,3 = grid_middle features)
    f4 h[5][2];            // block outputs
    unsigned long long mask;   // ReLU bits, layer l -> bits 8l..8l+7
};

// sin / cos on the hardware's v_sin_f32 / v_cos_f32, which take the argument in revolutions.  Measured on gfx950 over 4 M arguments
// in [-0.5, 0.5] rev (tools/ubench/vsin.hip): max abs error 1.25e-7 for both.  The argument is reduced in revolutions with a two-term
// split of 1/(2 pi): fma(x, HI, -k) is one rounding of the exact x*HI - k (k <= a few hundred here: |p| * 25 * N(0,1) ~ 1e2), so the
// reduced argument carries at most half an ulp of 0.5 rev = 1.9e-7 rad.  A polynomial sine on the vector ALU costs 15 instructions per
// value against 4 + one quarter-rate transcendental here, and vector instructions do not overlap the bf16 MFMAs of the other wave on a
// SIMD (tools/ubench/interleave.hip), so they add to the kernel time one for one.  (The library sinf carries a Payne-Hanek path.)
__device__ __forceinline__ float nsk_rev(float x)
{
    const float k = rintf(x * 0.15915494309189535f);
    const float f = fmaf(x, 0.159154936671257019f, -k);      // 1/(2 pi) split: 0x1.45f306p-3
    return fmaf(x, 6.42063833e-09f, f);                      //                  + 0x1.b9391p-28
}
__device__ __forceinline__ float nsk_sin(float x) { return __builtin_amdgcn_sinf(nsk_rev(x)); }
__device__ __forceinline__ void nsk_sincos(float x, float& sn, float& cs)
{
    const float f = nsk_rev(x);
    sn = __builtin_amdgcn_sinf(f);
    cs = __builtin_amdgcn_cosf(f);
}

// embedding e = sin(p B) (reference src/models/GaussianFFT.cpp:10-15), optional cos for the backward
template <bool WANT_COS>
__device__ __forceinline__ void embed(const float* __restrict__ Bm /*[3][96]*/, int g, float px, float py, float pz,
                                      f4 (&xe)[6], f4 (&xcos)[6])
{
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        f4 b0 = *reinterpret_cast<const f4*>(Bm + 16 * q + 4 * g);
        f4 b1 = *reinterpret_cast<const f4*>(Bm + 96 + 16 * q + 4 * g);
        f4 b2 = *reinterpret_cast<const f4*>(Bm + 192 + 16 * q + 4 * g);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int k = 16 * q + 4 * g + i;
            // torch::matmul(p, B) (GaussianFFT.cpp:13): the CPU BLAS accumulates the K = 3 products with FMAs in k order
            float s = __builtin_fmaf(pz, b2[i], __builtin_fmaf(py, b1[i], mul_rn(px, b0[i])));
            float sv, cv = 0.f;
            if (WANT_COS) nsk_sincos(s, sv, cv); else sv = nsk_sin(s);
            xe[q][i] = k < NSK_E ? sv : 0.f;
            xcos[q][i] = (WANT_COS && k < NSK_E) ? cv : 0.f;
        }
    }
}

// MLP::forward (reference src/models/MLP.cpp:76-102, intended loop D14): fills A.h and A.mask
template <int CQ, bool DUMP = false>
__device__ __forceinline__ void mlp_forward(const f4* __restrict__ img, int lane, Act<CQ>& A, float* dump = nullptr)
{
    typedef MlpFwdImg<CQ> I;
    const float* imgf = reinterpret_cast<const float*>(img);
    const int g = lane >> 4;
    unsigned long long mask = 0;
    f4 acc[2];
    // block 0
    load_bias(imgf + I::P_B, g, acc);
    gemm<2, 6>(img, I::W0E, lane, A.xe, acc);
    if constexpr (DUMP) dump_preact(dump, 0, g, acc);
    mask |= (unsigned long long)relu_mask(acc);
    { f4 bc[2]; load_bias(imgf + I::P_BC, g, bc); acc[0] += bc[0]; acc[1] += bc[1]; }
    gemm<2, CQ>(img, I::F0, lane, A.xc, acc);
    A.h[0][0] = acc[0]; A.h[0][1] = acc[1];
#pragma unroll
    for (int l = 1; l < 5; ++l) {
        load_bias(imgf + I::P_B + 32 * l, g, acc);
        if (l == 3) {
            gemm<2, 6>(img, I::W3E, lane, A.xe, acc);
            gemm<2, 2>(img, I::W3H, lane, A.h[2], acc);
        } else {
            gemm<2, 2>(img, I::W(l), lane, A.h[l - 1], acc);
        }
        if constexpr (DUMP) dump_preact(dump, l, g, acc);
        mask |= (unsigned long long)relu_mask(acc) << (8 * l);
        { f4 bc[2]; load_bias(imgf + I::P_BC + 32 * l, g, bc); acc[0] += bc[0]; acc[1] += bc[1]; }
        gemm<2, CQ>(img, I::F(l), lane, A.xc, acc);
        A.h[l][0] = acc[0]; A.h[l][1] = acc[1];
    }
    A.mask = mask;
}

// output_linear (32 -> OD) as a wave dot product: partial over this lane's 8 features, summed over g
template <int OD>
__device__ __forceinline__ void mlp_output(const float* __restrict__ Wo, const float* __restrict__ bo, int g,
                                           const f4 (&h4)[2], float (&out)[OD])
{
#pragma unroll
    for (int o = 0; o < OD; ++o) {
        f4 w0 = *reinterpret_cast<const f4*>(Wo + 32 * o + 4 * g);
        f4 w1 = *reinterpret_cast<const f4*>(Wo + 32 * o + 16 + 4 * g);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) s += w0[i] * h4[0][i] + w1[i] * h4[1][i];
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        out[o] = s + bo[o];
    }
}

// MLP_no_xyz::forward (reference src/models/MLP.cpp:165-182, intended loop D15)
struct ActC { f4 xc[2]; f4 h[5][2]; unsigned long long mask; };
__device__ __forceinline__ void coarse_forward(const f4* __restrict__ img, int lane, ActC& A)
{
    typedef CoarseFwdImg I;
    const float* imgf = reinterpret_cast<const float*>(img);
    const int g = lane >> 4;
    unsigned long long mask = 0;
    f4 acc[2];
    const int Wq[5] = {I::W0, I::W1, I::W2, I::W3H, I::W4};
#pragma unroll
    for (int l = 0; l < 5; ++l) {
        load_bias(imgf + I::P_B + 32 * l, g, acc);
        if (l == 0) gemm<2, 2>(img, I::W0, lane, A.xc, acc);
        else {
            if (l == 3) gemm<2, 2>(img, I::W3C, lane, A.xc, acc);
            gemm<2, 2>(img, Wq[l], lane, A.h[l - 1], acc);
        }
        mask |= (unsigned long long)relu_mask(acc) << (8 * l);
        A.h[l][0] = acc[0]; A.h[l][1] = acc[1];
    }
    A.mask = mask;
}

// ------------------------------------------------------------------------------------------------------
// K2: decoder forward over flat 16-sample tiles.  Persistent workgroups of 8 waves; the decoder's forward
// image lives in LDS for the lifetime of the workgroup.  WHICH: 0 coarse, 1 middle, 2 fine, 3 color.
// Sample m = ray (m / S), slot (m % S): p = o + d z[m]; or, when pts != nullptr, p = pts[m] (eval_points).
// ------------------------------------------------------------------------------------------------------
struct DecArgs {
    const float* rays_o; const float* rays_d; const float* z; const float* pts;
    const int* perm;          // cell-sorted sample order (k_sort_place): tile slot t -> sample perm[t]; nullptr = identity
    int M, S;
    unsigned S_magic;         // ceil(2^32 / S) (0 for S = 1): see ray_of; the host keeps M below 2^26
    float bound[6];
    GridD grid, grid_mid;
    const f4* img;            // forward image (global)
    const void* img16;        // bf16 3-piece forward image (nsk_bf16.h) or nullptr
    const f4* bimg;           // backward image (global)
    const void* bimg16;       // bf16 3-piece backward image (MlpBwdImgB) or nullptr
    int img_f4;               // forward image size in f4
    float* out;               // occupancy [M] (which<3) or rgb4 [M][4] (color)
    unsigned long long* masks;   // [M][4] ReLU bits or nullptr
    f4* hsave;                // block outputs h0..h4 of a trainable decoder, [tile][5][2][64 lanes] f4 (lane order = D layout), or nullptr
    // backward only
    const float* g_raw;       // [M][4] = (g_rgb[3], g_sigma)
    float* g_rays_o; float* g_rays_d;    // [N][3] accumulated with atomics, or nullptr
    float* g_dec;             // canonical decoder gradient (trainable) or nullptr
    unsigned flags;
    float* dump;              // test aid (nsk_debug_preact): ReLU inputs [M][5][32] by sample, or nullptr
    int skew;                 // start offset of a workgroup's upper four waves, in units of 1024 cycles (wave_skew)
    const float* dyn_resid;   // k_decode_bwd_track: per-ray |gt - depth| [dyn_n] (+inf: ray masked) -- the Tracker's median mask is applied HERE (see lower_median_x10)
    int dyn_n;
};

// The two waves that share a SIMD (w and w + 4 of a 512-thread workgroup) start a kernel in step, and a tile is a long vector-unit
// phase (sample, gather, embedding, operand split) followed by a long matrix-core phase; in step they queue on the same unit in
// both phases while the other unit idles (counters: MFMA busy 28 %, VALU 55 %, co-execution 11 % of the MFMA cycles).  Holding the
// upper four waves back once, by about one vector phase, lets one wave's MFMA chain run under the other's vector work from then on.
__device__ __forceinline__ void wave_skew(const DecArgs& A, int wave, int nw)
{
    const int mode = A.skew / 100, n = A.skew % 100;
    const bool late = mode == 0 ? wave >= nw / 2 : (mode == 1 ? (wave & 1) : ((wave >> 1) & 1));
    if (late)
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(16);
}

// ray of sample mm = mm / S without the generic division (~25 vector instructions): umulhi by ceil(2^32 / S), exact for mm < 2^32 / S
// a load at base + a 32-bit BYTE offset: scalar base, one unsigned vector offset -- no sign extension and no 64-bit shift-add per access (an
// `int` index costs v_ashrrev + v_lshl_add_u64; k_decode_bwd_multi held ~800 + ~1000 of them).  The caller's offset must stay below 2^32: the
// host keeps M below 2^26 samples, so sample-indexed arrays of up to 32 bytes per sample qualify (z 4, g_raw 16, ReLU bits 32, perm 4, rays 12).
template <typename T>
__device__ __forceinline__ T ld32(const void* base, unsigned byte_off) { return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off); }
__device__ __forceinline__ int ray_of(const DecArgs& A, int mm) { return A.S_magic ? (int)__umulhi((unsigned)mm, A.S_magic) : mm; }
struct __attribute__((packed, aligned(4))) Ray3 { float x, y, z; };
__device__ __forceinline__ void sample_point(const DecArgs& A, int mm, float& px, float& py, float& pz, float& zz, int& n)
{
    if (A.pts) { px = A.pts[3 * mm]; py = A.pts[3 * mm + 1]; pz = A.pts[3 * mm + 2]; zz = 0.f; n = 0; return; }
    NSK_IDX(1, mm, A.M);
    n = ray_of(A, mm);           // = mm / A.S (the generic division is ~25 vector instructions)
    NSK_IDX(4, n, (A.M + A.S - 1) / A.S);
    zz = A.z[mm];
    const Ray3 o = *reinterpret_cast<const Ray3*>(A.rays_o + 3 * n), d = *reinterpret_cast<const Ray3*>(A.rays_d + 3 * n);
    px = add_rn(o.x, mul_rn(d.x, zz));           // reference src/Renderer.cpp:121
    py = add_rn(o.y, mul_rn(d.y, zz));
    pz = add_rn(o.z, mul_rn(d.z, zz));
}

// the same in two steps, for software pipelining: the loads (no arithmetic on their results) and the point.
// (Round 3 tried one 16-byte record (px, py, pz, z) per sample written by k_sample, re-ordered by k_sort_place, instead of these seven loads:
// the forward got 7 us SLOWER at K3 on the same box, sampling + sort 6 us slower -- the ray arrays are cache-resident, the records stream --
// and parity moved: forming p in another kernel changes whether the compiler fuses o + d z, p moves by an ulp, p.B by ~1e-5 rad, and
// ReLUs within that of zero flip: K3 colour-grid gradient 3.5e-4 -> 1.15e-3 from the oracle, the very figure round 2's 32-bit addressing
// experiment met; the index audit of tools/exp_idx.py is clean, so that discrepancy was this rounding effect, not an out-of-range read.)
struct SampleRaw { float z, o[3], d[3]; };

__device__ __forceinline__ void sample_load(const DecArgs& A, int mm, SampleRaw& R)
{
    if (A.pts) { R.o[0] = A.pts[3 * mm]; R.o[1] = A.pts[3 * mm + 1]; R.o[2] = A.pts[3 * mm + 2]; R.z = 0.f; R.d[0] = R.d[1] = R.d[2] = 0.f; return; }
    NSK_IDX(1, mm, A.M);
    const int n = ray_of(A, mm);     // = mm / A.S
    NSK_IDX(4, n, (A.M + A.S - 1) / A.S);
    R.z = ld32<float>(A.z, (unsigned)mm * 4u);
    // a ray's origin and direction as ONE 12-byte load each (global_load_dwordx3 needs dword alignment only): the 16 samples of a cell-sorted tile
    // come from 16 rays, so every load instruction touches 16 cache lines, and seven of them per tile and wave kept the CU's address unit busy
    // for ~2 500 cycles of a trainable iteration (tools/exp_ph3.py: the stage_a segment); three do the same work
    const Ray3 o = ld32<Ray3>(A.rays_o, (unsigned)n * 12u), d = ld32<Ray3>(A.rays_d, (unsigned)n * 12u);
    R.o[0] = o.x; R.o[1] = o.y; R.o[2] = o.z; R.d[0] = d.x; R.d[1] = d.y; R.d[2] = d.z;
}
__device__ __forceinline__ void sample_finish(const DecArgs& A, const SampleRaw& R, float& px, float& py, float& pz)
{
    if (A.pts) { px = R.o[0]; py = R.o[1]; pz = R.o[2]; return; }
    px = add_rn(R.o[0], mul_rn(R.d[0], R.z));           // reference src/Renderer.cpp:121
    py = add_rn(R.o[1], mul_rn(R.d[1], R.z));
    pz = add_rn(R.o[2], mul_rn(R.d[2], R.z));
}

// sample handled by tile slot `slot` (clamped into [0, M)): the cell-sorted order when the launch has one, else the slot itself.
// Per-sample state that only the decoders exchange (ReLU bits, saved block outputs) is indexed by SLOT, so it stays coalesced;
// what the per-ray kernels read or write (z, occupancy, colour, g_raw) is indexed by SAMPLE.
__device__ __forceinline__ int slot_sample(const DecArgs& A, int slot)
{
    const int s = min(slot, A.M - 1);
    NSK_IDX(2, s, A.M);
#ifdef NSK_EXPERIMENT
    if (A.perm) { const int v = A.perm[s]; NSK_IDX(0, v, A.M); return v; }
#endif
    return A.perm ? ld32<int>(A.perm, (unsigned)s * 4u) : s;
}

// Tile schedule shared by the decoder kernels: wave `wg` of `nw` takes blocks of 2^sh consecutive tiles, dealt round-robin over the
// waves.  Single tiles (sh = 0) everywhere: dealing keeps the waves' loads equal (tiles in sparsely sampled space cost the scatter
// one flush per sample, tiles inside a well-sampled cell one per tile; waves that owned a contiguous range of the former ran 100 us
// behind), and blocks of 4 left up to 13 % of the forward's waves idle at 5000 rays.
__device__ __forceinline__ int tile_shift(int, int) { return 0; }
__device__ __forceinline__ int tile_of(int k, int wg, int nw, int sh) { return ((((k >> sh) * nw + wg)) << sh) + (k & ((1 << sh) - 1)); }
__device__ __forceinline__ int tiles_per_wave(int ntasks, int nw, int sh) { return ((((ntasks + (1 << sh) - 1) >> sh) + nw - 1) / nw) << sh; }

// The backward of a TRAINABLE decoder needs the block outputs h0..h4 as the X operands of its weight gradients.  The
// forward stores them (one coalesced KiB per quad and tile) instead of the backward recomputing the MLP: 640 B per sample
// of extra traffic against 277 MFMAs + two LDS image swaps per tile.
__device__ __forceinline__ void save_h(f4* __restrict__ hsave, int task, int lane, const f4 (&h)[5][2])
{
    f4* dst = hsave + ((size_t)task * 10) * 64 + lane;
#pragma unroll
    for (int l = 0; l < 5; ++l) { dst[(2 * l) * 64] = h[l][0]; dst[(2 * l + 1) * 64] = h[l][1]; }
}

// bid / nb: this workgroup's index and the number of workgroups working on this decoder (a launch may serve
// several decoders, each with its own slice of the grid: k_decode_fwd_multi)
template <int WHICH, int NW = 8, bool DUMP = false>
__device__ __forceinline__ void decode_fwd_body(const DecArgs& A, int bid, int nb)
{
    extern __shared__ __attribute__((aligned(16))) f4 smem[];
    copy_image_to_lds<64 * NW>(smem, A.img, A.img_f4);
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    const float* imgf = reinterpret_cast<const float*>(smem);
    const int ntasks = (A.M + 15) >> 4;
    const int tsh = tile_shift(ntasks, nb * NW);
    const int kmax = tiles_per_wave(ntasks, nb * NW, tsh);
    for (int k = 0; k < kmax; ++k) {
        const int task = tile_of(k, bid * NW + wave, nb * NW, tsh);
        if (task >= ntasks) break;
        asm volatile("" ::: "memory");      // keep the LDS fragment reads inside the loop (LICM would hoist + spill them)
        const int slot = task * 16 + j;
        const bool valid = slot < A.M;
        const int mm = slot_sample(A, slot);
        const int m = mm;
        float px, py, pz, zz; int n;
        sample_point(A, mm, px, py, pz, zz, n);
        Tri T;
        tri_setup(A.grid, A.bound, px, py, pz, T);
        if constexpr (WHICH == 0) {
            ActC C;
            tri_gather(A.grid, T, g, C.xc[0], C.xc[1]);
            coarse_forward(smem, lane, C);
            float out[1];
            mlp_output<1>(imgf + CoarseFwdImg::P_WO, imgf + CoarseFwdImg::P_BO, g, C.h[4], out);
            if (valid) {
                if (g == 0) A.out[m] = out[0];
                if (A.masks) A.masks[(size_t)slot * 4 + g] = C.mask;
            }
            if (A.hsave) save_h(A.hsave, task, lane, C.h);
        } else {
            constexpr int CQ = WHICH == 2 ? 4 : 2;
            constexpr int OD = WHICH == 3 ? 4 : 1;
            typedef MlpFwdImg<CQ> I;
            Act<CQ> C;
            tri_gather(A.grid, T, g, C.xc[0], C.xc[1]);
            if constexpr (WHICH == 2) {                     // reference MLP.cpp:79-84 concat_feat
                Tri Tm;
                tri_setup(A.grid_mid, A.bound, px, py, pz, Tm);
                tri_gather(A.grid_mid, Tm, g, C.xc[CQ - 2], C.xc[CQ - 1]);
            }
            f4 dummy[6];
            embed<false>(imgf + I::P_BM, g, px, py, pz, C.xe, dummy);
            float* dump = nullptr;
            if constexpr (DUMP) dump = valid ? A.dump + (size_t)m * 160 : nullptr;
            mlp_forward<CQ, DUMP>(smem, lane, C, dump);
            float out[OD];
            mlp_output<OD>(imgf + I::P_WO, imgf + I::P_BO, g, C.h[4], out);
            if (valid) {
                if (g == 0) {
                    if constexpr (OD == 4) *reinterpret_cast<f4*>(A.out + (size_t)m * 4) = (f4){out[0], out[1], out[2], out[OD - 1]};
                    else A.out[m] = out[0];
                }
                if (A.masks) A.masks[(size_t)slot * 4 + g] = C.mask;
            }
            if (A.hsave) save_h(A.hsave, task, lane, C.h);
        }
    }
}

template <int WHICH>
__global__ __launch_bounds__(512) void k_decode_fwd(DecArgs A) { decode_fwd_body<WHICH>(A, blockIdx.x, gridDim.x); }

// one launch for all decoders of a stage: workgroups [wg_end[r-1], wg_end[r]) serve decoder which[r]
struct MultiArgs {
    DecArgs a[3]; int which[3]; int train[3]; int wg_end[3]; int n;
    const float* sum_src; float* sum_dst; int sum_n;      // optional: one extra workgroup sums the per-ray losses (saves a launch)
    ScanArgs scan;                                        // optional (scan.nblocks > 0): the NEXT batch's cell-sort offsets ride behind the roles (nsk_map_prepare)
    const float* dyn_seed; float* dyn_thr_out;            // k_decode_bwd_track's last workgroup: the compositing's own d/d rays_d term [N][3], 10 x median (scalar)
};

__device__ __forceinline__ void block_sum(const float* __restrict__ x, int n, float* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) f4 smem[];      // dynamic LDS only: a static array would push the kernel past 160 KiB
    float* sh = reinterpret_cast<float*>(smem);
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += x[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { for (int w = 1; w < (int)(blockDim.x >> 6); ++w) s += sh[w]; *out = s; }
}
// (a 1024-thread form, 4 waves per SIMD at 128 VGPRs, measured 126 us against 73 us for this one at 1000 rays)
#ifdef NSK_EXPERIMENT
// per-workgroup (start, end, role, xcc) stamps of the last fwd [0] / bwd [1] multi launch (tools/exp_ts.py)
__device__ unsigned long long nsk_dbg_ts[2][1024][4];
#define NSK_TS_BEGIN(K) unsigned long long ts0_ = wall_clock64()
#define NSK_TS_END(K, role) do { __syncthreads(); if (threadIdx.x == 0 && blockIdx.x < 1024) { unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); \
    nsk_dbg_ts[K][blockIdx.x][0] = ts0_; nsk_dbg_ts[K][blockIdx.x][1] = wall_clock64(); nsk_dbg_ts[K][blockIdx.x][2] = (role); nsk_dbg_ts[K][blockIdx.x][3] = xcc & 15; } } while (0)
#else
#define NSK_TS_BEGIN(K)
#define NSK_TS_END(K, role)
#endif

__global__ __launch_bounds__(512) void k_decode_fwd_multi(MultiArgs MA)
{
    NSK_TS_BEGIN(0);
    int r = 0;
    while (r < MA.n - 1 && (int)blockIdx.x >= MA.wg_end[r]) ++r;
    const int b0 = r == 0 ? 0 : MA.wg_end[r - 1];
    const int bid = blockIdx.x - b0, nb = MA.wg_end[r] - b0;
    switch (MA.which[r]) {
    case 0: decode_fwd_body<0, 8>(MA.a[r], bid, nb); break;
    case 1: decode_fwd_body<1, 8>(MA.a[r], bid, nb); break;
    case 2: decode_fwd_body<2, 8>(MA.a[r], bid, nb); break;
    default: decode_fwd_body<3, 8>(MA.a[r], bid, nb); break;
    }
    NSK_TS_END(0, r);
}

// ------------------------------------------------------------------------------------------------------
// K3: compositing (reference include/torchlib/utils.h:148-172) + in-bound override (src/Renderer.cpp:26-36).
// One wave per ray, lane = sample.  mode 0: forward only.
// Backward modes produce g_raw [M][4] = (w*g_rgb, g_sigma) (g_sigma = 0 for out-of-bound samples) and the
// seed of g_rays_d from the |d| term:  1 = upstream arrays, 2 = Mapper loss, 3 = Tracker loss.
// ------------------------------------------------------------------------------------------------------
struct CompArgs {
    RParams R;
    int N, S, stage;
    const float* rays_o; const float* rays_d; const float* z;
    const float* occ_a; const float* occ_b;     // occupancy parts: stage0 coarse | 1 middle | 2,3 middle + fine
    const float* rgb4;                          // color decoder output [M][4] (stage 3) or nullptr
    float* rgb; float* depth; float* var; float* weights;      // forward outputs (may be nullptr)
    int mode;
    const float* g_rgb; const float* g_depth; const float* g_var;            // mode 1
    const float* gt_depth; const float* gt_color; float w_color; int use_color;   // modes 2,3
    const float* thr; int handle_dynamic; int detach_var;                     // mode 3 (thr: device scalar)
    float* resid; unsigned* bar; float* thr_out;                              // mode 4: per-ray |gt - depth| [N], grid barrier {arrived, left, timed out}, 10 x median
    float* loss;                                                              // modes 2,3: per-ray loss [N] or nullptr
    float* g_raw;                                                             // [M][4]
    float* g_rays_o; float* g_rays_d;                                         // [N][3] seeds or nullptr
    float* g_seed;                                                            // mode 5: [N][3] the d/d rays_d term of the compositing itself (g_rays_d gets 0)
    const uint8_t* keep;                                                      // [N] or nullptr: rays with keep == 0 are rendered but take no part in
                                                                              // the loss or its gradient (nsk_set_ray_mask: the reference drops them)
};

__device__ __forceinline__ float sgnf(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

// mode 0: forward only; 1: backward from given output gradients; 2: Mapper loss; 3: Tracker loss with the median threshold read from
// A.thr; 4: Tracker loss with the threshold computed HERE -- the residuals of all rays meet at a grid barrier (every workgroup
// is resident: the host uses this mode only while the grid has at most one workgroup per CU), each workgroup then finds
// 10 x the lower median by rank counting.  One launch instead of composite + k_median_thr + composite (Tracker.cpp:67-71).
#define NSK_MEDIAN_FUSED_MAX 1024         // rays (LDS copy of the residuals; the host also caps the grid at one workgroup per CU)
// The target-th smallest of rs[0..n4) (all >= 0, padding = +inf) by bisection on the bit pattern (non-negative floats order like their
// patterns): 31 counting passes over a few registers per thread, one barrier each; every thread of the NT-thread workgroup must call it.
// Until round 3 every workgroup of k_composite mode 4 ranked every residual against every other -- N^2 / 4 LDS reads per workgroup, 200 us
// of that launch at 1000 rays (tools/track_times.py).
template <int NT, int MAXN>
__device__ __forceinline__ float median_select(const float* __restrict__ rs, int n4, int target)
{
    constexpr int PER = (MAXN + NT - 1) / NT;
    __shared__ int part[2][NT / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned u[PER];
#pragma unroll
    for (int e = 0; e < PER; ++e) { const int i = threadIdx.x + e * NT; u[e] = i < n4 ? __float_as_uint(rs[i]) : 0xffffffffu; }
    unsigned ans = 0u;
    for (int bit = 30; bit >= 0; --bit) {
        const unsigned cand = ans | (1u << bit);
        int cnt = 0;
#pragma unroll
        for (int e = 0; e < PER; ++e) cnt += u[e] < cand ? 1 : 0;
        cnt = (int)wave_sum((float)cnt);
        if (lane == 0) part[bit & 1][wave] = cnt;
        __syncthreads();
        int tot = 0;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) tot += part[bit & 1][w];
        if (tot <= target) ans = cand;
    }
    return __uint_as_float(ans);
}

// Mode 5 (round 4): the Tracker's loss with the median mask DEFERRED to the backward launch.  The mask `|gt - depth| < 10 median` (Tracker.cpp:69-71)
// is one factor per ray on everything this kernel hands on -- g_raw, the per-ray loss, the compositing's own d/d rays_d term -- so this kernel writes
// them as if every ray passed, plus the residuals, and needs no grid barrier (mode 4's cost 12 of its 21 us at 200 rays); every workgroup of
// k_decode_bwd_track then finds the threshold from the residuals while its weight image and first tile are in flight, and zeroes g_raw of the
// rays that fail.  No inter-workgroup protocol is left: the residuals are ordinary stores of the previous launch.
// 10 x the lower median (torch.median) of the residuals in rs[0..n4) (non-negative or NaN; +inf = masked ray, padding); every thread of the
// NT-thread workgroup calls it after a barrier that made rs visible, and every thread gets the value.  part: 3 * NT / 64 ints of LDS.
template <int NT, int MAXN>
__device__ __forceinline__ float lower_median_x10(const float* __restrict__ rs, int n4, int* __restrict__ part)
{
    constexpr int NWV = NT / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // Counting is by ballot (a compare into a scalar pair + s_bcnt1): a first version summed per-lane counts with wave_sum and met at a barrier
    // in each of the 32 passes -- 12 us on every workgroup of the Tracker's backward (tools/track_times.py: 15.4 -> 27.2 us).
    if (n4 <= 256) {
        // up to 256 rays (the Tracker's 200): every WAVE selects on its own -- four residuals per lane, 4 compares + 4 s_bcnt1 per pass, no LDS
        // traffic and no barrier inside the 32 passes
        unsigned u[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { const int i = lane + 64 * e; u[e] = i < n4 ? __float_as_uint(rs[i]) : 0xffffffffu; }
        auto below = [&](unsigned cand) {
            int c = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) c += __popcll(__builtin_amdgcn_ballot_w64(u[e] < cand));
            return c;
        };
        const int target = (max(below(0x7f800000u), 1) - 1) / 2;      // finite residuals: masked rays and NaN renderings sort behind them
        unsigned ans = 0u;
        for (int bit = 30; bit >= 0; --bit) {
            const unsigned cand = ans | (1u << bit);
            if (below(cand) <= target) ans = cand;
        }
        return 10.f * __uint_as_float(ans);
    }
    constexpr int PER = (MAXN + NT - 1) / NT;
    unsigned u[PER];
#pragma unroll
    for (int e = 0; e < PER; ++e) { const int i = threadIdx.x + e * NT; u[e] = i < n4 ? __float_as_uint(rs[i]) : 0xffffffffu; }
    auto below = [&](unsigned cand, int buf) {                 // how many residuals have a bit pattern below cand (all threads get the total)
        int cnt = 0;
#pragma unroll
        for (int e = 0; e < PER; ++e) cnt += __popcll(__builtin_amdgcn_ballot_w64(u[e] < cand));
        if (lane == 0) part[buf * NWV + wave] = cnt;
        __syncthreads();
        int tot = 0;
#pragma unroll
        for (int w = 0; w < NWV; ++w) tot += part[buf * NWV + w];
        return tot;
    };
    const int valid = below(0x7f800000u, 2);
    const int target = (max(valid, 1) - 1) / 2;
    unsigned ans = 0u;
    for (int bit = 30; bit >= 0; --bit) {
        const unsigned cand = ans | (1u << bit);
        if (below(cand, bit & 1) <= target) ans = cand;
    }
    return 10.f * __uint_as_float(ans);
}

template <int RPW>           // rays (waves) per workgroup; bid / nb: this role's workgroup index and count inside the launch
__device__ __forceinline__ void composite_body(const CompArgs& A, int bid, int nb)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int n = bid * RPW + wave;
    if (n >= A.N) {
        if (A.mode != 4) return;
        n = A.N - 1;                       // mode 4 has workgroup barriers: a spare wave of the last workgroup repeats the last ray (same stores)
    }
    const int S = A.S;
    const bool act = lane < S;
    const size_t m = (size_t)n * S + (act ? lane : S - 1);
    const float ox = A.rays_o[3 * n], oy = A.rays_o[3 * n + 1], oz = A.rays_o[3 * n + 2];
    const float dx = A.rays_d[3 * n], dy = A.rays_d[3 * n + 1], dz = A.rays_d[3 * n + 2];
    const float z = A.z[m];
    const float px = add_rn(ox, mul_rn(dx, z)), py = add_rn(oy, mul_rn(dy, z)), pz = add_rn(oz, mul_rn(dz, z));
    const float* b = A.R.bound;
    const bool inb = px < b[1] && px > b[0] && py < b[3] && py > b[2] && pz < b[5] && pz > b[4];
    float occ = A.occ_a[m];
    if (A.occ_b) occ = A.occ_b[m] + occ;                       // fine_occ + middle_occ (NICE.cpp:40,49)
    const float sg = inb ? occ : 100.f;                        // Renderer.cpp:36
    f4 col = (f4)(0.f);
    if (A.rgb4) col = *reinterpret_cast<const f4*>(A.rgb4 + m * 4);
    const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
    float znext = __shfl_down(z, 1);
    const float dzv = (lane + 1 < S) ? (znext - z) : 1e10f;
    const float dist = dzv * nrm;
    float alpha, ex = 0.f;
    if (A.R.occupancy) alpha = 1.f / (1.f + expf(-10.f * sg));
    else { ex = expf(-fmaxf(sg, 0.f) * dist); alpha = 1.f - ex; }
    if (!act) alpha = 0.f;
    // T = exclusive prefix product of (1 - alpha + 1e-10)
    float fct = act ? (1.f - alpha + 1e-10f) : 1.f;
    float incl = fct;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float up = __shfl_up(incl, o);
        if (lane >= o) incl *= up;
    }
    float T = __shfl_up(incl, 1);
    if (lane == 0) T = 1.f;
    const float w = act ? alpha * T : 0.f;
    const float D = wave_sum(w * z);
    const float cr = wave_sum(w * col[0]), cg = wave_sum(w * col[1]), cb = wave_sum(w * col[2]);
    const float dzD = z - D;
    const float V = wave_sum(w * dzD * dzD);
    if (A.weights && act) A.weights[m] = w;
    if (lane == 0) {
        if (A.depth) A.depth[n] = D;
        if (A.var) A.var[n] = V;
        if (A.rgb) { A.rgb[3 * n] = cr; A.rgb[3 * n + 1] = cg; A.rgb[3 * n + 2] = cb; }
    }
    if (A.mode == 0) return;
    const bool kept = !A.keep || A.keep[n];
    float thr_here = 0.f;
    if (A.mode == 5 && lane == 0) A.resid[n] = kept ? fabsf(A.gt_depth[n] - D) : NSK_INF;
    if (A.mode == 4) {
        __shared__ __attribute__((aligned(16))) float rs[NSK_MEDIAN_FUSED_MAX];
        __shared__ float s_thr;
        __shared__ int s_valid;
        // The residuals travel as device-scope atomics (they are written through to the coherence point) and the counter is bumped only
        // after this workgroup's stores have been acknowledged (vmcnt): no release / acquire fence, which on this part writes back and
        // invalidates the whole L2 of the XCD (measured: the launch took 22 us with __threadfence(), three launches took 8)
        // (round 3 tried the residual slots themselves as the barrier -- -1 between launches, every thread polling the slot it copies: 24 us
        // against 21 at 200 rays, 44 against 27 at 1000: ten thousand pollers on the same few lines cost more than the two round trips saved)
        if (lane == 0) __hip_atomic_store(&A.resid[n], kept ? fabsf(A.gt_depth[n] - D) : NSK_INF, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // masked rays sort to the end (k_median_thr)
        if (threadIdx.x == 0) { s_thr = NSK_INF; s_valid = 0; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            // grid barrier: device-scope counter; bounded wait (a grid that is not resident as a whole must not hang the GPU: it leaves
            // with the threshold at infinity and the flag bar[2] set, which nsk_sync reports)
            __hip_atomic_fetch_add(&A.bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (__hip_atomic_load(&A.bar[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nb) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 20)) { A.bar[2] = 1u; break; }
            }
        }
        __syncthreads();
        const int N4 = (A.N + 3) & ~3;
        for (int i = threadIdx.x; i < N4; i += 64 * RPW) {
            const float v = i < A.N ? __hip_atomic_load(&A.resid[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : NSK_INF;
            rs[i] = v;
            if (v < NSK_INF) atomicAdd(&s_valid, 1);
        }
        __syncthreads();
        const int target = (max(s_valid, 1) - 1) / 2;          // torch.median: the lower median of the valid residuals
        const float med = median_select<64 * RPW, NSK_MEDIAN_FUSED_MAX>(rs, N4, target);
        if (threadIdx.x == 0) s_thr = 10.f * med;
        __syncthreads();
        thr_here = s_thr;
        if (threadIdx.x == 0) {
            if (bid == 0 && A.thr_out) *A.thr_out = thr_here;
            const unsigned old = __hip_atomic_fetch_add(&A.bar[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == (unsigned)nb - 1) {         // the last workgroup to leave re-arms the barrier for the next launch
                __hip_atomic_store(&A.bar[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&A.bar[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    // ---- seed gradients -------------------------------------------------------------------------------
    float gD = 0.f, gV = 0.f, gC[3] = {0.f, 0.f, 0.f};
    if (!kept) {
        // a masked ray contributes exactly nothing, whatever its rendering came out as (a frame that looks out of the bound renders
        // inf / NaN, and 0 * NaN would otherwise reach the grids: tests/test_gpu_configs.py::test_fully_masked_batch_is_a_no_op)
        if (A.loss && lane == 0) A.loss[n] = 0.f;
        if (act) *reinterpret_cast<f4*>(A.g_raw + m * 4) = (f4)(0.f);
        if (A.g_rays_d && lane < 3) { A.g_rays_d[3 * n + lane] = 0.f; A.g_rays_o[3 * n + lane] = 0.f; if (A.mode == 5) A.g_seed[3 * n + lane] = 0.f; }
        return;
    } else if (A.mode == 1) {
        gD = A.g_depth ? A.g_depth[n] : 0.f;
        gV = A.g_var ? A.g_var[n] : 0.f;
        gC[0] = A.g_rgb[3 * n]; gC[1] = A.g_rgb[3 * n + 1]; gC[2] = A.g_rgb[3 * n + 2];
    } else {
        const float gtd = A.gt_depth[n];
        const float r = gtd - D;
        float rc[3] = {0.f, 0.f, 0.f};
        if (A.use_color) { rc[0] = A.gt_color[3 * n] - cr; rc[1] = A.gt_color[3 * n + 1] - cg; rc[2] = A.gt_color[3 * n + 2] - cb; }
        float lsum = 0.f;
        if (A.mode == 2) {                                      // Mapper.cpp:435-442
            if (gtd > 0.f) { lsum += fabsf(r); gD = -sgnf(r); }
            if (A.use_color) {
                lsum += A.w_color * (fabsf(rc[0]) + fabsf(rc[1]) + fabsf(rc[2]));
                for (int k = 0; k < 3; ++k) gC[k] = -A.w_color * sgnf(rc[k]);
            }
        } else {                                                // Tracker.cpp:67-82
            bool mk = gtd > 0.f && (!A.handle_dynamic || A.mode == 5 || fabsf(r) < (A.mode == 4 ? thr_here : *A.thr));
            if (mk) {
                float u = sqrtf(V + 1e-10f);
                lsum += fabsf(r) / u;
                gD = -sgnf(r) / u;
                if (!A.detach_var) gV = -fabsf(r) / (2.f * u * u * u);
                if (A.use_color) {
                    lsum += A.w_color * (fabsf(rc[0]) + fabsf(rc[1]) + fabsf(rc[2]));
                    for (int k = 0; k < 3; ++k) gC[k] = -A.w_color * sgnf(rc[k]);
                }
            }
        }
        if (A.loss && lane == 0) A.loss[n] = lsum;               // per-ray loss, summed by k_sum
    }
    // ---- backward of the compositing (SURVEY.md 8a "Backward formulas") -------------------------------
    gD += gV * -2.f * wave_sum(w * dzD);
    float v = gD * z + gV * dzD * dzD + gC[0] * col[0] + gC[1] * col[1] + gC[2] * col[2];
    float vw = act ? v * w : 0.f;
    // exclusive suffix sum of v*w
    float suf = vw;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float dn = __shfl_down(suf, o);
        if (lane + o < 64) suf += dn;
    }
    suf -= vw;
    float g_alpha = v * T - suf / (1.f - alpha + 1e-10f);
    float g_sigma, g_n = 0.f;
    if (A.R.occupancy) g_sigma = g_alpha * 10.f * alpha * (1.f - alpha);
    else {
        float rs = fmaxf(sg, 0.f);
        g_sigma = sg > 0.f ? g_alpha * dist * ex : 0.f;
        g_n = g_alpha * rs * ex * dzv;
    }
    if (!inb) g_sigma = 0.f;
    if (act) *reinterpret_cast<f4*>(A.g_raw + m * 4) = (f4){w * gC[0], w * gC[1], w * gC[2], g_sigma};
    if (A.g_rays_d) {
        float gn = wave_sum(act ? g_n : 0.f);
        if (lane < 3) {
            float dk = lane == 0 ? dx : (lane == 1 ? dy : dz);
            const float seed = nrm > 0.f ? gn * dk / nrm : 0.f;              // utils.h:153 norm(rays_d)
            if (A.mode == 5) { A.g_seed[3 * n + lane] = seed; A.g_rays_d[3 * n + lane] = 0.f; }      // added by k_decode_bwd_track's last workgroup if the ray passes
            else A.g_rays_d[3 * n + lane] = seed;
            A.g_rays_o[3 * n + lane] = 0.f;
        }
    }
}
__global__ __launch_bounds__(256) void k_composite(CompArgs A) { composite_body<4>(A, blockIdx.x, gridDim.x); }
// The Mapper's compositing with the NEXT batch's sampling behind it in the same launch (nsk_map_prepare): the sampling's chain of dependent
// round trips runs beside the compositing's instead of in front of the next step's forward.  Eight rays per workgroup for both roles.
static_assert(NSK_SAMPLE_RAYS == 8, "k_composite_sample: both roles use 512-thread workgroups");
__global__ __launch_bounds__(512) void k_composite_sample(CompArgs A, SampArgs P, int comp_blocks)
{
    if ((int)blockIdx.x < comp_blocks) composite_body<8>(A, blockIdx.x, comp_blocks);
    else sample_body(P, (int)blockIdx.x - comp_blocks);
}

// ------------------------------------------------------------------------------------------------------
// scatter-add of a 16-sample tile's feature gradient into the grid gradient (voxel-major).
//
// Samples of a tile that share a cell form a run (consecutive along a ray; whole tiles in cell-sorted order).  The sums
//     out[run][corner][channel] = sum over the run's samples j of  w[j][corner] * g_c[j][channel]
// are OUTER PRODUCTS accumulated over the run: one v_mfma_f32_4x4x1_16b_f32 per sample (sixteen 4 x 4 blocks = 8 corners x 32 channels, two
// passes of the matrix pipe), chained through its accumulator while the run lasts.  Block b = lane >> 2 is (corner half cg = b >> 3, channel
// quad ch = b & 7); a lane feeds w[4 cg + (lane & 3)] as the A row and g_c[4 ch + (lane & 3)] as the B column and receives, in result
// register i, the sum for corner 4 cg + i and channel 4 ch + (lane & 3): so each of a run's four atomic wave-instructions adds two whole
// 128-byte voxel rows.  Operands for all sixteen samples are read from the per-wave LDS scratch up front (eight 16-byte reads); inside the
// walk over the samples nothing depends on a load: a run's voxel rows come from four v_readlane of its first sample's lane.  The branches of
// the walk (does a run start here?) are uniform: the run starts are one scalar bit mask.
// Until round 4 a pair of runs was one [2 runs x 8 corners] x [16 samples] x [32 channels] product on the 16x16x4 fp32 MFMA with the
// weights masked by run: eight 32-cycle instructions per pair of runs of which, at 1.2 samples per cell, 15 of 16 k-slots multiplied zeros --
// 1 800 matrix-pipe cycles per tile that also block the SIMD's vector issue (4.1), against 128 here -- followed by a dependent chain run
// start -> voxel indices -> 64-bit addresses -> atomics, seven times per tile (measured: ~600 cycles per pass, 2 us per tile of a frozen role).
// The optimiser mask is not consulted: Adam skips masked voxels and clears their gradient (k_adam_multi).
// scratch (floats): gT[32][20] | wT[8][20]  = 800 floats
// ------------------------------------------------------------------------------------------------------
#define NSK_SCRATCH_FLOATS 944
__device__ __forceinline__ f4 mfma_outer(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ void scatter_tile(const GridD& G, const Tri& T, const f4 (&gc)[2], int lane, bool valid,
                                             float* __restrict__ scratch)
{
    const int j = lane & 15, g = lane >> 4;
    float* gT = scratch;                                    // [32 channels][20]: column = sample
    float* wT = scratch + 640;                              // [8 corners][20]
#pragma unroll
    for (int i = 0; i < 4; ++i) {                          // (a sample past the end of the batch is a finite copy of the last one: its zero WEIGHTS below drop it)
        gT[(4 * g + i) * 20 + j] = gc[0][i];
        gT[(16 + 4 * g + i) * 20 + j] = gc[1][i];
    }
    // run starts among the tile's 16 samples (lanes 0..15 carry them; every quarter of the wave holds the same Tri): the cell of the sample
    // to the left by a row shift inside the 16 lanes (the row's first lane keeps its own and is a start anyway)
    const int cell = T.vox[0];
    const int prev = __builtin_amdgcn_update_dpp(cell, cell, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
    const bool st = j == 0 || cell != prev;
    const unsigned mask16 = (unsigned)__builtin_amdgcn_readfirstlane((int)(__builtin_amdgcn_ballot_w64(st) & 0xffffull));
    if (g == 0) {
#pragma unroll
        for (int c = 0; c < 8; ++c) wT[c * 20 + j] = valid ? T.w[c] : 0.f;
    }
    lds_fence();
#ifdef NSK_SCATTER_NO_WALK      // timing experiment: the tile's operands go to LDS, nothing else happens
    return;
#endif
    const int q = lane & 3, cg = lane >> 5, ch = (lane >> 2) & 7;
    const float* arow = wT + (4 * cg + q) * 20;             // this lane's A row: the weights of corner 4 cg + q, by sample
    const float* brow = gT + (4 * ch + q) * 20;             // this lane's B column: channel 4 ch + q, by sample
    char* const gbase = reinterpret_cast<char*>(G.g);
    const unsigned coff = (unsigned)(4 * ch + q) * 4u;      // byte offset of the lane's channel inside a voxel row (128 bytes; a level has < 2^25 voxels)
    const unsigned zsel = cg ? 0xffffffffu : 0u;
#if defined(NSK_EXPERIMENT) && !defined(NSK_SCATTER_PLAIN)
    const int dbg = __builtin_amdgcn_readfirstlane(nsk_dbg_flags);      // tools/exp_ts.py: 1 no atomics (addresses kept), 2 no matrix instructions (their branches cost ~10 us at 1000 rays: -DNSK_SCATTER_PLAIN leaves them out)
#else
    constexpr int dbg = 0;
#endif
    f4 d = (f4)(0.f);
    // The voxel rows of a run: corner c = 4 dz + 2 dy + dx sits at vox[0] + dx ox + dy oy + dz oz (tri_setup: the clamped neighbour along an axis
    // is the voxel itself or one stride further), so four scalars read from the lane of the run's first sample -- vox[0], vox[1], vox[2], vox[4] --
    // give all eight: result register i of this lane belongs to corner 4 cg + i = vox[i & 3 pattern] + cg oz.  No LDS, no load latency in the walk
    // (a first version fetched the rows from LDS one sample ahead and waited ~100 cycles for them at every run start).
    int s0 = 0, s1 = 0, s2 = 0, s4 = 0;
    auto run_start = [&](int s) {                            // s is a constant after unrolling
        s0 = __builtin_amdgcn_readlane(T.vox[0], s); s1 = __builtin_amdgcn_readlane(T.vox[1], s);
        s2 = __builtin_amdgcn_readlane(T.vox[2], s); s4 = __builtin_amdgcn_readlane(T.vox[4], s);
    };
    auto flush = [&]() {
        const unsigned vb = ((unsigned)((s4 - s0) << 7) & zsel) | coff;      // + dz oz (upper half of the wave) + the lane's channel
        const int sv[4] = {s0, s1, s2, s1 + s2 - s0};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float* dst = reinterpret_cast<float*>(gbase + (((unsigned)sv[i] << 7) + vb));
#if defined(NSK_SCATTER_WG_SCOPE)      // timing experiments only: the atomics at workgroup scope (wrong sums across XCDs) / no atomics
            __hip_atomic_fetch_add(dst, d[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#elif defined(NSK_SCATTER_NO_ATOMICS)
            asm volatile("" :: "v"(dst), "v"(d[i]));
#else
            if (dbg & 1) asm volatile("" :: "v"(dst), "v"(d[i])); else atomicAdd(dst, d[i]);
#endif
        }
    };
    run_start(0);
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const f4 a4 = *reinterpret_cast<const f4*>(arow + 4 * h);
        const f4 b4 = *reinterpret_cast<const f4*>(brow + 4 * h);
        // (a fast path for groups of four samples without a run start -- four matrix instructions, no branches -- measured nothing at K3 and +1 us at K2)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int s = 4 * h + t;
            if (s > 0 && ((mask16 >> s) & 1u)) {            // uniform: a run ends in front of sample s (an `if` without `else`: one branch per sample)
                flush();
                run_start(s);
                d = (f4)(0.f);
            }
            if (dbg & 2) d[0] += a4[t] * b4[t]; else d = mfma_outer(a4[t], b4[t], d);
        }
    }
    flush();
    lds_fence();
}

// spatial derivative of the trilinear lookup contracted with g_c: returns this lane's partial (its 8 channels)
__device__ __forceinline__ void tri_grad_p(const GridD& G, const Tri& T, int g, const f4 (&gc)[2], float (&gp)[3])
{
    float gi[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        int dx = c & 1, dy = (c >> 1) & 1, dz = c >> 2;
        // out-of-grid corners carry value 0 in the reference (their weight is 0 and the axis is clipped)
        const f4* vp = reinterpret_cast<const f4*>(G.v + (size_t)T.vox[c] * 32 + 4 * g);
        f4 a = vp[0], b = vp[4];
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) dot += a[i] * gc[0][i] + b[i] * gc[1][i];
        float wx = dx ? T.t[0] : 1.f - T.t[0], wy = dy ? T.t[1] : 1.f - T.t[1], wz = dz ? T.t[2] : 1.f - T.t[2];
        gi[0] += (dx ? dot : -dot) * wy * wz;
        gi[1] += (dy ? dot : -dot) * wx * wz;
        gi[2] += (dz ? dot : -dot) * wx * wy;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) gp[k] += gi[k] * T.gmul[k];
}

#ifdef NSK_EXPERIMENT
#define NSK_DBG(A, bit) (((A).flags >> (bit)) & 1u)
#else
#define NSK_DBG(A, bit) 0u
#endif

// ------------------------------------------------------------------------------------------------------
// K4 (frozen decoders): backward over 16-sample tiles from the ReLU bits the forward saved:
// g_out -> chain of transposed products -> g_c (-> run-deduplicated scatter into the grid gradient)
// [-> g_p for NSK_GRAD_RAYS: embedding and trilinear derivatives].  LDS holds the backward image.
// Trainable decoders use decode_bwd_train_body (nsk_train.h).
// The same copy in two halves, so that a body can put the first tile's own loads between them: the image loads are issued, then the
// sample loads (which wait only for the sample index fetched before the image: loads return in order), then the image is stored.
// Until round 3 a body copied its image, met at the barrier and only then started the chain index -> sample -> corners: three round
// trips behind the image's one (tools/exp_ph3.py: "first stage" 9 000 cycles after an image copy of 2 000).
template <int K> struct ImgRegs { f4 v[K]; };
template <int NT, int K>
__device__ __forceinline__ void image_issue(ImgRegs<K>& R, const f4* __restrict__ src, int n4)
{
#pragma unroll
    for (int u = 0; u < K; ++u) { const int i = u * NT + (int)threadIdx.x; R.v[u] = i < n4 ? src[i] : (f4)(0.f); }
}
template <int NT, int K>
__device__ __forceinline__ void image_commit(f4* __restrict__ dst, const ImgRegs<K>& R, const f4* __restrict__ src, int n4)
{
#pragma unroll
    for (int u = 0; u < K; ++u) { const int i = u * NT + (int)threadIdx.x; if (i < n4) dst[i] = R.v[u]; }
    if (n4 > K * NT) copy_image_to_lds<NT>(dst + K * NT, src + K * NT, n4 - K * NT);
}

// ------------------------------------------------------------------------------------------------------
template <int WHICH, bool RAYS, int NW = 8, bool FULL = false, bool DYN = false>      // FULL: the chain on the fp32 MFMA whatever RAYS says (nsk_set_backward_mode 0)
__device__ __forceinline__ void decode_bwd_body(const DecArgs& A, int bid, int nb)          // DYN: the Tracker's median mask is found and applied here (composite mode 5)
{
    static_assert(!DYN || RAYS, "the deferred median mask belongs to the Tracker's ray-gradient launch");
    constexpr bool XYZ = WHICH != 0;
    constexpr int OD = WHICH == 3 ? 4 : 1;
    constexpr bool NEED_E = XYZ && RAYS;
    // B16: an MLP decoder's chain (nine K=32 products per tile, fifteen with the embedding's g_e = W0e^T g_a0 + W3e^T g_a3 when ray gradients
    // are wanted) runs on the fp16 matrix cores with 2-piece operands (22 significant bits, nsk_bf16.h) instead of the fp32 MFMA, which blocks
    // the SIMD's vector issue.  Until round 4 the ray-gradient launches (the Tracker, bundle adjustment) kept the fp32 chains.
    constexpr bool B16 = XYZ && !FULL;
    constexpr int IMG_F = B16 ? MlpBwdImgH::TOTAL_F : (XYZ ? MlpBwdImg::TOTAL : CoarseBwdImg::TOTAL);
    extern __shared__ __attribute__((aligned(16))) f4 smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    float* smf = reinterpret_cast<float*>(smem);
    float* scratch = smf + IMG_F + wave * 960;                  // per-wave scatter scratch (3840 B)
    float* dyn_rs = smf + IMG_F + NW * 960;                     // DYN: the residuals [NSK_MEDIAN_FUSED_MAX], then 3 * NW counters (NSK_DYN_LDS_BYTES)
    float dyn_thr = 0.f;
    const f4* img_src = B16 ? reinterpret_cast<const f4*>(A.bimg16) : A.bimg;
    constexpr int IMG_K = (IMG_F / 4 + 64 * NW - 1) / (64 * NW) < 8 ? (IMG_F / 4 + 64 * NW - 1) / (64 * NW) : 8;
    ImgRegs<IMG_K> img_regs;
    const f4* bimg = smem;
    const h8* img16 = reinterpret_cast<const h8*>(smem);
    const float* bimgf = reinterpret_cast<const float*>(smem);
    const float* Bm = nullptr;
    if constexpr (XYZ) Bm = bimgf + (B16 ? MlpBwdImgH::P_BM : MlpBwdImg::P_BM);
    const float* Wo = B16 ? bimgf + MlpBwdImgH::P_WO : (XYZ ? bimgf + MlpBwdImg::P_WO : bimgf + CoarseBwdImg::P_WO);

    const int ntasks = (A.M + 15) >> 4;
    // The next tile's loads are issued at the top of this tile and forced to have landed before this tile's scatter: vmcnt
    // retires in order, so a load issued after the atomics would wait for all of them (the frozen roles spent more time
    // there than in their MFMA chain).
    // With a cell-sorted launch the sample index itself is a load (perm): it is fetched two tiles ahead, the sample's data one.
    struct Staged { SampleRaw r; f4 gr; unsigned long long mask; int mm; } nx;
    auto slot_of = [&](int task_) { return min(min(task_, ntasks - 1) * 16 + j, A.M - 1); };
    auto stage = [&](int task_, int mm_, Staged& S_) {
        const int sl_ = slot_of(task_);
        NSK_IDX(2, sl_, A.M);
        sample_load(A, mm_, S_.r);
        S_.gr = ld32<f4>(A.g_raw, (unsigned)mm_ * 16u);
        S_.mask = ld32<unsigned long long>(A.masks, ((unsigned)sl_ * 4u + (unsigned)g) * 8u);
        S_.mm = mm_;
    };
    const int nw = nb * NW, wg = bid * NW + wave;
    const int tsh = tile_shift(ntasks, nw);
    const int kmax = tiles_per_wave(ntasks, nw, tsh);
    {
        const int mm0 = slot_sample(A, slot_of(tile_of(0, wg, nw, tsh)));
        image_issue<64 * NW>(img_regs, img_src, IMG_F / 4);
        stage(tile_of(0, wg, nw, tsh), mm0, nx);
        if constexpr (DYN) {
            constexpr int PER = (NSK_MEDIAN_FUSED_MAX + 64 * NW - 1) / (64 * NW);
            float rv[PER];
#pragma unroll
            for (int e = 0; e < PER; ++e) { const int i = (int)threadIdx.x + e * 64 * NW; rv[e] = i < A.dyn_n ? A.dyn_resid[i] : NSK_INF; }
            image_commit<64 * NW>(smem, img_regs, img_src, IMG_F / 4);
#pragma unroll
            for (int e = 0; e < PER; ++e) { const int i = (int)threadIdx.x + e * 64 * NW; if (i < NSK_MEDIAN_FUSED_MAX) dyn_rs[i] = rv[e]; }
            __syncthreads();
            dyn_thr = lower_median_x10<64 * NW, NSK_MEDIAN_FUSED_MAX>(dyn_rs, (A.dyn_n + 3) & ~3, reinterpret_cast<int*>(dyn_rs + NSK_MEDIAN_FUSED_MAX));
        } else {
            image_commit<64 * NW>(smem, img_regs, img_src, IMG_F / 4);
            __syncthreads();
        }
    }
    int mm_next = slot_sample(A, slot_of(tile_of(1, wg, nw, tsh)));
    const bool det = (A.flags & 0x8000u) != 0;      // deterministic debug mode: every wave walks all kmax rounds (they meet at barriers)
    if (!det) wave_skew(A, wave, NW);
    for (int k = 0; k < kmax; ++k) {
        const int task = tile_of(k, wg, nw, tsh);
        if (task >= ntasks && !det) break;
        asm volatile("" ::: "memory");      // keep the LDS fragment reads inside the loop (LICM would hoist + spill them)
        const bool valid = task * 16 + j < A.M;
        const int mm = nx.mm;
        float px, py, pz;
        sample_finish(A, nx.r, px, py, pz);
        const float zz = A.pts ? 0.f : nx.r.z;
        const int n = A.pts ? 0 : ray_of(A, mm);
        Tri T;
        tri_setup(A.grid, A.bound, px, py, pz, T);
        float gout[OD];
        {
            f4 gr = nx.gr;
            if (!valid) gr = (f4)(0.f);
            if constexpr (DYN) { if (!(dyn_rs[n] < dyn_thr)) gr = (f4)(0.f); }        // the ray fails |gt - depth| < 10 median: Tracker.cpp:71
            if constexpr (OD == 4) { gout[0] = gr[0]; gout[1] = gr[1]; gout[2] = gr[2]; gout[3] = 0.f; }
            else gout[0] = gr[3];
        }
        float unscale = 1.f;
        if constexpr (B16) unscale = chain_scale<OD>(gout);
        const unsigned long long mask = nx.mask;
        stage(tile_of(k + 1, wg, nw, tsh), mm_next, nx);
        mm_next = slot_sample(A, slot_of(tile_of(k + 2, wg, nw, tsh)));
        f4 xcos[6];
        if constexpr (NEED_E) { f4 e[6]; embed<true>(Bm, g, px, py, pz, e, xcos); }
        f4 gh[2];                                                // g_h4 = Wo^T g_out
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float s = 0.f;
#pragma unroll
                for (int o = 0; o < OD; ++o) s += Wo[32 * o + 16 * r + 4 * g + i] * gout[o];
                gh[r][i] = s;
            }
        f4 gc[2] = {(f4)(0.f), (f4)(0.f)}, gcL[2] = {(f4)(0.f), (f4)(0.f)};
        f4 ge[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) ge[q] = (f4)(0.f);
#pragma unroll
        for (int l = 4; l >= 0; --l) {
            H2 xg;
            if constexpr (B16) { xg = split_block_h(gh[0], gh[1]); gemm_h(img16, MlpBwdImgH::FT(l), lane, xg, gc, gcL); }
            else if constexpr (XYZ) gemm<2, 2>(bimg, MlpBwdImg::FT(l), lane, gh, gc);       // g_c += fc[l]^T g_h
            f4 ga[2];
            if constexpr (!B16) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int i = 0; i < 4; ++i) ga[r][i] = ((mask >> (8 * l + 4 * r + i)) & 1ull) ? gh[r][i] : 0.f;
            }
            if constexpr (B16) {
                if (l >= 1 || NEED_E) {
                    const H2 xa = mask_block_h(xg, (unsigned)(mask >> (8 * l)) & 0xffu);      // g_a = ReLU'(.) g_h, already in pieces
                    if constexpr (NEED_E) {                                                    // g_e (carries the sample's scale until g_p below)
                        if (l == 3) gemm_e_h(img16, MlpBwdImgH::W3ET, lane, xa, ge);
                        if (l == 0) gemm_e_h(img16, MlpBwdImgH::W0ET, lane, xa, ge);
                    }
                    if (l >= 1) {
                        f4 ghn[2] = {(f4)(0.f), (f4)(0.f)}, ghl[2] = {(f4)(0.f), (f4)(0.f)};
                        gemm_h(img16, MlpBwdImgH::WT(l > 0 ? l : 1), lane, xa, ghn, ghl);
                        gh[0] = ghn[0] + ghl[0] * (1.f / NSK_H16_SCALE); gh[1] = ghn[1] + ghl[1] * (1.f / NSK_H16_SCALE);
                    }
                }
            } else if constexpr (XYZ) {
                if constexpr (NEED_E) {
                    if (l == 3) gemm_e(bimg, MlpBwdImg::W3ET, lane, ga, ge);
                    if (l == 0) gemm_e(bimg, MlpBwdImg::W0ET, lane, ga, ge);
                }
                if (l >= 1) {
                    f4 ghn[2] = {(f4)(0.f), (f4)(0.f)};
                    gemm<2, 2>(bimg, MlpBwdImg::WT(l > 0 ? l : 1), lane, ga, ghn);
                    gh[0] = ghn[0]; gh[1] = ghn[1];
                }
            } else {
                if (l == 3) gemm<2, 2>(bimg, CoarseBwdImg::W3CT, lane, ga, gc);
                if (l == 0) gemm<2, 2>(bimg, CoarseBwdImg::W0T, lane, ga, gc);
                else {
                    const int q0 = l == 1 ? CoarseBwdImg::W1T : (l == 2 ? CoarseBwdImg::W2T : (l == 3 ? CoarseBwdImg::W3HT : CoarseBwdImg::W4T));
                    f4 ghn[2] = {(f4)(0.f), (f4)(0.f)};
                    gemm<2, 2>(bimg, q0, lane, ga, ghn);
                    gh[0] = ghn[0]; gh[1] = ghn[1];
                }
            }
        }
        if constexpr (B16) {        // join the two accumulator sets and take the sample's scale off
            gc[0] = (gc[0] + gcL[0] * (1.f / NSK_H16_SCALE)) * unscale; gc[1] = (gc[1] + gcL[1] * (1.f / NSK_H16_SCALE)) * unscale;
        }
        // opaque use: the staged registers must hold their data here, i.e. the loads retire before the first atomic below
        asm volatile("" : "+v"(nx.r.z), "+v"(nx.r.o[0]), "+v"(nx.r.o[1]), "+v"(nx.r.o[2]), "+v"(nx.r.d[0]), "+v"(nx.r.d[1]), "+v"(nx.r.d[2]),
                          "+v"(nx.gr), "+v"(nx.mask), "+v"(mm_next));
        if constexpr (RAYS) {       // g_p through the embedding (g_e * cos(pB)) B^T and through the trilinear lookup
            float gp[3] = {0.f, 0.f, 0.f};
            if constexpr (NEED_E) {
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    ge[q] *= xcos[q];
                    f4 b0 = *reinterpret_cast<const f4*>(Bm + 16 * q + 4 * g);
                    f4 b1 = *reinterpret_cast<const f4*>(Bm + 96 + 16 * q + 4 * g);
                    f4 b2 = *reinterpret_cast<const f4*>(Bm + 192 + 16 * q + 4 * g);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { gp[0] += ge[q][i] * b0[i]; gp[1] += ge[q][i] * b1[i]; gp[2] += ge[q][i] * b2[i]; }
                }
                if constexpr (B16) { gp[0] *= unscale; gp[1] *= unscale; gp[2] *= unscale; }      // the chain ran on a power-of-two multiple of the gradient
            }
            tri_grad_p(A.grid, T, g, gc, gp);
#pragma unroll
            for (int k = 0; k < 3; ++k) { gp[k] += __shfl_xor(gp[k], 16); gp[k] += __shfl_xor(gp[k], 32); }
            if (A.g_rays_o) for (int turn = 0; turn < (det ? NW : 1); ++turn) {
                if (det) { __syncthreads(); if (turn != wave) continue; }
                // the 16 samples of a tile usually belong to one ray (S = 48 = 3 tiles): sum them in the wave and add once; sixteen
                // lanes adding to one address serialise (200 rays: the ray-gradient atomics were most of a 74 us kernel)
                const int n0 = __builtin_amdgcn_readfirstlane(n);
                if (__builtin_amdgcn_ballot_w64(valid && n != n0) == 0ull) {
                    float a[6];
#pragma unroll
                    for (int k = 0; k < 3; ++k) { a[k] = valid ? gp[k] : 0.f; a[3 + k] = valid ? gp[k] * zz : 0.f; }
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1)
#pragma unroll
                        for (int k = 0; k < 6; ++k) a[k] += __shfl_xor(a[k], o);
                    if (lane == 0) {
#pragma unroll
                        for (int k = 0; k < 3; ++k) { atomicAdd(A.g_rays_o + 3 * n0 + k, a[k]); atomicAdd(A.g_rays_d + 3 * n0 + k, a[3 + k]); }
                    }
                } else if (g == 0 && valid) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        atomicAdd(A.g_rays_o + 3 * n + k, gp[k]);
                        atomicAdd(A.g_rays_d + 3 * n + k, gp[k] * zz);
                    }
                }
                if (det) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        if ((A.flags & 1u) && A.grid.g && !NSK_DBG(A, 9)) {
            if (det) {                      // deterministic debug mode (one workgroup): the waves scatter in turn, so every atomic add has a fixed place in time
                for (int w = 0; w < NW; ++w) { if (wave == w) { scatter_tile(A.grid, T, gc, lane, valid, scratch); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } __syncthreads(); }
            } else scatter_tile(A.grid, T, gc, lane, valid, scratch);
        }
    }
}

template <int WHICH, bool RAYS>
__global__ __launch_bounds__(512) void k_decode_bwd(DecArgs A) { decode_bwd_body<WHICH, RAYS>(A, blockIdx.x, gridDim.x); }

// A launch of FROZEN decoders only (the fine stage, the Tracker): their bodies need ~105 VGPRs, and inside k_decode_bwd_multi they inherit the
// trainable body's ~240 and run at two waves per SIMD.  Here a workgroup is 16 waves (four per SIMD: two more tiles' latency chains -- scatter
// fences, LDS round trips, MFMA dependencies -- to interleave) around one LDS image.
#define NSK_FROZEN_NW 16
template <bool RAYS>
__global__ __launch_bounds__(64 * NSK_FROZEN_NW) void k_decode_bwd_frozen(MultiArgs MA)
{
    if (MA.sum_n > 0 && blockIdx.x == gridDim.x - 1) { block_sum(MA.sum_src, MA.sum_n, MA.sum_dst); return; }
    if ((int)blockIdx.x >= MA.wg_end[MA.n - 1]) { sort_scan_body<NSK_FROZEN_NW / 4>(MA.scan, (int)blockIdx.x - MA.wg_end[MA.n - 1]); return; }      // see k_decode_bwd_multi
    int r = 0;
    while (r < MA.n - 1 && (int)blockIdx.x >= MA.wg_end[r]) ++r;
    const int b0 = r == 0 ? 0 : MA.wg_end[r - 1];
    const int bid = blockIdx.x - b0, nb = MA.wg_end[r] - b0;
    switch (MA.which[r]) {
    case 0: decode_bwd_body<0, RAYS, NSK_FROZEN_NW>(MA.a[r], bid, nb); break;
    case 1: decode_bwd_body<1, RAYS, NSK_FROZEN_NW>(MA.a[r], bid, nb); break;
    case 2: decode_bwd_body<2, RAYS, NSK_FROZEN_NW>(MA.a[r], bid, nb); break;
    default: decode_bwd_body<3, RAYS, NSK_FROZEN_NW>(MA.a[r], bid, nb); break;
    }
}

// The Tracker's backward (frozen decoders, ray gradients, handle_dynamic): every role's workgroups find the median threshold themselves (composite
// mode 5 left the residuals) and drop the rays that fail it; the launch's last workgroup does the same for what the compositing handed over per
// ray: the loss terms (summed in block_sum's order, so the loss has the bits of the three-launch form) and the d/d rays_d term of the
// compositing itself, which it adds to g_rays_d (zeroed by the compositing) beside the roles' own atomic adds.
#define NSK_DYN_LDS_BYTES (NSK_MEDIAN_FUSED_MAX * 4 + 3 * 8 * 4)
__global__ __launch_bounds__(512) void k_decode_bwd_track(MultiArgs MA)
{
    if (blockIdx.x == gridDim.x - 1) {
        extern __shared__ __attribute__((aligned(16))) f4 smem[];
        float* rs = reinterpret_cast<float*>(smem);
        const DecArgs& A = MA.a[0];
        for (int i = threadIdx.x; i < NSK_MEDIAN_FUSED_MAX; i += 512) rs[i] = i < A.dyn_n ? A.dyn_resid[i] : NSK_INF;
        __syncthreads();
        const float thr = lower_median_x10<512, NSK_MEDIAN_FUSED_MAX>(rs, (A.dyn_n + 3) & ~3, reinterpret_cast<int*>(rs + NSK_MEDIAN_FUSED_MAX));
        if (threadIdx.x == 0 && MA.dyn_thr_out) *MA.dyn_thr_out = thr;
        float s = 0.f;
        for (int i = threadIdx.x; i < A.dyn_n; i += 512) {
            const bool pass = rs[i] < thr;
            if (MA.sum_n > 0) s += pass ? MA.sum_src[i] : 0.f;
            if (pass && A.g_rays_d) {
#pragma unroll
                for (int k = 0; k < 3; ++k) atomicAdd(A.g_rays_d + 3 * i + k, MA.dyn_seed[3 * i + k]);
            }
        }
        if (MA.sum_n > 0) {
            __syncthreads();                                   // rs is reused for the partial sums
            s = wave_sum(s);
            if ((threadIdx.x & 63) == 0) rs[threadIdx.x >> 6] = s;
            __syncthreads();
            if (threadIdx.x == 0) { for (int w = 1; w < 8; ++w) s += rs[w]; *MA.sum_dst = s; }
        }
        return;
    }
    int r = 0;
    while (r < MA.n - 1 && (int)blockIdx.x >= MA.wg_end[r]) ++r;
    const int b0 = r == 0 ? 0 : MA.wg_end[r - 1];
    const int bid = blockIdx.x - b0, nb = MA.wg_end[r] - b0;
    switch (MA.which[r]) {
    case 0: decode_bwd_body<0, true, 8, false, true>(MA.a[r], bid, nb); break;
    case 1: decode_bwd_body<1, true, 8, false, true>(MA.a[r], bid, nb); break;
    case 2: decode_bwd_body<2, true, 8, false, true>(MA.a[r], bid, nb); break;
    default: decode_bwd_body<3, true, 8, false, true>(MA.a[r], bid, nb); break;
    }
}
