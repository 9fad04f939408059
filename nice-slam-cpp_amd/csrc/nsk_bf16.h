// nsk_bf16.h -- fp32-accurate decoder forward on the bf16 matrix cores ("3-piece split").
//
// v_mfma_f32_16x16x4_f32 runs at the fp32 VECTOR rate and shares the SIMD's fp32 pipe with every VALU instruction
// (measured: chain time and VALU time add up, profiles/README.md).  v_mfma_f32_16x16x32_bf16 is 16x faster per MAC and
// runs on the matrix pipe proper.  An fp32 value splits exactly into three bf16 pieces x = h + m + l (8 significant
// bits each, 24 together), so  w*x = hh + hm + mh + hl + lh + mm  + O(2^-24 |w x|): six bf16 MFMAs with fp32
// accumulation reproduce the fp32 product (numpy emulation: 6e-8 relative on K=125 dot products, fp32 itself 2e-7).
// Cost per K=32 block: 6 x 16 = 96 matrix-pipe cycles instead of 8 x 32 = 256 fp32-pipe cycles, plus ~45 VALU
// operations to split a block of activations.
//
// Layouts.  A K=32 block of an input lives in two accumulator-layout quads (q0: features 32b + 4g + i, q1:
// 32b + 16 + 4g + i); lane group g therefore supplies the eight k values  f(g,j) = j<4 ? 4g+j : 16+4g+(j-4)  of the
// bf16 B operand (k = 8g + j).  Weights are stored as A fragments in the same k order, three pieces per fragment:
//     image16[((fg * 3 + piece) * 64 + lane) * 8 + j] = piece(W[16 rt + (lane&15)][col0 + 32 b + f(lane>>4, j)])
// with fg = 2 * (segment block index) + rt.
#pragma once
#include "nsk_device.h"

template <int CQ, int NP = 3>                     // NP pieces per weight: 3 = bf16 split (mode 1), 2 = fp16 split (mode 2)
struct MlpFwdImgB {                               // block (K=32) indices of the segments; fragment group = 2*blk + rt
    static constexpr int CB = CQ / 2;
    static constexpr int W0E = 0;                 // 3 blocks
    static constexpr int F0 = W0E + 3;
    static constexpr int W1 = F0 + CB;
    static constexpr int F1 = W1 + 1;
    static constexpr int W2 = F1 + CB;
    static constexpr int F2 = W2 + 1;
    static constexpr int W3E = F2 + CB;           // 3 blocks
    static constexpr int W3H = W3E + 3;
    static constexpr int F3 = W3H + 1;
    static constexpr int W4 = F3 + CB;
    static constexpr int F4 = W4 + 1;
    static constexpr int NBLK = F4 + CB;          // 15 (CQ=2), 20 (CQ=4)
    static constexpr int FRAG_BYTES = NBLK * 2 * NP * 1024;
    static constexpr int P_F32 = FRAG_BYTES / 4;  // float offset of the plain fp32 tail (same order as MlpFwdImg)
    static constexpr int P_B = P_F32;
    static constexpr int P_BC = P_B + 160;
    static constexpr int P_WO = P_BC + 160;
    static constexpr int P_BO = P_WO + 128;
    static constexpr int P_BM = P_BO + 4;
    static constexpr int TOTAL_F = P_BM + 288;    // total size in floats
    __host__ __device__ static constexpr int W(int l) { return l == 1 ? W1 : (l == 2 ? W2 : (l == 4 ? W4 : -1)); }
    __host__ __device__ static constexpr int F(int l) { return l == 0 ? F0 : (l == 1 ? F1 : (l == 2 ? F2 : (l == 3 ? F3 : F4))); }
};

__host__ __device__ inline int nsk_bf16_kperm(int g, int j) { return j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4); }

// the pieces of one weight: np = 3 -> bf16 h, m, l; np = 2 -> fp16 h and the 2048-fold low piece
__device__ __forceinline__ void store_pieces(unsigned short* __restrict__ img, int t, float x, int np)
{
    const int fg = t >> 9, lane = (t >> 3) & 63, j = t & 7;
    const size_t base = ((size_t)fg * np * 64 + lane) * 8 + j;
    if (np == 2) {
        const _Float16 h = (_Float16)x;
        const _Float16 l = (_Float16)((x - (float)h) * NSK_H16_SCALE);
        img[base] = __builtin_bit_cast(unsigned short, h);
        img[base + 64 * 8] = __builtin_bit_cast(unsigned short, l);
        return;
    }
    const __bf16 h = (__bf16)x;
    const float r1 = x - (float)h;
    const __bf16 m = (__bf16)r1;
    const __bf16 l = (__bf16)(r1 - (float)m);
    img[base] = __builtin_bit_cast(unsigned short, h);
    img[base + 64 * 8] = __builtin_bit_cast(unsigned short, m);
    img[base + 128 * 8] = __builtin_bit_cast(unsigned short, l);
}

// image build: one thread per (fragment group, lane, j); idx = canonical parameter offset or -1
__global__ void k_pack_bf16(unsigned short* __restrict__ img, const int* __restrict__ idx, const float* __restrict__ P, int n, int np = 3)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int k = idx[t];
    store_pieces(img, t, k >= 0 ? P[k] : 0.f, np);
}

// static schedule (see FwdSched): step s multiplies weight block blk[s] with input block xs[s]
template <int CQ>
struct FwdSchedB {
    static constexpr int CB = CQ / 2;
    static constexpr int NSTEP = 10 + 5 * CB;
    static constexpr int XE = 0, XC = 3, XH = 3 + CB;      // input block slots: e (3), c (CB), h (1, transient)
    int blk[NSTEP], xs[NSTEP], bnd[NSTEP], lay[NSTEP];
    constexpr FwdSchedB() : blk{}, xs{}, bnd{}, lay{}
    {
        typedef MlpFwdImgB<CQ> I;
        int s = 0;
        for (int l = 0; l < 5; ++l) {
            if (l == 0 || l == 3) { const int b0 = l == 0 ? I::W0E : I::W3E; for (int b = 0; b < 3; ++b) { blk[s] = b0 + b; xs[s] = XE + b; lay[s] = l; bnd[s] = 0; ++s; } }
            if (l != 0) { blk[s] = l == 3 ? I::W3H : I::W(l); xs[s] = XH; lay[s] = l; bnd[s] = 0; ++s; }
            bnd[s - 1] = 1;
            for (int b = 0; b < CB; ++b) { blk[s] = I::F(l) + b; xs[s] = XC + b; lay[s] = l; bnd[s] = 0; ++s; }
            bnd[s - 1] = 2;
        }
    }
};

// MLP::forward (reference src/models/MLP.cpp:76-102) on the bf16 matrix cores; same results as mlp_forward to fp32 rounding
template <int CQ, bool DUMP = false>
__device__ __forceinline__ void mlp_forward_bf16(const bf8* __restrict__ img, const float* __restrict__ imgf, int lane, Act<CQ>& A, float* dump = nullptr)
{
    typedef MlpFwdImgB<CQ> I;
    constexpr FwdSchedB<CQ> S{};
    const int g = lane >> 4;
    B3 X[S.XH + 1];
#pragma unroll
    for (int b = 0; b < 3; ++b) X[S.XE + b] = split_block(A.xe[2 * b], A.xe[2 * b + 1]);
#pragma unroll
    for (int b = 0; b < S.CB; ++b) X[S.XC + b] = split_block(A.xc[2 * b], A.xc[2 * b + 1]);
    Frag3 ring[2][2];
    ring[0][0] = load_frag(img, 2 * S.blk[0], lane); ring[0][1] = load_frag(img, 2 * S.blk[0] + 1, lane);
    unsigned long long mask = 0;
    f4 acc[2];
    load_bias(imgf + I::P_B, g, acc);
#pragma unroll
    for (int s = 0; s < S.NSTEP; ++s) {
        if (s + 1 < S.NSTEP) { ring[(s + 1) & 1][0] = load_frag(img, 2 * S.blk[s + 1], lane); ring[(s + 1) & 1][1] = load_frag(img, 2 * S.blk[s + 1] + 1, lane); }
        __builtin_amdgcn_sched_barrier(0);
        mac_block(ring[s & 1][0], ring[s & 1][1], X[S.xs[s]], acc);
        if (S.bnd[s] == 1) {
            if constexpr (DUMP) dump_preact(dump, S.lay[s], g, acc);
            mask |= (unsigned long long)relu_mask(acc) << (8 * S.lay[s]);
            f4 bc[2]; load_bias(imgf + I::P_BC + 32 * S.lay[s], g, bc); acc[0] += bc[0]; acc[1] += bc[1];
        } else if (S.bnd[s] == 2) {
            const int l = S.lay[s];
            A.h[l][0] = acc[0]; A.h[l][1] = acc[1];
            if (l < 4) { X[S.XH] = split_block(acc[0], acc[1]); load_bias(imgf + I::P_B + 32 * (l + 1), g, acc); }
        }
    }
    A.mask = mask;
}

// the same chain on fp16 pieces (matmul mode 2): three MFMAs per K=32 block and row tile instead of six, two accumulator sets
// (accH: Wh x_h from the bias on; accL: the two cross products, in units of 1/2048) that meet before every ReLU and block output
template <int CQ, bool DUMP = false>
__device__ __forceinline__ void mlp_forward_f16(const h8* __restrict__ img, const float* __restrict__ imgf, int lane, Act<CQ>& A, float* dump = nullptr)
{
    typedef MlpFwdImgB<CQ, 2> I;
    constexpr FwdSchedB<CQ> S{};
    const int g = lane >> 4;
    constexpr float INV = 1.f / NSK_H16_SCALE;
    H2 X[S.XH + 1];
#pragma unroll
    for (int b = 0; b < 3; ++b) X[S.XE + b] = split_block_h(A.xe[2 * b], A.xe[2 * b + 1]);
#pragma unroll
    for (int b = 0; b < S.CB; ++b) X[S.XC + b] = split_block_h(A.xc[2 * b], A.xc[2 * b + 1]);
    FragH ring[2][2];
    ring[0][0] = load_frag_h(img, 2 * S.blk[0], lane); ring[0][1] = load_frag_h(img, 2 * S.blk[0] + 1, lane);
    unsigned long long mask = 0;
    f4 acc[2], accL[2] = {(f4)(0.f), (f4)(0.f)};
    load_bias(imgf + I::P_B, g, acc);
#pragma unroll
    for (int s = 0; s < S.NSTEP; ++s) {
        if (s + 1 < S.NSTEP) { ring[(s + 1) & 1][0] = load_frag_h(img, 2 * S.blk[s + 1], lane); ring[(s + 1) & 1][1] = load_frag_h(img, 2 * S.blk[s + 1] + 1, lane); }
        __builtin_amdgcn_sched_barrier(0);
        mac_block_h(ring[s & 1][0], ring[s & 1][1], X[S.xs[s]], acc, accL);
        if (S.bnd[s] != 0) {
            acc[0] += accL[0] * INV; acc[1] += accL[1] * INV;
            accL[0] = (f4)(0.f); accL[1] = (f4)(0.f);
        }
        if (S.bnd[s] == 1) {
            if constexpr (DUMP) dump_preact(dump, S.lay[s], g, acc);
            mask |= (unsigned long long)relu_mask(acc) << (8 * S.lay[s]);
            f4 bc[2]; load_bias(imgf + I::P_BC + 32 * S.lay[s], g, bc); acc[0] += bc[0]; acc[1] += bc[1];
        } else if (S.bnd[s] == 2) {
            const int l = S.lay[s];
            A.h[l][0] = acc[0]; A.h[l][1] = acc[1];
            if (l < 4) { X[S.XH] = split_block_h(acc[0], acc[1]); load_bias(imgf + I::P_B + 32 * (l + 1), g, acc); }
        }
    }
    A.mask = mask;
}

// K2 (bf16-split form): same contract as decode_fwd_body for the MLP decoders (WHICH = 1, 2, 3)
template <int WHICH, int NW = 8, int NP = 3, bool DUMP = false>
__device__ __forceinline__ void decode_fwd_bf16_body(const DecArgs& A, int bid, int nb)
{
    constexpr int CQ = WHICH == 2 ? 4 : 2;
    constexpr int OD = WHICH == 3 ? 4 : 1;
    typedef MlpFwdImgB<CQ, NP> I;
    extern __shared__ __attribute__((aligned(16))) f4 smem[];
    const f4* src = reinterpret_cast<const f4*>(A.img16);
    copy_image_to_lds<64 * NW>(smem, src, I::TOTAL_F / 4);
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    const bf8* img = reinterpret_cast<const bf8*>(smem);
    const float* imgf = reinterpret_cast<const float*>(smem);
    const int ntasks = (A.M + 15) >> 4;
    // the sample loads (z, ray) of the next tile are issued before this tile's MLP: one of the two dependent L2 round
    // trips (z -> p -> gather) leaves the critical path (29 % of wave time was s_waitcnt, profiles/README.md)
    // (with a cell-sorted launch the sample index is itself a load, A.perm: fetched two tiles ahead)
    const int nw = nb * NW, wg = bid * NW + wave;
    const int tsh = tile_shift(ntasks, nw);
    const int kmax = tiles_per_wave(ntasks, nw, tsh);
    SampleRaw nx;
    int m = slot_sample(A, tile_of(0, wg, nw, tsh) * 16 + j);
    sample_load(A, m, nx);
    int m_next = slot_sample(A, tile_of(1, wg, nw, tsh) * 16 + j);
    wave_skew(A, wave, NW);
    for (int k = 0; k < kmax; ++k) {
        const int task = tile_of(k, wg, nw, tsh);
        if (task >= ntasks) break;
        asm volatile("" ::: "memory");
        const int slot = task * 16 + j;
        float px, py, pz;
        sample_finish(A, nx, px, py, pz);
        Tri T;
        tri_setup(A.grid, A.bound, px, py, pz, T);
        Act<CQ> C;
        GatherRaw R;                                   // the gather's 16 loads are in flight while the embedding is computed
        tri_gather_issue<true>(A.grid, T, g, R);
        sample_load(A, m_next, nx);                    // unconditional (clamped): no branch, no wait here
        const int m_cur = m;
        m = m_next;
        m_next = slot_sample(A, tile_of(k + 2, wg, nw, tsh) * 16 + j);
        f4 dummy[6];
        embed<false>(imgf + I::P_BM, g, px, py, pz, C.xe, dummy);
        asm volatile("" ::: "memory");                 // keep the order: loads, embedding, weighting
        tri_gather_reduce(T, R, C.xc[0], C.xc[1]);
        if constexpr (WHICH == 2) {
            Tri Tm;
            tri_setup(A.grid_mid, A.bound, px, py, pz, Tm);
            tri_gather<true>(A.grid_mid, Tm, g, C.xc[CQ - 2], C.xc[CQ - 1]);
        }
        float* dump = nullptr;
        if constexpr (DUMP) dump = slot < A.M ? A.dump + (size_t)m_cur * 160 : nullptr;
        if constexpr (NP == 2) mlp_forward_f16<CQ, DUMP>(reinterpret_cast<const h8*>(smem), imgf, lane, C, dump);
        else mlp_forward_bf16<CQ, DUMP>(img, imgf, lane, C, dump);
        float out[OD];
        mlp_output<OD>(imgf + I::P_WO, imgf + I::P_BO, g, C.h[4], out);
        if (slot < A.M) {
            if (g == 0) {
                if constexpr (OD == 4) *reinterpret_cast<f4*>(A.out + (unsigned)m_cur * 4u) = (f4){out[0], out[1], out[2], out[OD - 1]};
                else A.out[(unsigned)m_cur] = out[0];
            }
            if (A.masks) A.masks[(unsigned)slot * 4u + (unsigned)g] = C.mask;
        }
        if (A.hsave) save_h(A.hsave, task, lane, C.h);
    }
}

// The two occupancy decoders of a stage (middle, fine) as ONE role over the same tiles (round 4): the fine decoder's input is fine || middle
// features (reference src/models/NICE.cpp:40-49), so as two roles the middle level was looked up twice -- tri_setup and a 16-load gather in each
// role, the fine role's second gather the one load chain of the forward that nothing overlapped -- and the sample was loaded and finished twice.
// Here: both weight images in LDS (64 + 85 KB), the middle gather flies under the middle embedding, the fine gather under the fine embedding, the
// middle features stay in registers for the fine chain.  Each decoder's own arithmetic is the two-role form's, instruction for instruction
// (tools/ab_outputs.py: same digests).  Frozen decoders only (no block-output stores); fp16 two-piece operands (matmul mode 2).
template <int NW>
__device__ __forceinline__ void decode_fwd_occ_body(const DecArgs& Am, const DecArgs& Af, int bid, int nb)
{
    typedef MlpFwdImgB<2, 2> IM;
    typedef MlpFwdImgB<4, 2> IF;
    extern __shared__ __attribute__((aligned(16))) f4 smem[];
    copy_image_to_lds<64 * NW>(smem, reinterpret_cast<const f4*>(Am.img16), IM::TOTAL_F / 4);
    copy_image_to_lds<64 * NW>(smem + IM::TOTAL_F / 4, reinterpret_cast<const f4*>(Af.img16), IF::TOTAL_F / 4);
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    const float* imgf_m = reinterpret_cast<const float*>(smem);
    const float* imgf_f = imgf_m + IM::TOTAL_F;
    const h8* img_m = reinterpret_cast<const h8*>(imgf_m);
    const h8* img_f = reinterpret_cast<const h8*>(imgf_f);
    const int ntasks = (Af.M + 15) >> 4;
    const int nw = nb * NW, wg = bid * NW + wave;
    const int tsh = tile_shift(ntasks, nw);
    const int kmax = tiles_per_wave(ntasks, nw, tsh);
    SampleRaw nx;
    int m = slot_sample(Af, tile_of(0, wg, nw, tsh) * 16 + j);
    sample_load(Af, m, nx);
    int m_next = slot_sample(Af, tile_of(1, wg, nw, tsh) * 16 + j);
    wave_skew(Af, wave, NW);
    for (int k = 0; k < kmax; ++k) {
        const int task = tile_of(k, wg, nw, tsh);
        if (task >= ntasks) break;
        asm volatile("" ::: "memory");
        const int slot = task * 16 + j;
        float px, py, pz;
        sample_finish(Af, nx, px, py, pz);
        f4 dummy[6];
        f4 xm0, xm1;                                       // the middle level's features of the sample: the middle decoder's input and half of the fine one's
        unsigned long long mask_m;
        {
            Tri T;
            tri_setup(Am.grid, Am.bound, px, py, pz, T);
            Act<2> C;
            GatherRaw R;
            tri_gather_issue<true>(Am.grid, T, g, R);
            sample_load(Af, m_next, nx);                   // unconditional (clamped): no branch, no wait here
            embed<false>(imgf_m + IM::P_BM, g, px, py, pz, C.xe, dummy);
            asm volatile("" ::: "memory");                 // keep the order: loads, embedding, weighting
            tri_gather_reduce(T, R, xm0, xm1);
            C.xc[0] = xm0; C.xc[1] = xm1;
            mlp_forward_f16<2>(img_m, imgf_m, lane, C);
            float out[1];
            mlp_output<1>(imgf_m + IM::P_WO, imgf_m + IM::P_BO, g, C.h[4], out);
            if (slot < Am.M && g == 0) Am.out[(unsigned)m] = out[0];
            mask_m = C.mask;
        }
        const int m_cur = m;
        m = m_next;
        m_next = slot_sample(Af, tile_of(k + 2, wg, nw, tsh) * 16 + j);
        {
            Tri T;
            tri_setup(Af.grid, Af.bound, px, py, pz, T);
            Act<4> C;
            GatherRaw R;
            tri_gather_issue<true>(Af.grid, T, g, R);
            embed<false>(imgf_f + IF::P_BM, g, px, py, pz, C.xe, dummy);
            asm volatile("" ::: "memory");
            tri_gather_reduce(T, R, C.xc[0], C.xc[1]);
            C.xc[2] = xm0; C.xc[3] = xm1;
            mlp_forward_f16<4>(img_f, imgf_f, lane, C);
            float out[1];
            mlp_output<1>(imgf_f + IF::P_WO, imgf_f + IF::P_BO, g, C.h[4], out);
            if (slot < Af.M) {
                if (g == 0) Af.out[(unsigned)m_cur] = out[0];
                if (Af.masks) Af.masks[(unsigned)slot * 4u + (unsigned)g] = C.mask;
                if (Am.masks) Am.masks[(unsigned)slot * 4u + (unsigned)g] = mask_m;
            }
        }
    }
}

// roles: [0, wg_end[0]) the merged occupancy role on (a[0] = middle, a[1] = fine); [wg_end[0], wg_end[1]) the colour decoder on a[2] (MA.n == 2) or nothing
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_decode_fwd_multi_occ(MultiArgs MA)
{
    if ((int)blockIdx.x < MA.wg_end[0]) decode_fwd_occ_body<NW>(MA.a[0], MA.a[1], blockIdx.x, MA.wg_end[0]);
    else decode_fwd_bf16_body<3, NW, 2>(MA.a[2], (int)blockIdx.x - MA.wg_end[0], MA.wg_end[1] - MA.wg_end[0]);
}

template <int NW, int NP = 3>
__global__ __launch_bounds__(64 * NW) void k_decode_fwd_multi_bf16(MultiArgs MA)
{
    int r = 0;
    while (r < MA.n - 1 && (int)blockIdx.x >= MA.wg_end[r]) ++r;
    const int b0 = r == 0 ? 0 : MA.wg_end[r - 1];
    const int bid = blockIdx.x - b0, nb = MA.wg_end[r] - b0;
    switch (MA.which[r]) {
    case 0: decode_fwd_body<0, NW>(MA.a[r], bid, nb); break;
    case 1: decode_fwd_bf16_body<1, NW, NP>(MA.a[r], bid, nb); break;
    case 2: decode_fwd_bf16_body<2, NW, NP>(MA.a[r], bid, nb); break;
    default: decode_fwd_bf16_body<3, NW, NP>(MA.a[r], bid, nb); break;
    }
}

// test aid (nsk_debug_preact): one MLP decoder's forward over the samples of the last step, the same body as the step's launch with the
// ReLU inputs written out.  MODE = matmul mode (0 fp32 MFMA, 1 three bf16 pieces, 2 two fp16 pieces)
template <int MODE>
__global__ __launch_bounds__(512) void k_decode_fwd_dump(DecArgs A, int which)
{
    if constexpr (MODE == 0) {
        switch (which) {
        case 1: decode_fwd_body<1, 8, true>(A, blockIdx.x, gridDim.x); break;
        case 2: decode_fwd_body<2, 8, true>(A, blockIdx.x, gridDim.x); break;
        default: decode_fwd_body<3, 8, true>(A, blockIdx.x, gridDim.x); break;
        }
    } else {
        constexpr int NP = MODE == 2 ? 2 : 3;
        switch (which) {
        case 1: decode_fwd_bf16_body<1, 8, NP, true>(A, blockIdx.x, gridDim.x); break;
        case 2: decode_fwd_bf16_body<2, 8, NP, true>(A, blockIdx.x, gridDim.x); break;
        default: decode_fwd_bf16_body<3, 8, NP, true>(A, blockIdx.x, gridDim.x); break;
        }
    }
}
