// nsk_train.h -- backward of a TRAINABLE decoder: forward recompute, gradient chain, feature-gradient scatter and
// all parameter gradients.
//
// Parameter gradients are contractions over samples, dW[o][x] = sum_s G[o][s] X[x][s].  A workgroup's 8 waves each
// own one 16-sample tile per iteration; per layer they transpose their G (upstream) and X (layer input) tiles into
// one shared LDS panel [rows][8 x 16 samples], and every 16x16 output tile of dW is then produced by ONE wave as a
// K = 128 MFMA chain over the whole panel and kept in that wave's registers across all iterations.  No LDS or
// global atomics are involved until the single flush at the end of the kernel (measured: ds_add_f32 accumulation
// of per-wave partials cost 257 us of a 420 us kernel at 1000 rays; profiles/r01b_*).
#pragma once
#include "nsk_device.h"
#include <type_traits>

// The panel holds every element as its two leading bf16 pieces, h = bf16(x) and m = bf16(x - h), in two PLANES of 16-bit values:
// plane H at the panel base, plane M PN_MOFF(CQ) bytes behind it, rows of 128 samples (256 B) + 16 B pad.  The pad makes a row step
// four banks, so the 16-row b128 operand reads and the 4-row b16 transposing writes are both conflict-free.
#define PN_RB 272                     // bytes per plane row
#define PN_GROWS 32                   // rows 0..31: G; then 16*CQ rows of X (c in the FC phases, h in the W phases); then 96 rows of e (later g_s)
#define PN_EROWS(CQ) (PN_GROWS + 16 * (CQ))
#define PN_ROWS(CQ) (PN_EROWS(CQ) + 96)
#define PN_MOFF(CQ) (PN_ROWS(CQ) * PN_RB)
#define PN_FLOATS(CQ) (2 * PN_MOFF(CQ) / 4)          // CQ = 2: 160 rows x 2 planes = 87040 B (+ 63104 B backward image <= 160 KiB); its head doubles as the per-wave scatter scratch

// one phase = one (G, X) pair: RT row tiles of G, NC 16-row chunks of X, optional row sums (bias gradients)
struct TrainPhase {
    int RT, NC, rowsum;       // tiles = RT*NC (+ RT)
    int slot0, nslots;        // this wave's accumulator slots [slot0, slot0+nslots)
    int w_base, ld, col0;     // canonical destination of dW: w_base + o*ld + col0 + 16*chunk + x
    int rows, cols_total;     // valid rows of G, valid columns over all chunks
    int b_base;               // canonical destination of the row sums (bias), -1 if none
};

template <int WHICH>
struct TrainPlan {
    static constexpr bool XYZ = WHICH != 0;
    static constexpr int CQ = WHICH == 2 ? 4 : 2;
    static constexpr int OD = WHICH == 3 ? 4 : 1;
    // phase ids
    static constexpr int P_OUT = 0;
    static constexpr int P_FC0 = 1;                 // P_FC0 + l (XYZ only)
    static constexpr int P_W0 = 6;                  // P_W0 + l
    static constexpr int P_W3H = 11;                // second X panel of layer 3 (h2 for MLP, h2 for coarse)
    static constexpr int P_DB = 12;                 // embedding matrix B (XYZ only)
    static constexpr int NPH = 13;
    TrainPhase p[NPH];
    int nslots;
    constexpr TrainPlan() : p{}, nslots(0)
    {
        constexpr DecLayoutDev L = dec_layout_dev<WHICH>();
        int s = 0;
        auto set = [&](int id, int RT, int NC, int rowsum, int w_base, int ld, int col0, int rows, int cols_total, int b_base) {
            int tiles = RT * NC + (rowsum ? RT : 0);
            int ns = (tiles + 7) / 8;
            p[id] = TrainPhase{RT, NC, rowsum, s, ns, w_base, ld, col0, rows, cols_total, b_base};
            s += ns;
        };
        set(P_OUT, 1, 2, 1, L.oWo, 32, 0, OD, 32, L.obo);
        if (XYZ) {
            for (int l = 0; l < 5; ++l) set(P_FC0 + l, 2, CQ, 1, L.oFw[l], 16 * CQ, 0, 32, 16 * CQ, L.oFb[l]);
            set(P_W0 + 0, 2, 6, 1, L.oW[0], NSK_E, 0, 32, NSK_E, L.ob[0]);
            set(P_W0 + 1, 2, 2, 1, L.oW[1], 32, 0, 32, 32, L.ob[1]);
            set(P_W0 + 2, 2, 2, 1, L.oW[2], 32, 0, 32, 32, L.ob[2]);
            set(P_W0 + 3, 2, 6, 1, L.oW[3], 125, 0, 32, NSK_E, L.ob[3]);
            set(P_W0 + 4, 2, 2, 1, L.oW[4], 32, 0, 32, 32, L.ob[4]);
            set(P_W3H, 2, 2, 0, L.oW[3], 125, NSK_E, 32, 32, -1);
            set(P_DB, 1, 6, 0, L.oB, NSK_E, 0, 3, NSK_E, -1);
        } else {
            set(P_W0 + 0, 2, 2, 1, L.oW[0], 32, 0, 32, 32, L.ob[0]);
            set(P_W0 + 1, 2, 2, 1, L.oW[1], 32, 0, 32, 32, L.ob[1]);
            set(P_W0 + 2, 2, 2, 1, L.oW[2], 32, 0, 32, 32, L.ob[2]);
            set(P_W0 + 3, 2, 2, 1, L.oW[3], 64, 0, 32, 32, L.ob[3]);       // c part
            set(P_W0 + 4, 2, 2, 1, L.oW[4], 32, 0, 32, 32, L.ob[4]);
            set(P_W3H, 2, 2, 0, L.oW[3], 64, 32, 32, 32, -1);              // h2 part
        }
        nslots = s;
    }
};

template <int WHICH>
__host__ __device__ constexpr int plan_total() { return dec_layout_dev<WHICH>().total; }

// g[i] += sum over workgroup slabs; grid (ceil(n/256), 8): each block sums 1/8 of the slabs for 256 parameters
__global__ void k_dec_grad_reduce(int n, int n4, int nslabs, const float* __restrict__ slabs, float* __restrict__ g)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int per = (nslabs + gridDim.y - 1) / gridDim.y;
    int s0 = blockIdx.y * per, s1 = min(nslabs, s0 + per);
    float acc = 0.f;
    for (int s = s0; s < s1; ++s) acc += slabs[(size_t)s * n4 + i];
    if (acc != 0.f) atomicAdd(g + i, acc);
}

// Panel operands.  A lane's eight consecutive samples of one plane are the eight bf16 operand slots of v_mfma_f32_16x16x32_bf16, so
//     mfma(A_h, B_h) + mfma(A_m, B_m) + mfma(A_h, B_m) + mfma(A_m, B_h) = sum over 32 samples of (hA + mA)(hB + mB)
// i.e. four bf16 matrix instructions (4 x 16 cycles) contract 32 samples where the fp32 form needs eight v_mfma_f32_16x16x4_f32
// (8 x 32 cycles, during which the SIMD issues no vector instruction).  Products carry 2^-17 relative rounding each, sums stay
// fp32: measured against the fp64 oracle the weight gradients are as close as the fp32 ones (tests/test_gpu_parity.py).
// (Until round 2 the two pieces shared a 32-bit word: that form needs the halves of every B word swapped for the cross products --
// four v_alignbit per operand read, 12 % of this role's vector instructions -- and four more to merge the halves when packing.)
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split2_pair(float x0, float x1, unsigned& h01, unsigned& m01)
{
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    h01 = __builtin_bit_cast(unsigned, __builtin_convertvector((f2){x0, x1}, bf2));
    const float r0 = x0 - __uint_as_float(h01 << 16), r1 = x1 - __uint_as_float(h01 & 0xffff0000u);
    m01 = __builtin_bit_cast(unsigned, __builtin_convertvector((f2){r0, r1}, bf2));
}

// transpose one D-layout quad (16 feature rows x this wave's 16 samples) into the panel at row `row0`: 16-bit stores (low / high half)
__device__ __forceinline__ void pn_put(char* __restrict__ pn, int moff, int row0, int wave, int lane, f4 x, float us = 1.f)
{
    x *= us;                                // (gradients of a chain on fp16 pieces: the sample's scale comes off here)
    const int j = lane & 15, g = lane >> 4;
    char* ph = pn + (row0 + 4 * g) * PN_RB + (16 * wave + j) * 2;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        unsigned h01, m01;
        split2_pair(x[2 * p], x[2 * p + 1], h01, m01);
        *reinterpret_cast<unsigned short*>(ph + (2 * p) * PN_RB) = (unsigned short)h01;
        *reinterpret_cast<unsigned short*>(ph + (2 * p + 1) * PN_RB) = (unsigned short)(h01 >> 16);
        *reinterpret_cast<unsigned short*>(ph + moff + (2 * p) * PN_RB) = (unsigned short)m01;
        *reinterpret_cast<unsigned short*>(ph + moff + (2 * p + 1) * PN_RB) = (unsigned short)(m01 >> 16);
    }
}

// the tiles of one phase owned by this wave, accumulated over the panel's 128 samples
template <int NSL>
__device__ __forceinline__ void pn_tiles(const char* __restrict__ pn, int moff, int RT, int NC, int rowsum, int wave, int lane, f4* acc, int xrow0 = PN_GROWS,
                                         int grow0 = 0)
{
    const int r = lane & 15, sq = lane >> 4;
    const int ntiles = RT * NC + (rowsum ? RT : 0);
#pragma unroll
    for (int k = 0; k < NSL; ++k) {
        const int tile = 8 * k + wave;
        if (tile < ntiles) {
            const bool rs = tile >= RT * NC;
            const int rt = rs ? tile - RT * NC : tile / NC;
            const int ch = rs ? 0 : tile % NC;
            const char* ga = pn + (grow0 + 16 * rt + r) * PN_RB + 16 * sq;
            const char* xb = pn + (xrow0 + 16 * ch + r) * PN_RB + 16 * sq;
            f4 d0 = acc[k], d1 = (f4)(0.f);
#pragma unroll
            for (int b = 0; b < 4; ++b) {                   // 32 samples per step
                const bf8 ah = *reinterpret_cast<const bf8*>(ga + 64 * b);
                const bf8 am = *reinterpret_cast<const bf8*>(ga + moff + 64 * b);
                if (rs) {                                   // row sums (bias gradients): every operand slot of B is 1.0
                    const bf8 one = __builtin_bit_cast(bf8, (u4v)(0x3f803f80u));
                    d0 = mfma_b(ah, one, d0);
                    d1 = mfma_b(am, one, d1);
                } else {
                    const bf8 xh = *reinterpret_cast<const bf8*>(xb + 64 * b);
                    const bf8 xm = *reinterpret_cast<const bf8*>(xb + moff + 64 * b);
                    d0 = mfma_b(ah, xh, d0);
                    d1 = mfma_b(am, xm, d1);
                    d0 = mfma_b(ah, xm, d0);
                    d1 = mfma_b(am, xh, d1);
                }
            }
            acc[k] = d0 + d1;
        }
    }
}

// store this wave's tiles of one phase into the workgroup's partial-gradient slab (canonical parameter layout).
// Every parameter is covered by exactly one (wave, slot, lane, register), so plain stores suffice; the slabs of
// all workgroups are summed by k_dec_grad_reduce (256 workgroups adding atomically into the same 15.9k addresses
// cost 58 us of a 200 us kernel).
template <int NSL>
__device__ __forceinline__ void pn_flush(float* __restrict__ g_dec, const TrainPhase P, int wave, int lane, const f4* acc)
{
    const int x = lane & 15, g = lane >> 4;
    const int ntiles = P.RT * P.NC + (P.rowsum ? P.RT : 0);
#pragma unroll
    for (int k = 0; k < NSL; ++k) {
        const int tile = 8 * k + wave;
        if (tile < ntiles) {
            const bool rs = tile >= P.RT * P.NC;
            const int rt = rs ? tile - P.RT * P.NC : tile / P.NC;
            const int ch = rs ? 0 : tile % P.NC;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int o = 16 * rt + 4 * g + i;
                const float v = acc[k][i];
                if (o < P.rows) {
                    if (rs) { if (x == 0) g_dec[P.b_base + o] = v; }
                    else if (16 * ch + x < P.cols_total) g_dec[P.w_base + o * P.ld + P.col0 + 16 * ch + x] = v;
                }
            }
        }
    }
}

#ifdef NSK_EXPERIMENT
__device__ unsigned long long nsk_dbg_ph[8][8][96];      // [workgroup < 8][wave][point]: s_memtime at points of the LAST iteration
#define NSK_PH(k) do { if (bid < 8 && lane == 0) nsk_dbg_ph[bid][wave][k] = __builtin_readcyclecounter(); } while (0)
#define NSK_PHI(k) do { if (bid < 8 && lane == 0 && it < 2) nsk_dbg_ph[bid][wave][32 + 32 * it + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define NSK_PH(k)
#define NSK_PHI(k)
#endif
// workgroup barrier for LDS data only: __syncthreads() also drains vmcnt, i.e. every outstanding global load, store and
// atomic of the wave (an atomic stays counted for thousands of cycles), although nothing in global memory is exchanged here
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// (no experiment switch removes these barriers: the scatter's run table shares LDS with the panel, and a run without them faulted the GPU)
#define NSK_BAR() lds_barrier()

template <int WHICH, bool RAYS>
__device__ __forceinline__ void decode_bwd_train_body(const DecArgs& A, int bid, int nb)
{
    constexpr bool XYZ = WHICH != 0;
    constexpr int CQ = WHICH == 2 ? 4 : 2;
    constexpr int OD = WHICH == 3 ? 4 : 1;
    // SAVED: the forward stored this decoder's block outputs h0..h4 (DecArgs::hsave) and its ReLU bits, so nothing of the MLP
    // is recomputed here and the backward (transposed) image stays in LDS for the whole kernel.  The fine decoder (64 input
    // features, images too large for LDS beside the panel) keeps the older form: forward recompute, fragments streamed from L2.
    constexpr bool SAVED = WHICH != 2;
    // H16: the chain's transposed products (g_c, g_h, g_e) on the fp16 matrix cores with 2-piece operands and a per-sample scale
    // (nsk_device.h: MlpBwdImgH, chain_scale) instead of 240 fp32 MFMAs per tile; the image has the size of the fp32 one.
    // 90 MFMAs of 16 cycles replace 240 of 32.  First built, it did not shorten the iteration (K3: 27.7 against 27.1 us per 128
    // samples) because it spilled 59 VGPRs -- loop-invariant per-lane addresses, reloaded from scratch inside the loop, each reload
    // a vmcnt wait behind the previous iteration's atomics.  Without spills (opaque lane index per iteration, g_e formed after the
    // chain): 24 us, backward 270 -> 240 us at K3.
    constexpr bool H16 = XYZ && SAVED;
    typedef MlpFwdImg<CQ> FI;
    typedef TrainPlan<WHICH> PL;
    constexpr PL plan{};
    constexpr int BWD_F = XYZ ? MlpBwdImg::TOTAL : CoarseBwdImg::TOTAL;
    constexpr int IMG_F = SAVED ? BWD_F : 0;
    extern __shared__ __attribute__((aligned(16))) f4 smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane0 = threadIdx.x & 63;
    int lane = lane0, j = lane & 15, g = lane >> 4;
    float* smf = reinterpret_cast<float*>(smem);
    char* pn = reinterpret_cast<char*>(smf + IMG_F);     // shared panel (plane H; plane M at + PM)
    constexpr int PM = PN_MOFF(CQ);
    float* scratch = smf + IMG_F + wave * 1056;     // per-wave scatter scratch (NSK_SCRATCH_FLOATS <= 1056): the head of plane H, idle between the last phase and phase OUT
    static_assert(!H16 || MlpBwdImgH::TOTAL_F == BWD_F, "the fp16 backward image takes the fp32 image's place in LDS");
    copy_image_to_lds<512>(smem, H16 ? reinterpret_cast<const f4*>(A.bimg16) : A.bimg, IMG_F / 4);
    __syncthreads();
    const h8* imgh = reinterpret_cast<const h8*>(smem);
    const f4* fimg = A.img;                         // !SAVED only: forward fragments from L2
    const float* fimgf = reinterpret_cast<const float*>(fimg);
    const f4* bimg = SAVED ? smem : A.bimg;
    const float* bimgf = reinterpret_cast<const float*>(bimg);
    const float* Bm = nullptr;                      // embedding matrix [3][96]
    if constexpr (XYZ) Bm = SAVED ? bimgf + MlpBwdImg::P_BM : fimgf + FI::P_BM;
    static_assert(MlpBwdImgH::P_BM == MlpBwdImg::P_BM && MlpBwdImgH::P_WO == MlpBwdImg::P_WO, "same fp32 tail in both images");
    const float* Bmb = Bm;
    const float* Wo = SAVED ? bimgf + (XYZ ? MlpBwdImg::P_WO : CoarseBwdImg::P_WO) : fimgf + FI::P_WO;

    f4 acc[plan.nslots];
#pragma unroll
    for (int k = 0; k < plan.nslots; ++k) acc[k] = (f4)(0.f);

    const int ntasks = (A.M + 15) >> 4;
    const int iters = tiles_per_wave(ntasks, nb * 8, 0);
    const bool scat = (A.flags & 1u) && A.grid.g && !NSK_DBG(A, 9);
    // Everything iteration it+1 reads from global memory (its samples, upstream gradient, gathered features, and the
    // forward image) is fetched at the end of iteration it BEFORE that iteration's scatter: vmcnt retires in order, so a
    // load issued after the atomics would wait for all of them (measured: 13k cycles at the top of an iteration).
    struct Staged { float px, py, pz, zz; int n; bool valid; f4 gr; f4 xc[CQ]; f4 h4[2]; unsigned long long mask; } nx;
    // single tiles dealt round-robin (tile_of with sh = 0): this role synchronises its 8 waves every iteration and carries nothing over
    auto task_of = [&](int it_) { return tile_of(it_, bid * 8 + wave, nb * 8, 0); };
    auto slot_of = [&](int it_) { return task_of(it_) * 16 + j; };
    auto stage_a = [&](int it_, int mm, Staged& S_) {  // issue the sample loads (no use, no wait); mm = the slot's sample (perm, fetched an iteration earlier)
        const int task = task_of(it_);
        const int slot = task * 16 + j;
        S_.valid = slot < A.M;
        sample_point(A, mm, S_.px, S_.py, S_.pz, S_.zz, S_.n);
        S_.gr = *reinterpret_cast<const f4*>(A.g_raw + (size_t)mm * 4);
        if constexpr (SAVED) {
            const int tk = min(task, ntasks - 1);
            S_.mask = A.masks[(size_t)min(slot, A.M - 1) * 4 + g];
            S_.h4[0] = A.hsave[((size_t)tk * 10 + 8) * 64 + lane]; S_.h4[1] = A.hsave[((size_t)tk * 10 + 9) * 64 + lane];
        }
    };
    auto stage_b = [&](Staged& S_) {                    // dependent loads: the trilinear gather
        Tri T_;
        tri_setup(A.grid, A.bound, S_.px, S_.py, S_.pz, T_);
        tri_gather(A.grid, T_, g, S_.xc[0], S_.xc[1]);
        if constexpr (WHICH == 2) {
            Tri Tm;
            tri_setup(A.grid_mid, A.bound, S_.px, S_.py, S_.pz, Tm);
            tri_gather(A.grid_mid, Tm, g, S_.xc[CQ - 2], S_.xc[CQ - 1]);
        }
    };
    int mm_next = 0;
    if (iters > 0) { stage_a(0, slot_sample(A, slot_of(0)), nx); stage_b(nx); mm_next = slot_sample(A, slot_of(1)); }
    // (the same opaque use as before the scatter below: with the staged data complete on BOTH paths into the loop header the compiler
    // places no vmcnt wait at the top of the iteration -- a wait there also waits for the previous iteration's atomics)
    if constexpr (SAVED) asm volatile("" : "+v"(nx.h4[0]), "+v"(nx.h4[1]), "+v"(nx.gr), "+v"(nx.mask), "+v"(mm_next));
    for (int it = 0; it < iters; ++it) {
        asm volatile("" ::: "memory");
        // the lane index is made opaque once per iteration: per-lane LDS addresses (panel rows of every phase, fragment rows) are then
        // rebuilt from it with immediate offsets instead of being hoisted out of the loop, where ~50 of them were spilled to scratch
        // (and every reload inside the loop is a vmcnt wait behind the previous iteration's atomics)
        lane = lane0; asm volatile("" : "+v"(lane)); j = lane & 15; g = lane >> 4;
        NSK_PH(0); NSK_PHI(0);
        if (it > 0) lds_barrier();                   // the head of the panel was the waves' scatter scratch until here
        const bool valid = nx.valid;
        float px = nx.px, py = nx.py, pz = nx.pz, zz = nx.zz; const int n = nx.n;
        Tri T;
        float gout[OD];
        {
            f4 gr = nx.gr;
            if (!valid) gr = (f4)(0.f);
            if constexpr (OD == 4) { gout[0] = gr[0]; gout[1] = gr[1]; gout[2] = gr[2]; gout[3] = 0.f; }
            else gout[0] = gr[3];
        }
        f4 go;                                           // G of phase OUT: g_out itself (rows >= OD zero)
#pragma unroll
        for (int i = 0; i < 4; ++i) go[i] = (4 * g + i) < OD ? gout[(4 * g + i) < OD ? (4 * g + i) : 0] : 0.f;
        float us = 1.f;                                  // H16: gout becomes a power-of-two multiple of itself, us takes the scale off again
        if constexpr (H16) us = chain_scale<OD>(gout);
        Act<CQ> C;
        ActC CC;
        f4 xcos[6];
        unsigned long long mask;
        const int htask = min(task_of(it), ntasks - 1);                            // this wave's tile in hsave
        if constexpr (XYZ) {
#pragma unroll
            for (int q = 0; q < CQ; ++q) C.xc[q] = nx.xc[q];
            embed<false>(Bm, g, px, py, pz, C.xe, xcos);     // cos is recomputed after the chain (24 fewer live registers)
#pragma unroll
            for (int q = 0; q < 6; ++q) pn_put(pn, PM, PN_EROWS(CQ) + 16 * q, wave, lane, C.xe[q]);     // X of phases W3 and W0; xe is dead from here
            if constexpr (SAVED) { C.h[4][0] = nx.h4[0]; C.h[4][1] = nx.h4[1]; mask = nx.mask; }
            else { mlp_forward<CQ>(fimg, lane, C); mask = C.mask; }
        } else {
            CC.xc[0] = nx.xc[0]; CC.xc[1] = nx.xc[1];
            CC.h[4][0] = nx.h4[0]; CC.h[4][1] = nx.h4[1]; mask = nx.mask;
        }
        NSK_PH(1); NSK_PHI(1);
        f4 gh[2];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float s = 0.f;
#pragma unroll
                for (int o = 0; o < OD; ++o) s += Wo[32 * o + 16 * r + 4 * g + i] * gout[o];
                gh[r][i] = s;
            }
        // ---- phase OUT: G = g_out (rows >= OD zero), X = h4 ------------------------------------------------
        {
            if (!NSK_DBG(A, 14)) pn_put(pn, PM, 0, wave, lane, go);
            const f4* h4 = XYZ ? C.h[4] : CC.h[4];
            if (!NSK_DBG(A, 14)) pn_put(pn, PM, PN_GROWS, wave, lane, h4[0]);
            if (!NSK_DBG(A, 14)) pn_put(pn, PM, PN_GROWS + 16, wave, lane, h4[1]);
            NSK_BAR();
            constexpr TrainPhase P = plan.p[PL::P_OUT];
            if (!NSK_DBG(A, 13)) pn_tiles<P.nslots>(pn, PM, P.RT, P.NC, P.rowsum, wave, lane, acc + P.slot0);
            NSK_BAR();
        }
        NSK_PH(2); NSK_PHI(2);
        f4 gc[2] = {(f4)(0.f), (f4)(0.f)};
        f4 ge[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) ge[q] = (f4)(0.f);
        H2 xa3;
        auto load_h = [&](auto KC) {
            constexpr int k = decltype(KC)::value;
            const f4* src = A.hsave + ((size_t)htask * 10 + 2 * k) * 64 + lane;
            if constexpr (XYZ) { C.h[k][0] = src[0]; C.h[k][1] = src[64]; }
            else { CC.h[k][0] = src[0]; CC.h[k][1] = src[64]; }
        };
        if constexpr (SAVED) load_h(std::integral_constant<int, 3>{});
        auto layer = [&](auto LC) {
            constexpr int l = decltype(LC)::value;
            // h[l-1] is the X operand of this layer's weight phase.  It is fetched a whole layer ahead (h[3] at the top of the iteration):
            // one phase ahead, as it was, every W phase opened with s_waitcnt vmcnt on a load issued ~1000 cycles earlier
            if constexpr (SAVED && l >= 2) load_h(std::integral_constant<int, l - 2>{});
            if constexpr (l == 2) NSK_PH(20);
            if constexpr (XYZ) {
                if constexpr (H16) {        // (the low accumulators join g_c layer by layer: eight registers fewer across the panel phases)
                    const H2 xg = split_block_h(gh[0], gh[1]);
                    f4 gl[2] = {(f4)(0.f), (f4)(0.f)};
                    gemm_h(imgh, MlpBwdImgH::FT(l), lane, xg, gc, gl);
                    gc[0] += gl[0] * (1.f / NSK_H16_SCALE); gc[1] += gl[1] * (1.f / NSK_H16_SCALE);
                } else gemm<2, 2>(bimg, MlpBwdImg::FT(l), lane, gh, gc);                 // g_c += fc[l]^T g_h
                if constexpr (l == 2) NSK_PH(21);
                // ---- phase FC_l: G = g_h, X = c ------------------------------------------------------------
                if (!NSK_DBG(A, 14)) pn_put(pn, PM, 0, wave, lane, gh[0], us);
                if (!NSK_DBG(A, 14)) pn_put(pn, PM, 16, wave, lane, gh[1], us);
#pragma unroll
                for (int q = 0; q < CQ; ++q) if (!NSK_DBG(A, 14)) pn_put(pn, PM, PN_GROWS + 16 * q, wave, lane, C.xc[q]);
                if constexpr (l == 2) NSK_PH(22);
                NSK_BAR();
                if constexpr (l == 2) NSK_PH(23);
                constexpr TrainPhase P = plan.p[PL::P_FC0 + l];
                if (!NSK_DBG(A, 13)) pn_tiles<P.nslots>(pn, PM, P.RT, P.NC, P.rowsum, wave, lane, acc + P.slot0);
                if constexpr (l == 2) NSK_PH(24);
                NSK_BAR();
                if constexpr (l == 2) NSK_PH(25);
            }
            if constexpr (l == 3) NSK_PH(13); NSK_PHI(13);
            f4 ga[2];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) ga[r][i] = ((mask >> (8 * l + 4 * r + i)) & 1ull) ? gh[r][i] : 0.f;
            H2 xa;
            if constexpr (H16) xa = split_block_h(ga[0], ga[1]);       // (split again rather than masking g_h's pieces: those would stay live across phase FC)
            // ---- phase W_l: G = g_a, X = layer input ----------------------------------------------------------
            {
                if (!NSK_DBG(A, 14)) pn_put(pn, PM, 0, wave, lane, ga[0], us);
                if (!NSK_DBG(A, 14)) pn_put(pn, PM, 16, wave, lane, ga[1], us);
                if constexpr (XYZ) {
                    if constexpr (l == 0) {
                    } else if constexpr (l == 3) {
                        // layer 3 has two inputs, e (its own panel rows) and h2: h2 goes into the X rows in the same phase -- the G panel
                        // is the same for both products (this used to be a phase of its own: two more barriers per iteration)
                        if (!NSK_DBG(A, 14)) pn_put(pn, PM, PN_GROWS, wave, lane, C.h[2][0]);
                        if (!NSK_DBG(A, 14)) pn_put(pn, PM, PN_GROWS + 16, wave, lane, C.h[2][1]);
                    } else {
                        if (!NSK_DBG(A, 14)) pn_put(pn, PM, PN_GROWS, wave, lane, C.h[l - 1][0]);
                        if (!NSK_DBG(A, 14)) pn_put(pn, PM, PN_GROWS + 16, wave, lane, C.h[l - 1][1]);
                    }
                } else {
                    if constexpr (l == 0 || l == 3) { pn_put(pn, PM, PN_GROWS, wave, lane, CC.xc[0]); pn_put(pn, PM, PN_GROWS + 16, wave, lane, CC.xc[1]); }
                    else { pn_put(pn, PM, PN_GROWS, wave, lane, CC.h[l - 1][0]); pn_put(pn, PM, PN_GROWS + 16, wave, lane, CC.h[l - 1][1]); }
                }
                if constexpr (l == 2) NSK_PH(26);
                NSK_BAR();
                if constexpr (l == 2) NSK_PH(27);
                constexpr TrainPhase P = plan.p[PL::P_W0 + l];
                constexpr int xrow0 = (XYZ && (l == 0 || l == 3)) ? PN_EROWS(CQ) : PN_GROWS;
                if (!NSK_DBG(A, 13)) pn_tiles<P.nslots>(pn, PM, P.RT, P.NC, P.rowsum, wave, lane, acc + P.slot0, xrow0);
                if constexpr (XYZ && l == 3) {        // second input of layer 3 (h2, in the X rows), same phase; its four tiles go to waves 4..7
                    constexpr TrainPhase P2 = plan.p[PL::P_W3H];      // (the e tiles above give waves 0..5 two tiles and waves 6, 7 one)
                    if (!NSK_DBG(A, 13)) pn_tiles<P2.nslots>(pn, PM, P2.RT, P2.NC, P2.rowsum, (wave + 4) & 7, lane, acc + P2.slot0);
                }
                if constexpr (l == 2) NSK_PH(28);
                NSK_BAR();
                if constexpr (l == 2) NSK_PH(29);
                if constexpr (l == 3) NSK_PH(14); NSK_PHI(14);
                if constexpr (!XYZ && l == 3) {        // coarse decoder: second input panel of layer 3: h2 (G panel unchanged)
                    const f4* h2 = CC.h[2];
                    if (!NSK_DBG(A, 14)) pn_put(pn, PM, PN_GROWS, wave, lane, h2[0]);
                    if (!NSK_DBG(A, 14)) pn_put(pn, PM, PN_GROWS + 16, wave, lane, h2[1]);
                    NSK_BAR();
                    constexpr TrainPhase P2 = plan.p[PL::P_W3H];
                    if (!NSK_DBG(A, 13)) pn_tiles<P2.nslots>(pn, PM, P2.RT, P2.NC, P2.rowsum, wave, lane, acc + P2.slot0);
                    NSK_BAR();
                }
            }
            if constexpr (l == 3) NSK_PH(15); NSK_PHI(15);
            if constexpr (H16) {
                // g_e = W3e^T g_a3 + W0e^T g_a0 is needed only after the chain: layer 3 keeps the pieces of g_a3 (8 registers) instead of
                // forming its share of g_e (24 registers) three layers early
                if constexpr (l == 3) xa3 = xa;
                if constexpr (l == 0) { gemm_e_h(imgh, MlpBwdImgH::W3ET, lane, xa3, ge); gemm_e_h(imgh, MlpBwdImgH::W0ET, lane, xa, ge); }
                if constexpr (l >= 1) {
                    f4 ghn[2] = {(f4)(0.f), (f4)(0.f)}, ghl[2] = {(f4)(0.f), (f4)(0.f)};
                    gemm_h(imgh, MlpBwdImgH::WT(l), lane, xa, ghn, ghl);
                    gh[0] = ghn[0] + ghl[0] * (1.f / NSK_H16_SCALE); gh[1] = ghn[1] + ghl[1] * (1.f / NSK_H16_SCALE);
                    if constexpr (l == 2) NSK_PH(30);
                }
            } else if constexpr (XYZ) {
                if constexpr (l == 3) gemm_e(bimg, MlpBwdImg::W3ET, lane, ga, ge);
                if constexpr (l == 3) NSK_PH(16); NSK_PHI(16);
                if constexpr (l == 0) gemm_e(bimg, MlpBwdImg::W0ET, lane, ga, ge);
                if constexpr (l >= 1) {
                    f4 ghn[2] = {(f4)(0.f), (f4)(0.f)};
                    gemm<2, 2>(bimg, MlpBwdImg::WT(l), lane, ga, ghn);
                    gh[0] = ghn[0]; gh[1] = ghn[1];
                    if constexpr (l == 2) NSK_PH(30);
                }
            } else {
                if constexpr (l == 3) gemm<2, 2>(bimg, CoarseBwdImg::W3CT, lane, ga, gc);
                if constexpr (l == 0) gemm<2, 2>(bimg, CoarseBwdImg::W0T, lane, ga, gc);
                else {
                    constexpr int q0 = l == 1 ? CoarseBwdImg::W1T : (l == 2 ? CoarseBwdImg::W2T : (l == 3 ? CoarseBwdImg::W3HT : CoarseBwdImg::W4T));
                    f4 ghn[2] = {(f4)(0.f), (f4)(0.f)};
                    gemm<2, 2>(bimg, q0, lane, ga, ghn);
                    gh[0] = ghn[0]; gh[1] = ghn[1];
                }
            }
        };
        layer(std::integral_constant<int, 4>{});
        NSK_PH(3); NSK_PHI(3);
        layer(std::integral_constant<int, 3>{});
        NSK_PH(4); NSK_PHI(4);
        layer(std::integral_constant<int, 2>{});
        NSK_PH(5); NSK_PHI(5);
        layer(std::integral_constant<int, 1>{});
        NSK_PH(6); NSK_PHI(6);
        layer(std::integral_constant<int, 0>{});
        NSK_PH(7); NSK_PHI(7);
        if (it + 1 < iters) { stage_a(it + 1, mm_next, nx); mm_next = slot_sample(A, slot_of(it + 2)); }
        NSK_PH(8); NSK_PHI(8);
        if constexpr (H16) {        // take the sample's scale off g_c (g_e keeps it: phase DB and g_p below)
            gc[0] *= us; gc[1] *= us;
        }
        float gp[3] = {0.f, 0.f, 0.f};
        asm volatile("" : "+v"(px), "+v"(py), "+v"(pz));     // opaque: forces the recomputation below instead of keeping T / cos live
        tri_setup(A.grid, A.bound, px, py, pz, T);
        if constexpr (XYZ) {
            { f4 e2[6]; embed<true>(Bmb, g, px, py, pz, e2, xcos); }
#pragma unroll
            for (int q = 0; q < 6; ++q) ge[q] *= xcos[q];
            // ---- phase DB: G = p (3 rows), X = g_s ----------------------------------------------------------
            {
                f4 pq;
#pragma unroll
                for (int i = 0; i < 4; ++i) { int row = 4 * g + i; pq[i] = !valid ? 0.f : (row == 0 ? px : (row == 1 ? py : (row == 2 ? pz : 0.f))); }
                if (!NSK_DBG(A, 14)) pn_put(pn, PM, 0, wave, lane, pq, us);          // (H16: g_s below still carries the sample's scale; it comes off on this side of the product)
#pragma unroll
                for (int q = 0; q < 6; ++q) if (!NSK_DBG(A, 14)) pn_put(pn, PM, PN_EROWS(CQ) + 16 * q, wave, lane, ge[q]);
                NSK_BAR();
                constexpr TrainPhase P = plan.p[PL::P_DB];
                if (!NSK_DBG(A, 13)) pn_tiles<P.nslots>(pn, PM, P.RT, P.NC, P.rowsum, wave, lane, acc + P.slot0, PN_EROWS(CQ));
                NSK_BAR();
            }
            if constexpr (RAYS) {
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    f4 b0 = *reinterpret_cast<const f4*>(Bmb + 16 * q + 4 * g);
                    f4 b1 = *reinterpret_cast<const f4*>(Bmb + 96 + 16 * q + 4 * g);
                    f4 b2 = *reinterpret_cast<const f4*>(Bmb + 192 + 16 * q + 4 * g);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { gp[0] += ge[q][i] * b0[i]; gp[1] += ge[q][i] * b1[i]; gp[2] += ge[q][i] * b2[i]; }
                }
                if constexpr (H16) { gp[0] *= us; gp[1] *= us; gp[2] *= us; }
            }
        }
        if constexpr (RAYS) {
            tri_grad_p(A.grid, T, g, gc, gp);
#pragma unroll
            for (int k = 0; k < 3; ++k) { gp[k] += __shfl_xor(gp[k], 16); gp[k] += __shfl_xor(gp[k], 32); }
            if (A.g_rays_o) {        // one add per tile when its 16 samples share a ray (see decode_bwd_body)
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
            }
        }
        NSK_PH(9); NSK_PHI(9);
        if (it + 1 < iters) {
            stage_b(nx);
        }
        // Every staged load must have landed before the first atomic below: vmcnt retires in order, so a wait for any of them at the top of
        // the next iteration (gr, the ReLU bits, h4 are first used there) is also a wait for this scatter's sixteen atomics (a load behind
        // 16 atomics returns after ~1 600 cycles instead of ~1 100: tools/ubench/atomlat.hip).  The use is unconditional on purpose: the
        // waitcnt pass is path-insensitive, and under the `if` above it still saw a path (loads issued, use skipped) that reaches the
        // loop header with the loads pending.
        if constexpr (SAVED) asm volatile("" : "+v"(nx.h4[0]), "+v"(nx.h4[1]), "+v"(nx.gr), "+v"(nx.mask), "+v"(mm_next));
        NSK_PH(17); NSK_PHI(17);
        if (scat) {
            if (A.flags & 0x8000u) {        // deterministic debug mode: see decode_bwd_body
                for (int w = 0; w < 8; ++w) { if (wave == w) { scatter_tile(A.grid, T, gc, lane, valid, scratch); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } __syncthreads(); }
            } else scatter_tile(A.grid, T, gc, lane, valid, scratch);
        }
        NSK_PH(10); NSK_PHI(10);
    }
    NSK_PH(11);
    // ---- single flush of this wave's output tiles -------------------------------------------------------------
    float* slab = A.g_dec + (size_t)bid * ((plan_total<WHICH>() + 3) & ~3);
    pn_flush<plan.p[PL::P_OUT].nslots>(slab, plan.p[PL::P_OUT], wave, lane, acc + plan.p[PL::P_OUT].slot0);
#define NSK_FLUSH(ID) if constexpr (plan.p[ID].nslots > 0) pn_flush<plan.p[ID].nslots>(slab, plan.p[ID], wave, lane, acc + plan.p[ID].slot0);
    NSK_FLUSH(1) NSK_FLUSH(2) NSK_FLUSH(3) NSK_FLUSH(4) NSK_FLUSH(5) NSK_FLUSH(6) NSK_FLUSH(7) NSK_FLUSH(8) NSK_FLUSH(9)
    NSK_FLUSH(10) NSK_FLUSH(12)
    // (P_W3H = 11: for the MLP decoders its tiles were dealt to waves (tile + 4) & 7 -- see layer 3)
    if constexpr (plan.p[PL::P_W3H].nslots > 0)
        pn_flush<plan.p[PL::P_W3H].nslots>(slab, plan.p[PL::P_W3H], XYZ ? (wave + 4) & 7 : wave, lane, acc + plan.p[PL::P_W3H].slot0);
#undef NSK_FLUSH
    NSK_PH(12);
}


// ------------------------------------------------------------------------------------------------------------------------------
// Trainable middle / colour decoder, ONE panel phase per layer (round 2; measured no faster than the two-phase form -- the iteration
// is issue-bound, 21.7 us either way -- and no slower).  decode_bwd_train_body above spends 26 barriers per
// iteration on 13 phases; here the two weight phases of a layer (dFc_l = g_h c^T and dW_l = g_a x^T) share one store -> barrier ->
// tiles -> barrier sequence, and the grid features c are stored once per iteration.  That needs g_h, g_a, c and the layer input in
// the panel at the same time (128 rows + 96 rows of e = 224 rows, 121 856 B); the room comes from the e-part fragments
// (W0e^T, W3e^T: 24 KB), which g_e's two products -- both after the chain now -- read from global memory (L2) instead of LDS.
//   rows   0..31  G1: g_h (phase OUT: g_out; phase DB: p)         rows  64..95  XC: c (written once per iteration)
//   rows  32..63  G2: g_a                                          rows  96..127 XH: layer input h_{l-1} (phase OUT: h4)
//   rows 128..223 E : sin(pB), later g_s                           (the per-wave scatter scratch is plane H of rows 0..124)
// LDS image: fragment groups 0..17 of MlpBwdImgH (fc^T, W^T h parts) | Wo | B.
// ------------------------------------------------------------------------------------------------------------------------------
#define PM_G1 0
#define PM_G2 32
#define PM_XC 64
#define PM_XH 96
#define PM_E 128
#define PM_ROWS 224
#define PM_IMG_FRAG_F (18 * 2 * 1024 / 4)               // floats of the LDS-resident fragment groups
#define PM_IMG_F (PM_IMG_FRAG_F + 128 + 288)            // + Wo + B
#define PM_LDS_BYTES (PM_IMG_F * 4 + PM_ROWS * PN_RB * 2)

// acc[0..5] += W?e^T x with the fragments read from global memory, one 32-row slice (4 fragment loads) at a time
__device__ __forceinline__ void gemm_e2_global(const h8* __restrict__ gimg, int lane, const H2& x3, const H2& x0, f4 (&acc)[6])
{
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const FragH a30 = load_frag_h(gimg, MlpBwdImgH::W3ET + 2 * a, lane), a31 = load_frag_h(gimg, MlpBwdImgH::W3ET + 2 * a + 1, lane);
        const FragH a00 = load_frag_h(gimg, MlpBwdImgH::W0ET + 2 * a, lane), a01 = load_frag_h(gimg, MlpBwdImgH::W0ET + 2 * a + 1, lane);
        f4 tH[2] = {acc[2 * a], acc[2 * a + 1]}, tL[2] = {(f4)(0.f), (f4)(0.f)};
        mac_block_h(a30, a31, x3, tH, tL);
        mac_block_h(a00, a01, x0, tH, tL);
        acc[2 * a] = tH[0] + tL[0] * (1.f / NSK_H16_SCALE); acc[2 * a + 1] = tH[1] + tL[1] * (1.f / NSK_H16_SCALE);
    }
}

template <int WHICH, bool RAYS>
__device__ __forceinline__ void decode_bwd_train_m_body(const DecArgs& A, int bid, int nb)
{
    static_assert(WHICH == 1 || WHICH == 3, "merged-phase body: middle and colour decoders");
    constexpr int CQ = 2;
    constexpr int OD = WHICH == 3 ? 4 : 1;
    typedef TrainPlan<WHICH> PL;
    constexpr PL plan{};
    extern __shared__ __attribute__((aligned(16))) f4 smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane0 = threadIdx.x & 63;
    int lane = lane0, j = lane & 15, g = lane >> 4;
    float* smf = reinterpret_cast<float*>(smem);
    char* pn = reinterpret_cast<char*>(smf + PM_IMG_F);
    constexpr int PM = PM_ROWS * PN_RB;                  // plane M behind plane H
    float* scratch = smf + PM_IMG_F + wave * 1056;       // per-wave scatter scratch: plane H of rows 0..124 (G1, G2, XC, XH), all rewritten only after the next iteration's first barrier
    static_assert(8 * 1056 * 4 <= PM_E * PN_RB, "scatter scratch must end before the E rows");
    {
        const f4* src = reinterpret_cast<const f4*>(A.bimg16);
        copy_image_to_lds<512>(smem, src, PM_IMG_FRAG_F / 4);
        copy_image_to_lds<512>(smem + PM_IMG_FRAG_F / 4, src + MlpBwdImgH::P_WO / 4, (128 + 288) / 4);
    }
    __syncthreads();
    const h8* imgh = reinterpret_cast<const h8*>(smem);
    const h8* gimg = reinterpret_cast<const h8*>(A.bimg16);          // e-part fragments stay in global memory
    const float* Wo = smf + PM_IMG_FRAG_F;
    const float* Bm = Wo + 128;

    f4 acc[plan.nslots];
#pragma unroll
    for (int k = 0; k < plan.nslots; ++k) acc[k] = (f4)(0.f);

    const int ntasks = (A.M + 15) >> 4;
    const int iters = tiles_per_wave(ntasks, nb * 8, 0);
    const bool scat = (A.flags & 1u) && A.grid.g;
    struct Staged { float px, py, pz, zz; int n; bool valid; f4 gr; f4 xc[CQ]; f4 h4[2]; unsigned long long mask; } nx;
    auto task_of = [&](int it_) { return tile_of(it_, bid * 8 + wave, nb * 8, 0); };
    auto slot_of = [&](int it_) { return task_of(it_) * 16 + j; };
    auto stage_a = [&](int it_, int mm, Staged& S_) {
        const int task = task_of(it_);
        const int slot = task * 16 + j;
        S_.valid = slot < A.M;
        sample_point(A, mm, S_.px, S_.py, S_.pz, S_.zz, S_.n);
        S_.gr = *reinterpret_cast<const f4*>(A.g_raw + (size_t)mm * 4);
        const int tk = min(task, ntasks - 1);
        S_.mask = A.masks[(size_t)min(slot, A.M - 1) * 4 + g];
        S_.h4[0] = A.hsave[((size_t)tk * 10 + 8) * 64 + lane]; S_.h4[1] = A.hsave[((size_t)tk * 10 + 9) * 64 + lane];
    };
    auto stage_b = [&](Staged& S_) {
        Tri T_;
        tri_setup(A.grid, A.bound, S_.px, S_.py, S_.pz, T_);
        tri_gather(A.grid, T_, g, S_.xc[0], S_.xc[1]);
    };
    int mm_next = 0;
    if (iters > 0) { stage_a(0, slot_sample(A, slot_of(0)), nx); stage_b(nx); mm_next = slot_sample(A, slot_of(1)); }
    asm volatile("" : "+v"(nx.h4[0]), "+v"(nx.h4[1]), "+v"(nx.gr), "+v"(nx.mask), "+v"(mm_next));
    for (int it = 0; it < iters; ++it) {
        asm volatile("" ::: "memory");
        lane = lane0; asm volatile("" : "+v"(lane)); j = lane & 15; g = lane >> 4;      // (see decode_bwd_train_body)
        if (it > 0) lds_barrier();                   // scratch (G1 / G2), XC and E are rewritten from here on
        const bool valid = nx.valid;
        float px = nx.px, py = nx.py, pz = nx.pz, zz = nx.zz; const int n = nx.n;
        float gout[OD];
        {
            f4 gr = nx.gr;
            if (!valid) gr = (f4)(0.f);
            if constexpr (OD == 4) { gout[0] = gr[0]; gout[1] = gr[1]; gout[2] = gr[2]; gout[3] = 0.f; }
            else gout[0] = gr[3];
        }
        f4 go;
#pragma unroll
        for (int i = 0; i < 4; ++i) go[i] = (4 * g + i) < OD ? gout[(4 * g + i) < OD ? (4 * g + i) : 0] : 0.f;
        const float us = chain_scale<OD>(gout);
        const unsigned long long mask = nx.mask;
        const int htask = min(task_of(it), ntasks - 1);
        f4 hq[5][2];                                      // block outputs h0..h4 as they are fetched
        hq[4][0] = nx.h4[0]; hq[4][1] = nx.h4[1];
        auto load_h = [&](auto KC) {
            constexpr int k = decltype(KC)::value;
            const f4* src = A.hsave + ((size_t)htask * 10 + 2 * k) * 64 + lane;
            hq[k][0] = src[0]; hq[k][1] = src[64];
        };
        load_h(std::integral_constant<int, 3>{});
        {
            f4 xe[6], dummy[6];
            embed<false>(Bm, g, px, py, pz, xe, dummy);
#pragma unroll
            for (int q = 0; q < 6; ++q) pn_put(pn, PM, PM_E + 16 * q, wave, lane, xe[q]);
        }
        pn_put(pn, PM, PM_XC, wave, lane, nx.xc[0]);
        pn_put(pn, PM, PM_XC + 16, wave, lane, nx.xc[1]);
        f4 gh[2];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float s = 0.f;
#pragma unroll
                for (int o = 0; o < OD; ++o) s += Wo[32 * o + 16 * r + 4 * g + i] * gout[o];
                gh[r][i] = s;
            }
        // ---- phase OUT: G = g_out (rows >= OD zero), X = h4 --------------------------------------------------------------------
        {
            pn_put(pn, PM, PM_G1, wave, lane, go);
            pn_put(pn, PM, PM_XH, wave, lane, hq[4][0]);
            pn_put(pn, PM, PM_XH + 16, wave, lane, hq[4][1]);
            lds_barrier();
            constexpr TrainPhase P = plan.p[PL::P_OUT];
            pn_tiles<P.nslots>(pn, PM, P.RT, P.NC, P.rowsum, wave, lane, acc + P.slot0, PM_XH, PM_G1);
            lds_barrier();
        }
        f4 gc[2] = {(f4)(0.f), (f4)(0.f)};
        H2 xa3, xa;
        auto layer = [&](auto LC) {
            constexpr int l = decltype(LC)::value;
            if constexpr (l >= 2) load_h(std::integral_constant<int, l - 2>{});      // a whole layer ahead (see decode_bwd_train_body)
            {
                const H2 xg = split_block_h(gh[0], gh[1]);
                f4 gl[2] = {(f4)(0.f), (f4)(0.f)};
                gemm_h(imgh, MlpBwdImgH::FT(l), lane, xg, gc, gl);               // g_c += fc[l]^T g_h
                gc[0] += gl[0] * (1.f / NSK_H16_SCALE); gc[1] += gl[1] * (1.f / NSK_H16_SCALE);
            }
            f4 ga[2];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) ga[r][i] = ((mask >> (8 * l + 4 * r + i)) & 1ull) ? gh[r][i] : 0.f;
            xa = split_block_h(ga[0], ga[1]);
            // ---- the layer's phase: dFc_l = g_h c^T (G1 x XC), dW_l = g_a x^T (G2 x XH or E) ------------------------------------
            pn_put(pn, PM, PM_G1, wave, lane, gh[0], us);
            pn_put(pn, PM, PM_G1 + 16, wave, lane, gh[1], us);
            pn_put(pn, PM, PM_G2, wave, lane, ga[0], us);
            pn_put(pn, PM, PM_G2 + 16, wave, lane, ga[1], us);
            if constexpr (l >= 1) {
                constexpr int k = l == 3 ? 2 : l - 1;                            // layer 3 reads e (its own rows) and h2
                pn_put(pn, PM, PM_XH, wave, lane, hq[k][0]);
                pn_put(pn, PM, PM_XH + 16, wave, lane, hq[k][1]);
            }
            lds_barrier();
            {
                // FC tiles: six per layer.  In the layers whose W phase has fourteen tiles (0, 3) they are dealt to waves 6, 7, 0..3 so that no
                // wave gets more than three (W: waves 0..5 two, 6..7 one; W3H: waves 4..7)
                constexpr TrainPhase PF = plan.p[PL::P_FC0 + l];
                constexpr int rot = (l == 0 || l == 3) ? 2 : 0;
                pn_tiles<PF.nslots>(pn, PM, PF.RT, PF.NC, PF.rowsum, (wave + rot) & 7, lane, acc + PF.slot0, PM_XC, PM_G1);
                constexpr TrainPhase PW = plan.p[PL::P_W0 + l];
                pn_tiles<PW.nslots>(pn, PM, PW.RT, PW.NC, PW.rowsum, wave, lane, acc + PW.slot0, (l == 0 || l == 3) ? PM_E : PM_XH, PM_G2);
                if constexpr (l == 3) {
                    constexpr TrainPhase P2 = plan.p[PL::P_W3H];
                    pn_tiles<P2.nslots>(pn, PM, P2.RT, P2.NC, P2.rowsum, (wave + 4) & 7, lane, acc + P2.slot0, PM_XH, PM_G2);
                }
            }
            lds_barrier();
            if constexpr (l == 3) xa3 = xa;
            if constexpr (l >= 1) {
                f4 ghn[2] = {(f4)(0.f), (f4)(0.f)}, ghl[2] = {(f4)(0.f), (f4)(0.f)};
                gemm_h(imgh, MlpBwdImgH::WT(l), lane, xa, ghn, ghl);
                gh[0] = ghn[0] + ghl[0] * (1.f / NSK_H16_SCALE); gh[1] = ghn[1] + ghl[1] * (1.f / NSK_H16_SCALE);
            }
        };
        layer(std::integral_constant<int, 4>{});
        layer(std::integral_constant<int, 3>{});
        layer(std::integral_constant<int, 2>{});
        layer(std::integral_constant<int, 1>{});
        layer(std::integral_constant<int, 0>{});
        f4 ge[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) ge[q] = (f4)(0.f);
        gemm_e2_global(gimg, lane, xa3, xa, ge);            // g_e = W3e^T g_a3 + W0e^T g_a0 (still carries the sample's scale)
        if (it + 1 < iters) { stage_a(it + 1, mm_next, nx); mm_next = slot_sample(A, slot_of(it + 2)); }
        gc[0] *= us; gc[1] *= us;
        float gp[3] = {0.f, 0.f, 0.f};
        Tri T;
        tri_setup(A.grid, A.bound, px, py, pz, T);
        {
            f4 e2[6], xcos[6];
            embed<true>(Bm, g, px, py, pz, e2, xcos);
#pragma unroll
            for (int q = 0; q < 6; ++q) ge[q] *= xcos[q];
        }
        // ---- phase DB: G = p (3 rows, times the unscale factor), X = g_s -----------------------------------------------------------
        {
            f4 pq;
#pragma unroll
            for (int i = 0; i < 4; ++i) { int row = 4 * g + i; pq[i] = !valid ? 0.f : (row == 0 ? px : (row == 1 ? py : (row == 2 ? pz : 0.f))); }
            pn_put(pn, PM, PM_G1, wave, lane, pq, us);
#pragma unroll
            for (int q = 0; q < 6; ++q) pn_put(pn, PM, PM_E + 16 * q, wave, lane, ge[q]);
            lds_barrier();
            constexpr TrainPhase P = plan.p[PL::P_DB];
            pn_tiles<P.nslots>(pn, PM, P.RT, P.NC, P.rowsum, wave, lane, acc + P.slot0, PM_E, PM_G1);
            lds_barrier();
        }
        if constexpr (RAYS) {
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                f4 b0 = *reinterpret_cast<const f4*>(Bm + 16 * q + 4 * g);
                f4 b1 = *reinterpret_cast<const f4*>(Bm + 96 + 16 * q + 4 * g);
                f4 b2 = *reinterpret_cast<const f4*>(Bm + 192 + 16 * q + 4 * g);
#pragma unroll
                for (int i = 0; i < 4; ++i) { gp[0] += ge[q][i] * b0[i]; gp[1] += ge[q][i] * b1[i]; gp[2] += ge[q][i] * b2[i]; }
            }
            gp[0] *= us; gp[1] *= us; gp[2] *= us;
            tri_grad_p(A.grid, T, g, gc, gp);
#pragma unroll
            for (int k = 0; k < 3; ++k) { gp[k] += __shfl_xor(gp[k], 16); gp[k] += __shfl_xor(gp[k], 32); }
            if (A.g_rays_o) {
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
            }
        }
        if (it + 1 < iters) stage_b(nx);
        // every staged load must have landed before the first atomic below (see decode_bwd_train_body)
        asm volatile("" : "+v"(nx.h4[0]), "+v"(nx.h4[1]), "+v"(nx.gr), "+v"(nx.mask), "+v"(mm_next));
        if (scat) {
            if (A.flags & 0x8000u) {        // deterministic debug mode: see decode_bwd_body
                for (int w = 0; w < 8; ++w) { if (wave == w) { scatter_tile(A.grid, T, gc, lane, valid, scratch); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } __syncthreads(); }
            } else scatter_tile(A.grid, T, gc, lane, valid, scratch);
        }
    }
    // ---- single flush of this wave's output tiles (the wave rotations of the tiles above) ---------------------------------------
    float* slab = A.g_dec + (size_t)bid * ((plan_total<WHICH>() + 3) & ~3);
#define NSK_FLUSHW(ID, W) if constexpr (plan.p[ID].nslots > 0) pn_flush<plan.p[ID].nslots>(slab, plan.p[ID], W, lane, acc + plan.p[ID].slot0);
    NSK_FLUSHW(PL::P_OUT, wave)
    NSK_FLUSHW(PL::P_FC0 + 0, (wave + 2) & 7) NSK_FLUSHW(PL::P_FC0 + 1, wave) NSK_FLUSHW(PL::P_FC0 + 2, wave) NSK_FLUSHW(PL::P_FC0 + 3, (wave + 2) & 7) NSK_FLUSHW(PL::P_FC0 + 4, wave)
    NSK_FLUSHW(PL::P_W0 + 0, wave) NSK_FLUSHW(PL::P_W0 + 1, wave) NSK_FLUSHW(PL::P_W0 + 2, wave) NSK_FLUSHW(PL::P_W0 + 3, wave) NSK_FLUSHW(PL::P_W0 + 4, wave)
    NSK_FLUSHW(PL::P_W3H, (wave + 4) & 7) NSK_FLUSHW(PL::P_DB, wave)
#undef NSK_FLUSHW
}

// which body a trainable decoder's backward runs: the merged-phase form for the middle and colour decoders
template <int WHICH, bool RAYS>
__device__ __forceinline__ void decode_bwd_train_any(const DecArgs& A, int bid, int nb)
{
    if constexpr (WHICH == 1 || WHICH == 3) decode_bwd_train_m_body<WHICH, RAYS>(A, bid, nb);
    else decode_bwd_train_body<WHICH, RAYS>(A, bid, nb);
}

template <int WHICH, bool RAYS>
__global__ __launch_bounds__(512) void k_decode_bwd_train(DecArgs A) { decode_bwd_train_any<WHICH, RAYS>(A, blockIdx.x, gridDim.x); }

// one launch for the backward of all decoders of a stage (roles as in k_decode_fwd_multi; train[r] selects the
// trainable-decoder body, whose workgroups synchronise only among themselves)
template <bool RAYS>
__global__ __launch_bounds__(512) void k_decode_bwd_multi(MultiArgs MA)
{
    if (MA.sum_n > 0 && blockIdx.x == gridDim.x - 1) { block_sum(MA.sum_src, MA.sum_n, MA.sum_dst); return; }
    NSK_TS_BEGIN(1);
    int r = 0;
    while (r < MA.n - 1 && (int)blockIdx.x >= MA.wg_end[r]) ++r;
    const int b0 = r == 0 ? 0 : MA.wg_end[r - 1];
    const int bid = blockIdx.x - b0, nb = MA.wg_end[r] - b0;
    const int sel = MA.which[r] * 2 + (MA.train[r] ? 1 : 0);
    switch (sel) {
    case 0: decode_bwd_body<0, RAYS>(MA.a[r], bid, nb); break;
    case 1: decode_bwd_train_any<0, RAYS>(MA.a[r], bid, nb); break;
    case 2: decode_bwd_body<1, RAYS>(MA.a[r], bid, nb); break;
    case 3: decode_bwd_train_any<1, RAYS>(MA.a[r], bid, nb); break;
    case 4: decode_bwd_body<2, RAYS>(MA.a[r], bid, nb); break;
    case 5: break;     // trainable fine decoder (64-wide c: 1.4 KB of scratch per lane): launched on its own as k_decode_bwd_train<2>, never here
    case 6: decode_bwd_body<3, RAYS>(MA.a[r], bid, nb); break;
    default: decode_bwd_train_any<3, RAYS>(MA.a[r], bid, nb); break;
    }
    NSK_TS_END(1, r);
}
