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
#include <utility>

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

// transpose one D-layout quad (16 feature rows x this wave's 16 samples) into the panel at row `row0`.
// Until round 3 every lane stored its four rows as 16-bit values, eight ds_write_b16 per quad: an LDS store takes its four cycles whatever its
// width, 368 of them per wave and iteration kept the put phases on the LDS store path (11.8 k cycles CU-wide, tools/exp_ph3.py).  Now the two
// lanes of adjacent samples share the work: the even lane keeps rows 0, 1 of both samples, the odd lane rows 2, 3 (one DPP exchange of the dword
// the partner needs, two byte permutes), and each stores two 32-bit words per plane -- half the store instructions for +6 vector instructions.
__device__ __forceinline__ void pair_words(unsigned v01, unsigned v23, bool odd, unsigned selA, unsigned selB, unsigned& wA, unsigned& wB)
{
    const unsigned keep = odd ? v23 : v01, send = odd ? v01 : v23;
    const unsigned recv = (unsigned)__builtin_amdgcn_mov_dpp((int)send, 0xB1, 0xF, 0xF, true);      // quad_perm [1, 0, 3, 2]: the neighbouring sample's lane
    wA = __builtin_amdgcn_perm(keep, recv, selA);      // first kept row:  (even sample | odd sample << 16)
    wB = __builtin_amdgcn_perm(keep, recv, selB);      // second kept row
}
struct PairSel { bool odd; unsigned selA, selB; int rowoff, col2; };
__device__ __forceinline__ PairSel pair_sel(int wave, int lane)
{
    PairSel P;
    P.odd = (lane & 1) != 0;
    P.selA = P.odd ? 0x05040100u : 0x01000504u; P.selB = P.odd ? 0x07060302u : 0x03020706u;
    P.rowoff = 4 * (lane >> 4) + (P.odd ? 2 : 0);
    P.col2 = (16 * wave + (lane & 14)) * 2;
    return P;
}
__device__ __forceinline__ void pn_put(char* __restrict__ pn, int moff, int row0, int wave, int lane, f4 x, float us = 1.f)
{
    x *= us;                                // (gradients of a chain on fp16 pieces: the sample's scale comes off here)
    const PairSel P = pair_sel(wave, lane);
    char* ph = pn + (row0 + P.rowoff) * PN_RB + P.col2;
    unsigned h01, m01, h23, m23, a, b;
    split2_pair(x[0], x[1], h01, m01);
    split2_pair(x[2], x[3], h23, m23);
    pair_words(h01, h23, P.odd, P.selA, P.selB, a, b);
    *reinterpret_cast<unsigned*>(ph) = a; *reinterpret_cast<unsigned*>(ph + PN_RB) = b;
    pair_words(m01, m23, P.odd, P.selA, P.selB, a, b);
    *reinterpret_cast<unsigned*>(ph + moff) = a; *reinterpret_cast<unsigned*>(ph + moff + PN_RB) = b;
}

// Half-word masks of a layer's eight ReLU bits: dword d covers elements 2d (low half) and 2d + 1 (high half) of the lane's block -- the
// packing both the fp16 chain pieces (H2) and the bf16 panel pieces (pn_put) use.  v_bfe_i32 spreads a bit over a dword, v_bfi_b32 joins
// two of them: three instructions per dword.
struct Mask4 { unsigned d[4]; };
__device__ __forceinline__ Mask4 relu_mask_dwords(unsigned long long mask, int l)
{
    const unsigned w = l < 4 ? (unsigned)mask : (unsigned)(mask >> 32);
    const int sh = l < 4 ? 8 * l : 0;
    Mask4 M;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const unsigned lo = (unsigned)__builtin_amdgcn_sbfe((int)w, sh + 2 * d, 1), hi = (unsigned)__builtin_amdgcn_sbfe((int)w, sh + 2 * d + 1, 1);
        M.d[d] = (lo & 0xffffu) | (hi & 0xffff0000u);
    }
    return M;
}
__device__ __forceinline__ H2 mask_block_h4(const H2& x, const Mask4& M)
{
    u4v h = __builtin_bit_cast(u4v, x.h), l = __builtin_bit_cast(u4v, x.l);
#pragma unroll
    for (int d = 0; d < 4; ++d) { h[d] &= M.d[d]; l[d] &= M.d[d]; }
    H2 r;
    r.h = __builtin_bit_cast(h8, h); r.l = __builtin_bit_cast(h8, l);
    return r;
}
// pn_put of a gradient quad x (rows row1..) TOGETHER with its ReLU-masked copy (rows row2..): g_a = ReLU'(.) g_h, and the pieces of a zero
// are zeros, so the masked quad's pieces are the quad's pieces ANDed with the half-word masks -- one scale and one split for both.
__device__ __forceinline__ void pn_put_masked(char* __restrict__ pn, int moff, int row1, int row2, int wave, int lane, f4 x, float us, unsigned k01, unsigned k23)
{
    x *= us;
    const PairSel P = pair_sel(wave, lane);
    char* p1 = pn + (row1 + P.rowoff) * PN_RB + P.col2;
    char* p2 = p1 + (row2 - row1) * PN_RB;
    unsigned h01, m01, h23, m23, a, b;
    split2_pair(x[0], x[1], h01, m01);
    split2_pair(x[2], x[3], h23, m23);
    pair_words(h01, h23, P.odd, P.selA, P.selB, a, b);
    *reinterpret_cast<unsigned*>(p1) = a; *reinterpret_cast<unsigned*>(p1 + PN_RB) = b;
    pair_words(m01, m23, P.odd, P.selA, P.selB, a, b);
    *reinterpret_cast<unsigned*>(p1 + moff) = a; *reinterpret_cast<unsigned*>(p1 + moff + PN_RB) = b;
    pair_words(h01 & k01, h23 & k23, P.odd, P.selA, P.selB, a, b);
    *reinterpret_cast<unsigned*>(p2) = a; *reinterpret_cast<unsigned*>(p2 + PN_RB) = b;
    pair_words(m01 & k01, m23 & k23, P.odd, P.selA, P.selB, a, b);
    *reinterpret_cast<unsigned*>(p2 + moff) = a; *reinterpret_cast<unsigned*>(p2 + moff + PN_RB) = b;
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

template <int WHICH, bool RAYS, bool FULL = false>      // FULL: the chain's products on the fp32 MFMA (nsk_set_backward_mode 0); the weight-gradient panels keep two bf16 pieces
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
    constexpr bool H16 = XYZ && SAVED && !FULL;
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
// the same from an LDS copy of fragment groups W0ET..W3ET+5 (group index relative to W0ET)
__device__ __forceinline__ void gemm_e2_lds(const h8* __restrict__ eimg, int lane, const H2& x3, const H2& x0, f4 (&acc)[6])
{
    constexpr int G3 = MlpBwdImgH::W3ET - MlpBwdImgH::W0ET;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const FragH a30 = load_frag_h(eimg, G3 + 2 * a, lane), a31 = load_frag_h(eimg, G3 + 2 * a + 1, lane);
        const FragH a00 = load_frag_h(eimg, 2 * a, lane), a01 = load_frag_h(eimg, 2 * a + 1, lane);
        f4 tH[2] = {acc[2 * a], acc[2 * a + 1]}, tL[2] = {(f4)(0.f), (f4)(0.f)};
        mac_block_h(a30, a31, x3, tH, tL);
        mac_block_h(a00, a01, x0, tH, tL);
        acc[2 * a] = tH[0] + tL[0] * (1.f / NSK_H16_SCALE); acc[2 * a + 1] = tH[1] + tL[1] * (1.f / NSK_H16_SCALE);
    }
}
// The same product transposed: the chain's pieces as the A operand (rows = this wave's 16 samples), the fragment groups as B (columns = 16 embedding
// features) -- the register contents of both are what gemm_e2_lds feeds the other way round.  acc[q][i] = g_e[feature 16q + (lane & 15)][sample
// 4g + i]: the D layout then holds, per lane, one feature of four samples, which is the A operand of a 16 x 16 x 16 product over samples.
__device__ __forceinline__ void gemm_e2T_lds(const h8* __restrict__ eimg, int lane, const H2& x3, const H2& x0, f4 (&acc)[6])
{
    constexpr int G3 = MlpBwdImgH::W3ET - MlpBwdImgH::W0ET;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const FragH f3 = load_frag_h(eimg, G3 + q, lane), f0 = load_frag_h(eimg, q, lane);
        f4 tH = (f4)(0.f), tL = (f4)(0.f);
        tH = mfma_h(x3.h, f3.h, tH); tL = mfma_h(x3.h, f3.l, tL); tL = mfma_h(x3.l, f3.h, tL);
        tH = mfma_h(x0.h, f0.h, tH); tL = mfma_h(x0.h, f0.l, tL); tL = mfma_h(x0.l, f0.h, tL);
        acc[q] = tH + tL * (1.f / NSK_H16_SCALE);
    }
}
typedef short s4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4 mfma_bf16_k16(unsigned a01, unsigned a23, unsigned b01, unsigned b23, f4 c)
{
    typedef unsigned int u2v __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s4v, (u2v){a01, a23}), __builtin_bit_cast(s4v, (u2v){b01, b23}), c, 0, 0, 0);
}
// (the loads of slice a + 1 are issued before the products of slice a: written slice by slice, every slice opened with a wait for an L2
// round trip -- three of them, ~1 000 cycles each, per tile)
struct FragE { FragH a30, a31, a00, a01; };
__device__ __forceinline__ FragE load_frag_e(const h8* __restrict__ gimg, int a, int lane)
{
    FragE F;
    F.a30 = load_frag_h(gimg, MlpBwdImgH::W3ET + 2 * a, lane); F.a31 = load_frag_h(gimg, MlpBwdImgH::W3ET + 2 * a + 1, lane);
    F.a00 = load_frag_h(gimg, MlpBwdImgH::W0ET + 2 * a, lane); F.a01 = load_frag_h(gimg, MlpBwdImgH::W0ET + 2 * a + 1, lane);
    return F;
}
__device__ __forceinline__ void gemm_e2_global(const h8* __restrict__ gimg, int lane, const H2& x3, const H2& x0, f4 (&acc)[6])
{
    FragE F[3];
    F[0] = load_frag_e(gimg, 0, lane);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#ifdef NSK_V_NOGE
        if (a > 0) F[a] = load_frag_e(gimg, a, lane);
#else
        if (a + 1 < 3) F[a + 1] = load_frag_e(gimg, a + 1, lane);
#endif
        f4 tH[2] = {acc[2 * a], acc[2 * a + 1]}, tL[2] = {(f4)(0.f), (f4)(0.f)};
        mac_block_h(F[a].a30, F[a].a31, x3, tH, tL);
        mac_block_h(F[a].a00, F[a].a01, x0, tH, tL);
        acc[2 * a] = tH[0] + tL[0] * (1.f / NSK_H16_SCALE); acc[2 * a + 1] = tH[1] + tL[1] * (1.f / NSK_H16_SCALE);
    }
}

// ---- weight-gradient tiles of the merged-phase body as JOBS (round 3) ------------------------------------------------------------------------
// A job = one wave, one 16-row block of G (the A operand) and up to four 16x16 output tiles that share it: X blocks (the B operands) or the
// row sums (bias gradients).  The A fragments are read once per job and K-step instead of once per tile (a layer's twelve tiles read 160
// fragments, its four to eight jobs 96 to 128: the tile phases are bound by LDS reads, not by the matrix pipe -- 1 000 cycles per tile and wave
// against 256 of MFMA, tools/exp_ph3.py), and the jobs of every phase are dealt so that the two waves of each SIMD (w, w + 4) carry the same
// number of tiles and no wave owns more than twelve tiles in all: 48 accumulator registers instead of 60.
struct TJob { int wave, grow, nt; int xrow[4]; int ph[4]; int rt; int ch[4]; };      // ch < 0: row sums (xrow unused)
template <int WHICH>
struct JobPlan {
    typedef TrainPlan<WHICH> PL;
    static constexpr int NJ = 32;
    static constexpr int PH_OUT = 0, PH_DB = 6;          // panel phases: 0 OUT, 1 + (4 - l) layer l, 6 DB
    TJob j[NJ];
    int slot0[NJ];
    int first[8];                                        // first job of each panel phase (first[7] = NJ)
    int nslots;
    constexpr JobPlan() : j{}, slot0{}, first{}, nslots(0)
    {
        int n = 0;
        auto add = [&](int wave, int grow, int rt, int nt, int x0, int p0, int c0, int x1, int p1, int c1, int x2 = 0, int p2 = 0, int c2 = 0, int x3 = 0, int p3 = 0, int c3 = 0) {
            j[n] = TJob{wave, grow, nt, {x0, x1, x2, x3}, {p0, p1, p2, p3}, rt, {c0, c1, c2, c3}};
            ++n;
        };
        const int RS = -1;
        // phase OUT: G1 = g_out, X = h4
        first[0] = n;
        add(0, PM_G1, 0, 3, PM_XH, PL::P_OUT, 0, PM_XH + 16, PL::P_OUT, 1, 0, PL::P_OUT, RS);
        for (int l = 4; l >= 0; --l) {
            first[1 + (4 - l)] = n;
            const int FC = PL::P_FC0 + l, W = PL::P_W0 + l;
            if (l == 4 || l == 2 || l == 1) {            // four jobs of three tiles, one per SIMD; the waves alternate to even out the totals
                const int w0 = l == 4 ? 0 : 4;
                add(w0 + 0, PM_G1, 0, 3, PM_XC, FC, 0, PM_XC + 16, FC, 1, 0, FC, RS);
                add(w0 + 1, PM_G1 + 16, 1, 3, PM_XC, FC, 0, PM_XC + 16, FC, 1, 0, FC, RS);
                add(w0 + 2, PM_G2, 0, 3, PM_XH, W, 0, PM_XH + 16, W, 1, 0, W, RS);
                add(w0 + 3, PM_G2 + 16, 1, 3, PM_XH, W, 0, PM_XH + 16, W, 1, 0, W, RS);
            } else if (l == 3) {                         // 24 tiles: six per SIMD
                add(0, PM_G2, 0, 3, PM_E, W, 0, PM_E + 16, W, 1, 0, W, RS);
                add(4, PM_G1, 0, 3, PM_XC, FC, 0, PM_XC + 16, FC, 1, 0, FC, RS);
                add(1, PM_G2 + 16, 1, 3, PM_E, W, 0, PM_E + 16, W, 1, 0, W, RS);
                add(5, PM_G1 + 16, 1, 3, PM_XC, FC, 0, PM_XC + 16, FC, 1, 0, FC, RS);
                add(2, PM_G2, 0, 4, PM_E + 32, W, 2, PM_E + 48, W, 3, PM_XH, PL::P_W3H, 0, PM_XH + 16, PL::P_W3H, 1);
                add(6, PM_G2, 0, 2, PM_E + 64, W, 4, PM_E + 80, W, 5);
                add(3, PM_G2 + 16, 1, 4, PM_E + 32, W, 2, PM_E + 48, W, 3, PM_XH, PL::P_W3H, 0, PM_XH + 16, PL::P_W3H, 1);
                add(7, PM_G2 + 16, 1, 2, PM_E + 64, W, 4, PM_E + 80, W, 5);
            } else {                                     // l == 0: 20 tiles, five per SIMD
                add(0, PM_G2, 0, 3, PM_E, W, 0, PM_E + 16, W, 1, 0, W, RS);
                add(4, PM_G2, 0, 2, PM_E + 32, W, 2, PM_E + 48, W, 3);
                add(1, PM_G2 + 16, 1, 3, PM_E, W, 0, PM_E + 16, W, 1, 0, W, RS);
                add(5, PM_G2 + 16, 1, 2, PM_E + 32, W, 2, PM_E + 48, W, 3);
                add(2, PM_G1, 0, 3, PM_XC, FC, 0, PM_XC + 16, FC, 1, 0, FC, RS);
                add(6, PM_G2, 0, 2, PM_E + 64, W, 4, PM_E + 80, W, 5);
                add(3, PM_G1 + 16, 1, 3, PM_XC, FC, 0, PM_XC + 16, FC, 1, 0, FC, RS);
                add(7, PM_G2 + 16, 1, 2, PM_E + 64, W, 4, PM_E + 80, W, 5);
            }
        }
        first[6] = n;                                    // phase DB: G1 = p (3 rows), X = g_s
        add(6, PM_G1, 0, 2, PM_E, PL::P_DB, 0, PM_E + 16, PL::P_DB, 1);
        add(7, PM_G1, 0, 2, PM_E + 32, PL::P_DB, 2, PM_E + 48, PL::P_DB, 3);
        add(1, PM_G1, 0, 2, PM_E + 64, PL::P_DB, 4, PM_E + 80, PL::P_DB, 5);
        first[7] = n;
        int used[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int q = 0; q < n; ++q) { slot0[q] = used[j[q].wave]; used[j[q].wave] += j[q].nt; }
        for (int w = 0; w < 8; ++w) nslots = used[w] > nslots ? used[w] : nslots;
    }
};

// every output tile of the decoder's TrainPlan -- (phase, row tile, column chunk or row sums) -- must be owned by exactly one job
template <int WHICH>
constexpr bool jobplan_covers_plan()
{
    constexpr JobPlan<WHICH> JP{};
    constexpr TrainPlan<WHICH> plan{};
    for (int ph = 0; ph < TrainPlan<WHICH>::NPH; ++ph) {
        const TrainPhase P = plan.p[ph];
        for (int rt = 0; rt < P.RT; ++rt)
            for (int ch = -1; ch < P.NC; ++ch) {
                if (ch < 0 && !P.rowsum) continue;
                int owners = 0;
                for (int q = 0; q < JobPlan<WHICH>::NJ; ++q)
                    for (int k = 0; k < JP.j[q].nt; ++k)
                        if (JP.j[q].ph[k] == ph && JP.j[q].rt == rt && (ch < 0 ? JP.j[q].ch[k] < 0 : JP.j[q].ch[k] == ch)) ++owners;
                if (owners != 1) return false;
            }
    }
    int tiles = 0, want = 0;
    for (int q = 0; q < JobPlan<WHICH>::NJ; ++q) tiles += JP.j[q].nt;
    for (int ph = 0; ph < TrainPlan<WHICH>::NPH; ++ph) want += plan.p[ph].RT * plan.p[ph].NC + (plan.p[ph].rowsum ? plan.p[ph].RT : 0);
    return tiles == want && JP.nslots <= 12;
}
static_assert(jobplan_covers_plan<1>() && jobplan_covers_plan<3>(), "JobPlan: a weight-gradient tile is unowned, owned twice, or a wave holds more than 12 tiles");

template <int WHICH, int JJ>
__device__ __forceinline__ void pn_job(const char* __restrict__ pn, int moff, int wave, int lane, f4* acc)
{
    constexpr JobPlan<WHICH> JP{};
    constexpr TJob J = JP.j[JJ];
    constexpr int s0 = JP.slot0[JJ];
    if (wave != J.wave) return;
    const int r = lane & 15, sq = lane >> 4;
    const char* ga = pn + (J.grow + r) * PN_RB + 16 * sq;
    f4 d0[J.nt], d1[J.nt];
#pragma unroll
    for (int k = 0; k < J.nt; ++k) { d0[k] = acc[s0 + k]; d1[k] = (f4)(0.f); }
#pragma unroll
    for (int b = 0; b < 4; ++b) {                       // 32 samples per step
        const bf8 ah = *reinterpret_cast<const bf8*>(ga + 64 * b);
        const bf8 am = *reinterpret_cast<const bf8*>(ga + moff + 64 * b);
#pragma unroll
        for (int k = 0; k < J.nt; ++k) {
            if (J.ch[k] < 0) {                          // row sums: every operand slot of B is 1.0
                const bf8 one = __builtin_bit_cast(bf8, (u4v)(0x3f803f80u));
                d0[k] = mfma_b(ah, one, d0[k]);
                d1[k] = mfma_b(am, one, d1[k]);
            } else {
                const char* xb = pn + (J.xrow[k] + r) * PN_RB + 16 * sq;
                const bf8 xh = *reinterpret_cast<const bf8*>(xb + 64 * b);
                const bf8 xm = *reinterpret_cast<const bf8*>(xb + moff + 64 * b);
                d0[k] = mfma_b(ah, xh, d0[k]);
                d1[k] = mfma_b(am, xm, d1[k]);
                d0[k] = mfma_b(ah, xm, d0[k]);
                d1[k] = mfma_b(am, xh, d1[k]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < J.nt; ++k) acc[s0 + k] = d0[k] + d1[k];
}
template <int WHICH, int J0, int... Is>
__device__ __forceinline__ void pn_jobs_seq(const char* __restrict__ pn, int moff, int wave, int lane, f4* acc, std::integer_sequence<int, Is...>)
{
    (pn_job<WHICH, J0 + Is>(pn, moff, wave, lane, acc), ...);
}
// all jobs of panel phase PH (0 OUT, 1..5 layers 4..0, 6 DB)
template <int WHICH, int PH>
__device__ __forceinline__ void pn_phase_jobs(const char* __restrict__ pn, int moff, int wave, int lane, f4* acc)
{
    constexpr JobPlan<WHICH> JP{};
    pn_jobs_seq<WHICH, JP.first[PH]>(pn, moff, wave, lane, acc, std::make_integer_sequence<int, JP.first[PH + 1] - JP.first[PH]>{});
}
// store the tiles of job JJ (if this wave owns it) into the workgroup's slab, canonical parameter layout
template <int WHICH, int JJ, bool SKIP_DB = false>
__device__ __forceinline__ void pn_job_flush(float* __restrict__ g_dec, int wave, int lane, const f4* acc)
{
    constexpr JobPlan<WHICH> JP{};
    constexpr TrainPlan<WHICH> plan{};
    constexpr TJob J = JP.j[JJ];
    constexpr int s0 = JP.slot0[JJ];
    if (wave != J.wave) return;
    const int x = lane & 15, g = lane >> 4;
#pragma unroll
    for (int k = 0; k < J.nt; ++k) {
        constexpr int dummy = 0; (void)dummy;
        const TrainPhase P = plan.p[J.ph[k]];
        if (SKIP_DB && J.ph[k] == TrainPlan<WHICH>::P_DB) continue;       // (mapping steps keep d/dB outside the jobs: see the loop's tail)
        const bool rs = J.ch[k] < 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = 16 * J.rt + 4 * g + i;
            const float v = acc[s0 + k][i];
            if (o < P.rows) {
                if (rs) { if (x == 0) g_dec[P.b_base + o] = v; }
                else if (16 * J.ch[k] + x < P.cols_total) g_dec[P.w_base + o * P.ld + P.col0 + 16 * J.ch[k] + x] = v;
            }
        }
    }
}
template <int WHICH, bool SKIP_DB, int... Is>
__device__ __forceinline__ void pn_jobs_flush_all(float* __restrict__ g_dec, int wave, int lane, const f4* acc, std::integer_sequence<int, Is...>)
{
    (pn_job_flush<WHICH, Is, SKIP_DB>(g_dec, wave, lane, acc), ...);
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
    NSK_PH(28);
    float* scratch = smf + PM_IMG_F + wave * 1056;       // per-wave scatter scratch: plane H of rows 0..124 (G1, G2, XC, XH), all rewritten only after the next iteration's first barrier
    static_assert(8 * 1056 * 4 <= PM_E * PN_RB, "scatter scratch must end before the E rows");
    // (the image copy itself sits further down, around the first tile's sample loads: image_issue, nsk_device.h)
    const h8* imgh = reinterpret_cast<const h8*>(smem);
    // The e-part fragments (W0e^T, W3e^T: 12 groups, 24 KB) have no room in LDS beside the panel.  Until round 3 every wave read them from L2 for
    // every tile: 192 KB per iteration through the CU's vector cache, ~3 000 cycles of streaming in front of everything issued behind them
    // (tools/exp_ph3.py).  Now the workgroup copies them ONCE per iteration into panel rows that are dead between layer 0's tiles and the next
    // iteration (plane H of G2 / XC / XH, rows 32..127: 26 112 B): global loads issued before layer 0, LDS stores after its last barrier.
    const f4* gimg_e = reinterpret_cast<const f4*>(A.bimg16) + (size_t)MlpBwdImgH::W0ET * 2 * 1024 / 16;      // groups 18..29, contiguous
    constexpr int EIMG_F4 = 12 * 2 * 1024 / 16;                      // 1536 float4 = 3 per thread
    static_assert(EIMG_F4 == 3 * 512 && EIMG_F4 * 16 <= (PM_E - PM_G2) * PN_RB, "e-part image: three float4 per thread, inside plane H of rows 32..127");
    f4* eimg = reinterpret_cast<f4*>(pn + PM_G2 * PN_RB);
    const h8* eimgh = reinterpret_cast<const h8*>(eimg);
    const float* Wo = smf + PM_IMG_FRAG_F;
    const float* Bm = Wo + 128;

    constexpr JobPlan<WHICH> JP{};
    f4 acc[JP.nslots];
#pragma unroll
    for (int k = 0; k < JP.nslots; ++k) acc[k] = (f4)(0.f);

    const int ntasks = (A.M + 15) >> 4;
    const int iters = tiles_per_wave(ntasks, nb * 8, 0);
    const bool scat = (A.flags & 1u) && A.grid.g && !NSK_DBG(A, 9);
    const bool no_put = NSK_DBG(A, 14), no_tiles = NSK_DBG(A, 13);      // experiment builds only (constant false otherwise); the barriers always stay
    (void)no_put; (void)no_tiles;
    // Staged: what iteration it + 1 needs, fetched during iteration it in three steps, none of which waits for a load it has just issued:
    //   stage_a   (after the chain)        sample data as loaded (z, ray), upstream gradient, ReLU bits, h4   -- issue only
    //   stage_b1  (before phase DB's barrier)  the point p, its cell, and the 16 corner loads of the gather    -- issue only
    //   stage_b2  (after phase DB)         the trilinear reduction of those corners into xc
    // (until round 3 stage_a computed p at once and stage_b reduced its gather at once: ~2 700 + ~5 000 cycles of an iteration spent
    // waiting for round trips to L2 with nothing else to issue -- tools/exp_ph3.py)
    struct Staged { SampleRaw r; float px, py, pz; int mm; bool valid; f4 gr; f4 xc[CQ]; f4 h4[2]; unsigned long long mask; } nx;
    auto task_of = [&](int it_) { return tile_of(it_, bid * 8 + wave, nb * 8, 0); };
    auto slot_of = [&](int it_) { return task_of(it_) * 16 + j; };
    auto stage_a = [&](int it_, int mm, Staged& S_) {
        const int task = task_of(it_);
        const int slot = task * 16 + j;
        S_.valid = slot < A.M;
        S_.mm = mm;
        if (!NSK_DBG(A, 10)) sample_load(A, mm, S_.r);                       // (experiment bits 10, 11, 15: which of these loads stalls the issue -- tools/exp_ph3.py)
        if (!NSK_DBG(A, 11)) S_.gr = ld32<f4>(A.g_raw, (unsigned)mm * 16u);
        const int tk = min(task, ntasks - 1);
        NSK_IDX(3, tk, ntasks); NSK_IDX(2, min(slot, A.M - 1), A.M);
        if (!NSK_DBG(A, 11)) S_.mask = ld32<unsigned long long>(A.masks, ((unsigned)min(slot, A.M - 1) * 4u + (unsigned)g) * 8u);
        if (!NSK_DBG(A, 15)) { S_.h4[0] = A.hsave[((size_t)tk * 10 + 8) * 64 + lane]; S_.h4[1] = A.hsave[((size_t)tk * 10 + 9) * 64 + lane]; }
    };
    Tri Tn; GatherRaw GR;                                // the next tile's cell and its corner lines in flight
    auto stage_b1 = [&](Staged& S_) {
        sample_finish(A, S_.r, S_.px, S_.py, S_.pz);
        tri_setup(A.grid, A.bound, S_.px, S_.py, S_.pz, Tn);
        tri_gather_issue<true>(A.grid, Tn, g, GR);
    };
    auto stage_b2 = [&](Staged& S_) { tri_gather_reduce(Tn, GR, S_.xc[0], S_.xc[1]); };
    int mm_next = 0;
    {
        const f4* src = reinterpret_cast<const f4*>(A.bimg16);
        constexpr int K0 = (PM_IMG_FRAG_F / 4 + 511) / 512;
        ImgRegs<K0> ir0; ImgRegs<1> ir1;
        const int mm0 = slot_sample(A, slot_of(0));
        image_issue<512>(ir0, src, PM_IMG_FRAG_F / 4);
        image_issue<512>(ir1, src + MlpBwdImgH::P_WO / 4, (128 + 288) / 4);
        if (iters > 0) stage_a(0, mm0, nx);
        image_commit<512>(smem, ir0, src, PM_IMG_FRAG_F / 4);
        image_commit<512>(smem + PM_IMG_FRAG_F / 4, ir1, src + MlpBwdImgH::P_WO / 4, (128 + 288) / 4);
        __syncthreads();
        NSK_PH(29);
        if (iters > 0) { stage_b1(nx); stage_b2(nx); mm_next = slot_sample(A, slot_of(1)); }
    }
    asm volatile("" : "+v"(nx.h4[0]), "+v"(nx.h4[1]), "+v"(nx.gr), "+v"(nx.mask), "+v"(mm_next));
    f4 accB[2] = {(f4)(0.f), (f4)(0.f)};                // d loss / d B, this wave's tiles (see the loop's tail)
    (void)accB;
    NSK_PH(30);
    for (int it = 0; it < iters; ++it) {
        asm volatile("" ::: "memory");
        lane = lane0; asm volatile("" : "+v"(lane)); j = lane & 15; g = lane >> 4;      // (see decode_bwd_train_body)
        NSK_PH(0); NSK_PHI(0);
        if (it > 0) lds_barrier();                   // scratch (G1 / G2), XC and E are rewritten from here on
        const bool valid = nx.valid;
        float px = nx.px, py = nx.py, pz = nx.pz, zz = A.pts ? 0.f : nx.r.z; const int n = A.pts ? 0 : ray_of(A, nx.mm);
        (void)zz; (void)n;
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
        NSK_IDX(3, htask, ntasks);
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
            for (int q = 0; q < 6; ++q) if (!no_put) pn_put(pn, PM, PM_E + 16 * q, wave, lane, xe[q]);
        }
        if (!no_put) pn_put(pn, PM, PM_XC, wave, lane, nx.xc[0]);
        if (!no_put) pn_put(pn, PM, PM_XC + 16, wave, lane, nx.xc[1]);
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
        // ---- phase OUT: G = g_out (rows >= OD zero), X = h4 --------------------------------------------------------------------
        {
            if (!no_put) pn_put(pn, PM, PM_G1, wave, lane, go);
            if (!no_put) pn_put(pn, PM, PM_XH, wave, lane, hq[4][0]);
            if (!no_put) pn_put(pn, PM, PM_XH + 16, wave, lane, hq[4][1]);
            lds_barrier();
            if (!no_tiles) pn_phase_jobs<WHICH, JobPlan<WHICH>::PH_OUT>(pn, PM, wave, lane, acc);
            lds_barrier();
        }
        NSK_PH(2); NSK_PHI(2);
        f4 gc[2] = {(f4)(0.f), (f4)(0.f)};
        H2 xa3, xa;
        auto layer = [&](auto LC) {
            constexpr int l = decltype(LC)::value;
            if constexpr (l >= 2) load_h(std::integral_constant<int, l - 2>{});      // a whole layer ahead (see decode_bwd_train_body)
            // g_a = ReLU'(.) g_h: the same half-word masks give the fp16 pieces of g_a (the chain's next operand) from those of g_h and the
            // bf16 panel pieces of g_a from the panel pieces of g_h -- one split each instead of two, no fp32 select
            if constexpr (l == 2) NSK_PH(20);
            const Mask4 RM = relu_mask_dwords(mask, l);
            {
                const H2 xg = split_block_h(gh[0], gh[1]);
                f4 gl[2] = {(f4)(0.f), (f4)(0.f)};
                gemm_h(imgh, MlpBwdImgH::FT(l), lane, xg, gc, gl);               // g_c += fc[l]^T g_h
                gc[0] += gl[0] * (1.f / NSK_H16_SCALE); gc[1] += gl[1] * (1.f / NSK_H16_SCALE);
                xa = mask_block_h4(xg, RM);
            }
            // ---- the layer's phase: dFc_l = g_h c^T (G1 x XC), dW_l = g_a x^T (G2 x XH or E) ------------------------------------
            if constexpr (l == 2) NSK_PH(21);
            if (!no_put) pn_put_masked(pn, PM, PM_G1, PM_G2, wave, lane, gh[0], us, RM.d[0], RM.d[1]);
            if (!no_put) pn_put_masked(pn, PM, PM_G1 + 16, PM_G2 + 16, wave, lane, gh[1], us, RM.d[2], RM.d[3]);
            if constexpr (l >= 1) {
                constexpr int k = l == 3 ? 2 : l - 1;                            // layer 3 reads e (its own rows) and h2
                if (!no_put) pn_put(pn, PM, PM_XH, wave, lane, hq[k][0]);
                if (!no_put) pn_put(pn, PM, PM_XH + 16, wave, lane, hq[k][1]);
            }
            if constexpr (l == 2) NSK_PH(22);
            lds_barrier();
            if constexpr (l == 2) NSK_PH(23);
            if (!no_tiles) pn_phase_jobs<WHICH, 1 + (4 - l)>(pn, PM, wave, lane, acc);       // dFc_l, dW_l (and their bias rows) as jobs: see JobPlan
            if constexpr (l == 2) NSK_PH(24);
            lds_barrier();
            if constexpr (l == 2) NSK_PH(25);
            if constexpr (l == 3) xa3 = xa;
            if constexpr (l >= 1) {
                f4 ghn[2] = {(f4)(0.f), (f4)(0.f)}, ghl[2] = {(f4)(0.f), (f4)(0.f)};
                gemm_h(imgh, MlpBwdImgH::WT(l), lane, xa, ghn, ghl);
                gh[0] = ghn[0] + ghl[0] * (1.f / NSK_H16_SCALE); gh[1] = ghn[1] + ghl[1] * (1.f / NSK_H16_SCALE);
            }
            if constexpr (l == 2) NSK_PH(26);
        };
        layer(std::integral_constant<int, 4>{});
        NSK_PH(3); NSK_PHI(3);
        layer(std::integral_constant<int, 3>{});
        NSK_PH(4); NSK_PHI(4);
        layer(std::integral_constant<int, 2>{});
        NSK_PH(5); NSK_PHI(5);
        layer(std::integral_constant<int, 1>{});
        NSK_PH(6); NSK_PHI(6);
        f4 ecp[3];                                           // this thread's share of the e-part image, in flight across layer 0
#pragma unroll
        for (int u = 0; u < 3; ++u) ecp[u] = gimg_e[u * 512 + (int)threadIdx.x];
        layer(std::integral_constant<int, 0>{});
        NSK_PH(7); NSK_PHI(7);
#pragma unroll
        for (int u = 0; u < 3; ++u) eimg[u * 512 + (int)threadIdx.x] = ecp[u];      // rows 32..127 are dead since layer 0's last barrier
        NSK_PH(14); NSK_PHI(14);
        lds_barrier();
        NSK_PH(15); NSK_PHI(15);
        gc[0] *= us; gc[1] *= us;
        float gp[3] = {0.f, 0.f, 0.f};
        Tri T;
        f4 ge[6];
        if constexpr (!RAYS) {
            // ---- d loss / d B without a panel phase (mapping steps: no ray gradients) ---------------------------------------------------
            // dB[k][f] = sum_s us p_k[s] cos_f[s] g_e[f][s].  Until round 3 this was a seventh panel phase: g_s = g_e cos transposed through LDS
            // (six put calls, 96 16-bit stores per lane), two barriers, six tiles.  The transposed e-part product leaves g_e with the D layout
            // "one feature, four samples" per lane, which IS the A operand of v_mfma_f32_16x16x16_bf16; the cosines are computed in that layout
            // (same arguments, same instructions: same values), and B = [us p] for the lane's (feature block, coordinate) column: the sums over
            // the wave's 16 samples come out of 24 K=16 products per tile, accumulated over the wave's tiles in 8 registers, and the waves'
            // partial sums meet once, in LDS, after the loop.  Column n = 3 (q mod 5) + k of accumulator q / 5 holds block q, coordinate k.
            if (it + 1 < iters) { stage_a(it + 1, mm_next, nx); mm_next = slot_sample(A, slot_of(it + 2)); }
            gemm_e2T_lds(eimgh, lane, xa3, xa, ge);
            NSK_PH(16); NSK_PHI(16);
            stage_b1(nx);
            NSK_PH(8); NSK_PHI(8);
            float* xch = reinterpret_cast<float*>(pn + PM + PM_G1 * PN_RB) + wave * 64;          // plane M of the G rows: dead since layer 0's tiles
            if (g == 0) *reinterpret_cast<f4*>(xch + 4 * j) = (f4){px, py, pz, valid ? us : 0.f};
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int nb3 = j / 3, nc = j - 3 * nb3;                                              // this lane's B column: feature block (mod 5), coordinate
            unsigned ph01, pm01, ph23, pm23;
            f4 sp[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) sp[i] = *reinterpret_cast<const f4*>(xch + 4 * (4 * g + i));
            {
                float pv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) pv[i] = (nc == 0 ? sp[i][0] : (nc == 1 ? sp[i][1] : sp[i][2])) * sp[i][3];
                split2_pair(pv[0], pv[1], ph01, pm01); split2_pair(pv[2], pv[3], ph23, pm23);
            }
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const int f = 16 * q + j;
                const float b0 = Bm[f], b1 = Bm[96 + f], b2 = Bm[192 + f];
                float gs[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float s = add_rn(add_rn(mul_rn(sp[i][0], b0), mul_rn(sp[i][1], b1)), mul_rn(sp[i][2], b2));
                    const float cv = __builtin_amdgcn_cosf(nsk_rev(s));
                    gs[i] = f < NSK_E ? ge[q][i] * cv : 0.f;
                }
                unsigned ah01, am01, ah23, am23;
                split2_pair(gs[0], gs[1], ah01, am01); split2_pair(gs[2], gs[3], ah23, am23);
                const bool on = nb3 == (q % 5);
                const unsigned bh01 = on ? ph01 : 0u, bh23 = on ? ph23 : 0u, bm01 = on ? pm01 : 0u, bm23 = on ? pm23 : 0u;
                f4& d = accB[q / 5];
                d = mfma_bf16_k16(ah01, ah23, bh01, bh23, d);
                d = mfma_bf16_k16(ah01, ah23, bm01, bm23, d);
                d = mfma_bf16_k16(am01, am23, bh01, bh23, d);
                d = mfma_bf16_k16(am01, am23, bm01, bm23, d);
            }
            tri_setup(A.grid, A.bound, px, py, pz, T);
        } else {
#pragma unroll
            for (int q = 0; q < 6; ++q) ge[q] = (f4)(0.f);
            gemm_e2_lds(eimgh, lane, xa3, xa, ge);              // g_e = W3e^T g_a3 + W0e^T g_a0 (still carries the sample's scale)
            NSK_PH(16); NSK_PHI(16);
            // the next tile's sample data: issued (not waited for) behind the e-part fragments -- loads return in order, and in front of them
            // these (scattered, often beyond L2) made the first product wait for their round trip -- and ahead of the cosines and panel stores below
            if (it + 1 < iters) { stage_a(it + 1, mm_next, nx); mm_next = slot_sample(A, slot_of(it + 2)); }
            NSK_PH(8); NSK_PHI(8);
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
                if (!no_put) pn_put(pn, PM, PM_G1, wave, lane, pq, us);
#pragma unroll
                for (int q = 0; q < 6; ++q) if (!no_put) pn_put(pn, PM, PM_E + 16 * q, wave, lane, ge[q]);
                // the next tile's gather goes out here: its 16 corner lines travel while the workgroup meets at the barrier and runs the tiles
                // (unconditional, like its reduction below: under `if (it + 1 < iters)` the compiler cannot tell that both run or neither, keeps the 64
                // corner registers alive around the whole loop and spills 85 of them; in the last iteration nx still holds this tile's sample,
                // so the extra gather reads valid lines and its result is never used)
                stage_b1(nx);
                lds_barrier();
                if (!no_tiles) pn_phase_jobs<WHICH, JobPlan<WHICH>::PH_DB>(pn, PM, wave, lane, acc);
                lds_barrier();
            }
        }
        NSK_PH(9); NSK_PHI(9);
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
        stage_b2(nx);
        // every staged load must have landed before the first atomic below (see decode_bwd_train_body)
        asm volatile("" : "+v"(nx.h4[0]), "+v"(nx.h4[1]), "+v"(nx.gr), "+v"(nx.mask), "+v"(mm_next), "+v"(nx.xc[0]), "+v"(nx.xc[1]));
        NSK_PH(10); NSK_PHI(10);
        if (scat) {
            if (A.flags & 0x8000u) {        // deterministic debug mode: see decode_bwd_body
                for (int w = 0; w < 8; ++w) { if (wave == w) { scatter_tile(A.grid, T, gc, lane, valid, scratch); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } __syncthreads(); }
            } else scatter_tile(A.grid, T, gc, lane, valid, scratch);
        }
        NSK_PH(11); NSK_PHI(11);
    }
    NSK_PH(12);
    // ---- single flush of this wave's output tiles (job by job) ---------------------------------------
    float* slab = A.g_dec + (size_t)bid * ((plan_total<WHICH>() + 3) & ~3);
    pn_jobs_flush_all<WHICH, !RAYS>(slab, wave, lane, acc, std::make_integer_sequence<int, JobPlan<WHICH>::NJ>{});
    if constexpr (!RAYS) {
        // d loss / d B: the eight waves' partial sums meet in LDS (the panel is dead), 128 lanes add them up and store [3][93]
        __syncthreads();
        f4* red = reinterpret_cast<f4*>(pn);
        red[(wave * 2 + 0) * 64 + lane] = accB[0]; red[(wave * 2 + 1) * 64 + lane] = accB[1];
        __syncthreads();
        if (threadIdx.x < 128) {
            const int a = threadIdx.x >> 6;
            f4 v = red[a * 64 + lane];
#pragma unroll
            for (int w = 1; w < 8; ++w) v += red[(w * 2 + a) * 64 + lane];
            constexpr TrainPhase P = plan.p[PL::P_DB];
            const int q = a == 0 ? j / 3 : 5, k = j - 3 * (j / 3);
            if (a == 0 ? j < 15 : j < 3) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int f = 16 * q + 4 * g + i;
                    if (f < NSK_E) slab[P.w_base + k * P.ld + f] = v[i];
                }
            }
        }
    }
    NSK_PH(13);
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
    if constexpr (!RAYS) {
        // the NEXT batch's cell-sort offsets (nsk_map_prepare): short workgroups behind the roles', two 256-cell chunks each; they start when the last
        // frozen workgroup has been placed and need nothing of this launch
        if ((int)blockIdx.x >= MA.wg_end[MA.n - 1]) { sort_scan_body<2>(MA.scan, (int)blockIdx.x - MA.wg_end[MA.n - 1]); return; }
    }
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

// The same launch with every chain on the fp32 MFMA (nsk_set_backward_mode 0: full-width operands, what the reference's fp32 autograd multiplies,
// src/Mapper.cpp:443-444): frozen roles = the body the ray-gradient launches use, trainable role = the one-phase-per-weight body with fp32
// fragments.  A measuring stick (tests, bench extras), not the product path: 240 fp32 MFMAs of 32 cycles per trainable tile instead of 90 of 16.
template <bool RAYS>
__global__ __launch_bounds__(512) void k_decode_bwd_multi_full(MultiArgs MA)
{
    if (MA.sum_n > 0 && blockIdx.x == gridDim.x - 1) { block_sum(MA.sum_src, MA.sum_n, MA.sum_dst); return; }
    int r = 0;
    while (r < MA.n - 1 && (int)blockIdx.x >= MA.wg_end[r]) ++r;
    const int b0 = r == 0 ? 0 : MA.wg_end[r - 1];
    const int bid = blockIdx.x - b0, nb = MA.wg_end[r] - b0;
    const int sel = MA.which[r] * 2 + (MA.train[r] ? 1 : 0);
    switch (sel) {
    case 0: decode_bwd_body<0, RAYS, 8, true>(MA.a[r], bid, nb); break;
    case 1: decode_bwd_train_body<0, RAYS, true>(MA.a[r], bid, nb); break;
    case 2: decode_bwd_body<1, RAYS, 8, true>(MA.a[r], bid, nb); break;
    case 3: decode_bwd_train_body<1, RAYS, true>(MA.a[r], bid, nb); break;
    case 4: decode_bwd_body<2, RAYS, 8, true>(MA.a[r], bid, nb); break;
    case 5: break;     // (the trainable fine decoder is launched on its own: see k_decode_bwd_multi)
    case 6: decode_bwd_body<3, RAYS, 8, true>(MA.a[r], bid, nb); break;
    default: decode_bwd_train_body<3, RAYS, true>(MA.a[r], bid, nb); break;
    }
}
