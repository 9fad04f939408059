// nsk_layout.h -- device-side data layouts shared by the host packer and the gfx950 kernels.
//
// MFMA tiling (v_mfma_f32_16x16x4_f32, exact fp32): every activation tile is [16 features] x [16 samples]
// held in the accumulator ("D") layout: lane l = (j = l&15 -> sample, g = l>>4), register i -> feature 4g+i.
// The same registers are fed back as the B operand of the next layer (B[k][j]: lane supplies k-group l>>4),
// so a K=16 block ("quad") of any contraction is four MFMA steps i=0..3 with k = 16q + 4g + i.  Every weight
// matrix is therefore stored as "A fragments" in exactly the order the lanes consume them:
//     image[((quad0 + rt*KQ + q) * 64 + lane) * 4 + i] = W[row = 16 rt + (lane&15)][k = 16 q + 4 (lane>>4) + i]
// (one 16-byte read per lane per quad, conflict free), rt = 16-row tile of the output, q = quad of K.
// A "quad" is 64 lanes x 4 floats = 256 floats = 1 KiB.
#pragma once

#define NSK_E 93        // GaussianFFT mapping size (reference src/models/MLP.cpp:21)
#define NSK_EP 96       // padded to 6 quads
#define NSK_HID 32        // hidden size (reference src/main.cpp:29)

// ---- forward image of MLP (middle / fine / color), CQ = c_dim/16 --------------------------------------
template <int CQ>
struct MlpFwdImg {
    static constexpr int W0E = 0;               // [32 x 96]   rt=2, kq=6
    static constexpr int F0 = W0E + 12;         // [32 x 16CQ] rt=2, kq=CQ
    static constexpr int W1 = F0 + 2 * CQ;      // [32 x 32]
    static constexpr int F1 = W1 + 4;
    static constexpr int W2 = F1 + 2 * CQ;
    static constexpr int F2 = W2 + 4;
    static constexpr int W3E = F2 + 2 * CQ;     // pts_linear[3][:, 0:93]
    static constexpr int W3H = W3E + 12;        // pts_linear[3][:, 93:125]
    static constexpr int F3 = W3H + 4;
    static constexpr int W4 = F3 + 2 * CQ;
    static constexpr int F4 = W4 + 4;
    static constexpr int NQ = F4 + 2 * CQ;
    static constexpr int P_B = NQ * 256;        // pts_linear bias [5][32]
    static constexpr int P_BC = P_B + 160;      // fc bias        [5][32]
    static constexpr int P_WO = P_BC + 160;     // output weight  [4][32] (rows >= out_dim zero)
    static constexpr int P_BO = P_WO + 128;     // output bias    [4]
    static constexpr int P_BM = P_BO + 4;       // embedding B    [3][96] (cols >= 93 zero)
    static constexpr int TOTAL = P_BM + 288;    // floats, multiple of 4
    __host__ __device__ static constexpr int W(int l) { return l == 1 ? W1 : (l == 2 ? W2 : (l == 4 ? W4 : -1)); }
    __host__ __device__ static constexpr int F(int l) { return l == 0 ? F0 : (l == 1 ? F1 : (l == 2 ? F2 : (l == 3 ? F3 : F4))); }
};

// ---- backward image of MLP: transposed fragments, K = 32 output features ------------------------------
struct MlpBwdImg {
    static constexpr int FT0 = 0;               // fc[l]^T      [32 c-rows x 32]  rt=2,kq=2  (5 of them)
    static constexpr int WT1 = 20;              // pts_linear[l]^T h-part [32 x 32], l=1..4 (4 of them)
    static constexpr int W0ET = 36;             // pts_linear[0]^T [96 x 32] rt=6,kq=2
    static constexpr int W3ET = 48;             // pts_linear[3][:, 0:93]^T
    static constexpr int NQ = 60;
    static constexpr int P_WO = NQ * 256;       // output weight [4][32]
    static constexpr int P_BM = P_WO + 128;     // embedding B [3][96]
    static constexpr int TOTAL = P_BM + 288;
    __host__ __device__ static constexpr int FT(int l) { return FT0 + 4 * l; }
    __host__ __device__ static constexpr int WT(int l) { return WT1 + 4 * (l - 1); }
};

// ---- coarse decoder (MLP_no_xyz) ----------------------------------------------------------------------
struct CoarseFwdImg {
    static constexpr int W0 = 0, W1 = 4, W2 = 8, W3C = 12, W3H = 16, W4 = 20, NQ = 24;
    static constexpr int P_B = NQ * 256;        // bias [5][32]
    static constexpr int P_WO = P_B + 160;      // [4][32]
    static constexpr int P_BO = P_WO + 128;
    static constexpr int TOTAL = P_BO + 4;
};
struct CoarseBwdImg {
    static constexpr int W0T = 0, W1T = 4, W2T = 8, W3CT = 12, W3HT = 16, W4T = 20, NQ = 24;
    static constexpr int P_WO = NQ * 256;
    static constexpr int TOTAL = P_WO + 128;
};

// canonical (torch) packed parameter offsets, identical to oracle/nso.c make_layout
struct DecLayout {
    int has_xyz, c_dim, out_dim, in_dim[5];
    int oB, oW[5], ob[5], oFw[5], oFb[5], oWo, obo, total;
};
inline DecLayout nsk_dec_layout(int which)
{
    DecLayout L{};
    L.has_xyz = which != 0;
    L.c_dim = which == 2 ? 64 : 32;
    L.out_dim = which == 3 ? 4 : 1;
    int o = 0;
    if (L.has_xyz) {
        int d[5] = {NSK_E, NSK_HID, NSK_HID, NSK_HID + NSK_E, NSK_HID};
        for (int i = 0; i < 5; ++i) L.in_dim[i] = d[i];
        L.oB = o; o += 3 * NSK_E;
    } else {
        int d[5] = {32, NSK_HID, NSK_HID, NSK_HID + 32, NSK_HID};
        for (int i = 0; i < 5; ++i) L.in_dim[i] = d[i];
    }
    for (int i = 0; i < 5; ++i) { L.oW[i] = o; o += NSK_HID * L.in_dim[i]; L.ob[i] = o; o += NSK_HID; }
    if (L.has_xyz)
        for (int i = 0; i < 5; ++i) { L.oFw[i] = o; o += NSK_HID * L.c_dim; L.oFb[i] = o; o += NSK_HID; }
    L.oWo = o; o += L.out_dim * NSK_HID; L.obo = o; o += L.out_dim;
    L.total = o;
    return L;
}

// constexpr twin of nsk_dec_layout for device code
struct DecLayoutDev {
    int in_dim[5];
    int oB, oW[5], ob[5], oFw[5], oFb[5], oWo, obo, total;
};
template <int WHICH>
__host__ __device__ constexpr DecLayoutDev dec_layout_dev()
{
    DecLayoutDev L{};
    constexpr bool xyz = WHICH != 0;
    constexpr int c_dim = WHICH == 2 ? 64 : 32;
    constexpr int out_dim = WHICH == 3 ? 4 : 1;
    int o = 0;
    if (xyz) { L.in_dim[0] = NSK_E; L.in_dim[1] = NSK_HID; L.in_dim[2] = NSK_HID; L.in_dim[3] = NSK_HID + NSK_E; L.in_dim[4] = NSK_HID; L.oB = o; o += 3 * NSK_E; }
    else { L.in_dim[0] = 32; L.in_dim[1] = NSK_HID; L.in_dim[2] = NSK_HID; L.in_dim[3] = NSK_HID + 32; L.in_dim[4] = NSK_HID; }
    for (int i = 0; i < 5; ++i) { L.oW[i] = o; o += NSK_HID * L.in_dim[i]; L.ob[i] = o; o += NSK_HID; }
    if (xyz) for (int i = 0; i < 5; ++i) { L.oFw[i] = o; o += NSK_HID * c_dim; L.oFb[i] = o; o += NSK_HID; }
    L.oWo = o; o += out_dim * NSK_HID; L.obo = o; o += out_dim;
    L.total = o;
    return L;
}
