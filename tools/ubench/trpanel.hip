// micro-benchmark and semantics check for the [sample][feature] bf16 panel of nsk_train.h: packed ds_write_b64 transposing stores from
// the MFMA accumulator layout, operands read back with ds_read_b64_tr_b16.  Checks every element and times both directions.
// build: hipcc --offload-arch=gfx950 -O3 -o trpanel trpanel.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s4 lds_s4;
#define S_DW 88                       // dwords per sample row (160 features = 80 dwords + 8 pad): 88 = 8 * 11
__device__ __forceinline__ int pn_addr(int s, int fdw)      // byte address of dword `fdw` (two features) of sample s
{
    const int piece = fdw >> 3, chunk = (fdw >> 1) & 3, d = fdw & 1;
    return 4 * (s * S_DW + piece * 8 + ((chunk ^ ((s >> 2) & 3)) << 1) + d);
}
__global__ __launch_bounds__(512) void k(unsigned short* out, unsigned long long* cyc, int reps)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    // ---- stores: quad q0 (16 features) of this wave's 16 samples; lane (j, g) holds features 16 q0 + 4g + i of sample 16 wave + j
    const int s = 16 * wave + j;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < reps; ++rep)
#pragma unroll
        for (int q0 = 0; q0 < 10; ++q0) {
            s4 v;
            for (int i = 0; i < 4; ++i) v[i] = (short)(s * 160 + 16 * q0 + 4 * g + i);
            *reinterpret_cast<s4*>(smem + pn_addr(s, 8 * q0 + 2 * g)) = v;
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    // ---- transposed reads: tile t = 16 features, k-step kb = 32 samples; lane (r = lane & 15, sq = lane >> 4) must end up with
    // samples 32 kb + 4 sq + (0..3) [first read] and 32 kb + 16 + 4 sq + (0..3) [second read] of feature 16 t + r
    const int q = (lane >> 2) & 3, p = lane & 3, sq = lane >> 4;
    s4 acc = (s4)(0);
    unsigned long long t2 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < reps; ++rep)
#pragma unroll
        for (int t = 0; t < 10; ++t)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const int s1 = 32 * kb + 4 * sq + q;
                const s4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(smem + pn_addr(s1, 8 * t + 2 * p)));
                const s4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(smem + pn_addr(s1 + 16, 8 * t + 2 * p)));
                if (rep == reps - 1 && t == (wave % 10)) {
                    unsigned short* o = out + ((((size_t)wave * 4 + kb) * 64 + lane) * 8);
                    for (int i = 0; i < 4; ++i) { o[i] = (unsigned short)a[i]; o[4 + i] = (unsigned short)b[i]; }
                }
                acc += a + b;
            }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned long long t3 = __builtin_amdgcn_s_memtime();
    if (acc[0] == 12345 && acc[1] == 777) out[0] = 1;
    if (lane == 0 && blockIdx.x == 0) { cyc[2 * wave] = t1 - t0; cyc[2 * wave + 1] = t3 - t2; }
}
int main()
{
    const int reps = 200;
    unsigned short* d; unsigned long long* c;
    hipMalloc(&d, 8 * 4 * 64 * 8 * 2); hipMalloc(&c, 16 * 8);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * S_DW * 4);
    k<<<256, 512, 128 * S_DW * 4>>>(d, c, reps);
    hipDeviceSynchronize();
    std::vector<unsigned short> h(8 * 4 * 64 * 8); std::vector<unsigned long long> hc(16);
    hipMemcpy(h.data(), d, h.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(hc.data(), c, 128, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < 8; ++w) for (int kb = 0; kb < 4; ++kb) for (int lane = 0; lane < 64; ++lane) for (int e = 0; e < 8; ++e) {
        const int r = lane & 15, sq = lane >> 4, t = w % 10;
        const int smp = 32 * kb + (e >= 4 ? 16 : 0) + 4 * sq + (e & 3);
        const unsigned short want = (unsigned short)(smp * 160 + 16 * t + r);
        const unsigned short got = h[(((size_t)w * 4 + kb) * 64 + lane) * 8 + e];
        if (got != want && bad++ < 8) printf("wave %d kb %d lane %d e %d: got %u (sample %u feature %u) want sample %d feature %d\n", w, kb, lane, e, got, got / 160, got % 160, smp, 16 * t + r);
    }
    printf("%s: %d mismatches\n", bad ? "FAILED" : "transposed panel ok", bad);
    for (int w = 0; w < 8; ++w)
        printf("wave %d: %.1f cycles per ds_write_b64 (10/rep), %.1f cycles per tr read (80/rep); 8 waves share the CU\n", w, (double)hc[2 * w] / (reps * 10), (double)hc[2 * w + 1] / (reps * 80));
    return bad != 0;
}
