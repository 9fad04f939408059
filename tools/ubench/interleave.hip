// micro-benchmark: how many independent vector-ALU instructions fit behind one bf16 MFMA of the same wave on gfx950, and what is the
// MFMA rate of a SIMD with 1, 2, 3, 4 waves issuing them?   build: hipcc --offload-arch=gfx950 -O3 -o interleave interleave.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f4;

template <int K, int NACC>
__global__ __launch_bounds__(1024) void k(int iters, unsigned wavemask, unsigned long long* out)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (!((wavemask >> wave) & 1)) return;
    bf8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(lane + i); b[i] = (__bf16)(float)(lane * 3 + i); }
    f4 acc[2] = {};
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = lane * 0.001f + i;
    const float m = 1.0001f, c = 0.0001f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[NACC == 2 ? (u & 1) : 0]) : "v"(a), "v"(b));
#pragma unroll
            for (int i = 0; i < K; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i & 7]) : "v"(m), "v"(c));
        }
    }
    asm volatile("s_nop 0" ::: "memory");
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    for (int q = 0; q < 2; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (s == 12345.678f) out[100 + threadIdx.x] = (unsigned long long)s;
    if (blockIdx.x == 7 && lane == 0) out[wave] = t1 - t0;
}

template <int K, int NACC>
static void run(const char* name, unsigned wavemask, unsigned long long* d)
{
    const int IT = 20000;
    hipMemset(d, 0, 16 * 8);
    k<K, NACC><<<256, 1024>>>(IT, wavemask, d);
    hipDeviceSynchronize();
    k<K, NACC><<<256, 1024>>>(IT, wavemask, d);
    hipDeviceSynchronize();
    unsigned long long cyc[16]; hipMemcpy(cyc, d, 128, hipMemcpyDeviceToHost);
    unsigned long long mx = 0; for (int w = 0; w < 16; ++w) mx = cyc[w] > mx ? cyc[w] : mx;
    printf("%-44s K=%2d acc=%d : %7.1f ticks per MFMA group (8 groups/iter)\n", name, K, NACC, (double)mx / IT / 8);
}

int main()
{
    unsigned long long* d; hipMalloc(&d, 16384);
#define ROW(K) run<K, 1>("1 wave/SIMD (wave 0)", 0x1, d); run<K, 2>("1 wave/SIMD (wave 0)", 0x1, d); \
               run<K, 1>("2 waves/SIMD (0,4)", 0x11, d); run<K, 1>("3 waves/SIMD (0,4,8)", 0x111, d); run<K, 1>("4 waves/SIMD (0,4,8,12)", 0x1111, d); \
               run<K, 1>("8 waves, 2/SIMD", 0xff, d); run<K, 1>("16 waves, 4/SIMD", 0xffff, d);
    ROW(0) ROW(2) ROW(4) ROW(6) ROW(8) ROW(12)
    return 0;
}
