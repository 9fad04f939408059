// microbenchmark: does VALU fp32 work hide in the shadow of v_mfma_f32_16x16x4_f32 (32 cyc/SIMD) or add to it?
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu mfma_valu.hip ; run: ./mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int F, int BF16>
__global__ void k(float* out, long long* cyc, int iters)
{
    f4 d0 = {0, 0, 0, 0}, d1 = d0, d2 = d0, d3 = d0;
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
    bf8 pa, pb;
    for (int i = 0; i < 8; ++i) { pa[i] = (__bf16)(a + i); pb[i] = (__bf16)(b + i); }
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#define VAL(n) if (F > n) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[n & 7]) : "v"(b));
#define MF(d) if (BF16) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(pa), "v"(pb)); \
              else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
#define GRP(d) MF(d) VAL(0) VAL(1) VAL(2) VAL(3) VAL(4) VAL(5) VAL(6) VAL(7) VAL(8) VAL(9) VAL(10) VAL(11) VAL(12) VAL(13) VAL(14) VAL(15)
        GRP(d0) GRP(d1) GRP(d2) GRP(d3)
    }
    long long t1 = __builtin_readcyclecounter();
    f4 s = d0 + d1 + d2 + d3;
    float r = s[0] + s[1] + s[2] + s[3];
    for (int i = 0; i < 8; ++i) r += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int F, int BF16>
void run(int threads, float* out, long long* cyc)
{
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<F, BF16><<<256, threads>>>(out, cyc, 10);
    hipEventRecord(e0);
    k<F, BF16><<<256, threads>>>(out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%s waves/SIMD=%d  VALU per MFMA=%2d : %.1f ns per MFMA-group (wall), %.1f memtime ticks per group\n", BF16 ? "bf16 16x16x32" : "f32 16x16x4 ", threads / 256, F,
           1e6 * ms / (iters * 4.0), (double)c / (iters * 4.0));
}
int main()
{
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
    for (int threads : {256, 512}) {
        run<0, 0>(threads, out, cyc); run<2, 0>(threads, out, cyc); run<4, 0>(threads, out, cyc); run<6, 0>(threads, out, cyc);
        run<8, 0>(threads, out, cyc); run<12, 0>(threads, out, cyc); run<16, 0>(threads, out, cyc);
        run<0, 1>(threads, out, cyc); run<2, 1>(threads, out, cyc); run<4, 1>(threads, out, cyc); run<8, 1>(threads, out, cyc);
    }
    return 0;
}
