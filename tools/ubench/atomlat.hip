// micro-benchmark: how long does a global load wait behind fire-and-forget float atomics of the same wave (vmcnt retires in order)?
// Every wave adds NA x 256 B (64 lanes x 4 B on NA different 256-byte segments, like one scatter flush) and then loads one dword.
// build: hipcc --offload-arch=gfx950 -O3 -o atomlat atomlat.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(float* grid, const float* src, size_t nseg, int na, int reps, unsigned long long* out, float* sink)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t gw = (size_t)blockIdx.x * 8 + wave;
    unsigned long long tot = 0;
    float acc = 0.f;
    for (int r = 0; r < reps; ++r) {
        unsigned long long h = (gw * 2654435761ull + (unsigned long long)r * 40503ull);
        for (int a = 0; a < na; ++a) {
            const size_t seg = (h + (unsigned long long)a * 7919ull) % nseg;
            atomicAdd(grid + seg * 64 + lane, 1.0f);          // no return value: fire and forget, but counted by vmcnt
        }
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        const float v = src[(h % nseg) * 64 + lane];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        acc += v;
        tot += t1 - t0;
        __builtin_amdgcn_s_sleep(20);                          // ~1300 cycles of "other work" between rounds
    }
    if (lane == 0) out[gw] = tot;
    if (acc == 12345.f) sink[0] = acc;
}
int main()
{
    const size_t nseg = 1 << 16;                               // 16 MB of gradient lines
    float *grid, *src, *sink; unsigned long long* out;
    hipMalloc(&grid, nseg * 256); hipMalloc(&src, nseg * 256); hipMalloc(&sink, 4); hipMalloc(&out, 256 * 8 * 8);
    hipMemset(grid, 0, nseg * 256); hipMemset(src, 0, nseg * 256);
    const int reps = 200;
    for (int blocks : {1, 64, 256})
        for (int na : {0, 1, 8, 16}) {
            k<<<blocks, 512>>>(grid, src, nseg, na, reps, out, sink);
            hipDeviceSynchronize();
            unsigned long long h[256 * 8]; hipMemcpy(h, out, (size_t)blocks * 8 * 8, hipMemcpyDeviceToHost);
            double s = 0; for (int i = 0; i < blocks * 8; ++i) s += h[i];
            printf("%3d workgroups x 8 waves, %2d atomics before the load: %.0f cycles from load issue to vmcnt(0)\n", blocks, na, s / (blocks * 8) / reps);
        }
    return 0;
}
