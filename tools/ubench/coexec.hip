// micro-benchmark: do bf16 MFMA chains of one wave and vector ALU work of another wave on the same SIMD overlap on gfx950?
// build: hipcc --offload-arch=gfx950 -O3 -o coexec coexec.hip ; run: ./coexec
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f4;

// role[w] for wave w of the workgroup: 0 idle, 1 MFMA chain (nacc independent accumulators), 2 FMA stream, 3 alternate phases (MFMA then FMA)
struct Roles { int r[8]; int nacc; int iters; int phase; };

__global__ __launch_bounds__(512) void k(Roles R, float* out)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int role = R.r[wave];
    if (role == 0) return;
    bf8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(lane + i); b[i] = (__bf16)(float)(lane * 3 + i); }
    f4 acc[4] = {};
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = lane * 0.001f + i;
    const float m = 1.0001f, c = 0.0001f;
    auto do_mfma = [&](int n) {
        for (int it = 0; it < n; ++it) {
            if (R.nacc == 1) {
#pragma unroll
                for (int u = 0; u < 8; ++u) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(a), "v"(b));
            } else if (R.nacc == 2) {
#pragma unroll
                for (int u = 0; u < 4; ++u) { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(a), "v"(b)); asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[1]) : "v"(a), "v"(b)); }
            } else {
#pragma unroll
                for (int u = 0; u < 2; ++u) { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(a), "v"(b)); asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[1]) : "v"(a), "v"(b));
                                              asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[2]) : "v"(a), "v"(b)); asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[3]) : "v"(a), "v"(b)); }
            }
        }
    };
    auto do_fma = [&](int n) {            // 8 independent chains: 32 FMAs per iteration, issue-bound
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(m), "v"(c));
        }
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (role == 1) do_mfma(R.iters);
    else if (role == 2) do_fma(R.iters);
    else {                                   // phases: `phase` iterations of one kind, then of the other; role 4 starts with the other kind
        const int rounds = R.iters / R.phase;
        for (int q = 0; q < rounds; ++q) {
            if (role == 3) { do_mfma(R.phase); do_fma(R.phase); }
            else { do_fma(R.phase); do_mfma(R.phase); }
        }
    }
    asm volatile("s_nop 0" ::: "memory");
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    for (int q = 0; q < 4; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (s == 12345.678f) out[threadIdx.x] = s;
    if (blockIdx.x == 7 && lane == 0) reinterpret_cast<unsigned long long*>(out + 1024)[wave] = t1 - t0;
}

static unsigned long long g_c0, g_c1;
static float run(Roles R, float* d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<256, 512>>>(R, d);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<<<256, 512>>>(R, d);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long cyc[8]; hipMemcpy(cyc, d + 1024, 64, hipMemcpyDeviceToHost);
    g_c0 = 0; g_c1 = 0;
    for (int w = 0; w < 8; ++w) { if (R.r[w] == 1 || R.r[w] == 3) g_c0 = cyc[w] > g_c0 ? cyc[w] : g_c0; if (R.r[w] == 2 || R.r[w] == 4) g_c1 = cyc[w] > g_c1 ? cyc[w] : g_c1; }
    return ms * 1000.f;
}

int main()
{
    float* d; hipMalloc(&d, 8192); hipMemset(d, 0, 8192);
    const int IT = 20000;
    auto mk = [&](std::initializer_list<int> r, int nacc, int phase = 100) { Roles R; int i = 0; for (int v : r) R.r[i++] = v; R.nacc = nacc; R.iters = IT; R.phase = phase; return R; };
    // 8 MFMAs per iteration: 16-cycle MFMA -> 128 cycles/iter alone at full rate; 32 FMAs per iteration -> 128 cycles/iter
    printf("cycles per iteration at 2.4 GHz = us * 2400 / %d\n", IT);
    struct { const char* name; Roles R; } T[] = {
        {"mfma chain, 1 acc, wave 0 only", mk({1,0,0,0,0,0,0,0}, 1)},
        {"mfma chain, 2 acc, wave 0 only", mk({1,0,0,0,0,0,0,0}, 2)},
        {"mfma chain, 4 acc, wave 0 only", mk({1,0,0,0,0,0,0,0}, 4)},
        {"mfma 1 acc, waves 0 and 4", mk({1,0,0,0,1,0,0,0}, 1)},
        {"mfma 1 acc, waves 0 and 1", mk({1,1,0,0,0,0,0,0}, 1)},
        {"mfma 1 acc, waves 0 and 2", mk({1,0,1,0,0,0,0,0}, 1)},
        {"mfma 2 acc, waves 0 and 4", mk({1,0,0,0,1,0,0,0}, 2)},
        {"fma stream, wave 0 only", mk({2,0,0,0,0,0,0,0}, 1)},
        {"fma stream, waves 0 and 4", mk({2,0,0,0,2,0,0,0}, 1)},
        {"fma stream, waves 0 and 1", mk({2,2,0,0,0,0,0,0}, 1)},
        {"mfma 1 acc wave 0 + fma wave 4", mk({1,0,0,0,2,0,0,0}, 1)},
        {"mfma 2 acc wave 0 + fma wave 4", mk({1,0,0,0,2,0,0,0}, 2)},
        {"mfma 1 acc wave 0 + fma wave 1", mk({1,2,0,0,0,0,0,0}, 1)},
        {"all 8: waves 0-3 mfma 1 acc, 4-7 fma", mk({1,1,1,1,2,2,2,2}, 1)},
        {"all 8 mfma 1 acc", mk({1,1,1,1,1,1,1,1}, 1)},
        {"all 8 fma", mk({2,2,2,2,2,2,2,2}, 1)},
        {"phased in step (0,4 both mfma-then-fma), phase 25", mk({3,0,0,0,3,0,0,0}, 1, 25)},
        {"phased opposite (0 mfma-first, 4 fma-first), phase 25", mk({3,0,0,0,4,0,0,0}, 1, 25)},
        {"phased in step, all 8, phase 25", mk({3,3,3,3,3,3,3,3}, 1, 25)},
        {"phased opposite, all 8, phase 25", mk({3,3,3,3,4,4,4,4}, 1, 25)},
    };
    for (auto& t : T) { float us = run(t.R, d); printf("%-58s %9.1f us | s_memtime/iter: mfma-or-role3 waves %7.1f  fma-or-role4 waves %7.1f\n", t.name, us, (double)g_c0 / IT, (double)g_c1 / IT); }
    return 0;
}
