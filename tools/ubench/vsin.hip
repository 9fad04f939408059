// accuracy of the hardware v_sin_f32 / v_cos_f32 (argument in revolutions) on gfx950 against double precision
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(int n, const float* f, float* s, float* c) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) { s[i] = __builtin_amdgcn_sinf(f[i]); c[i] = __builtin_amdgcn_cosf(f[i]); } }
int main()
{
    const int n = 1 << 22;
    std::vector<float> f(n), s(n), c(n);
    std::mt19937 g(1);
    std::uniform_real_distribution<float> u(-0.5f, 0.5f);
    for (int i = 0; i < n; ++i) f[i] = u(g);
    for (int i = 0; i < 4096; ++i) f[i] = (i - 2048) / 4096.0f * 1e-3f;         // near zero
    float *df, *ds, *dc; hipMalloc(&df, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dc, n * 4);
    hipMemcpy(df, f.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(n, df, ds, dc);
    hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost); hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
    double es = 0, ec = 0, rs = 0; int is = 0, ic = 0;
    for (int i = 0; i < n; ++i) {
        const double x = 2.0 * M_PI * (double)f[i];
        const double a = fabs((double)s[i] - sin(x)), b = fabs((double)c[i] - cos(x));
        if (a > es) { es = a; is = i; }
        if (b > ec) { ec = b; ic = i; }
        if (fabs(sin(x)) > 1e-30) rs = fmax(rs, a / fabs(sin(x)));
    }
    printf("v_sin_f32: max abs error %.3e at f=%.8f (sin %.8f); max rel error %.3e\nv_cos_f32: max abs error %.3e at f=%.8f\n", es, f[is], s[is], rs, ec, f[ic]);
    return 0;
}
