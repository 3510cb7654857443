// Does v_mfma_f32_32x32x16_f16 honour f16 SUBNORMAL inputs (A and B side), and how exact is its accumulation?
// hipcc --offload-arch=gfx950 -O3 -o bin/mfma_f16_denorm mfma_f16_denorm.hip && bin/mfma_f16_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(const float* a, const float* b, float* out) {
    // lane l: A[row l&31][k = 8(l>>5)+j] = a[8(l>>5)+j] (same for every row), B[k][col l&31] = b[k]
    const int lane = threadIdx.x;
    h8 A, B;
    for (int j = 0; j < 8; ++j) { A[j] = (_Float16)a[8 * (lane >> 5) + j]; B[j] = (_Float16)b[8 * (lane >> 5) + j]; }
    f16v acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc, 0, 0, 0);
    if (lane == 0) out[0] = acc[0];
}
int main() {
    float ha[16], hb[16], *da, *db, *dout, r;
    hipMalloc(&da, 64); hipMalloc(&db, 64); hipMalloc(&dout, 4);
    struct { const char* name; float a0, b0, a1, b1; } cases[] = {
        {"normal x normal        (0.5 * 0.25)", 0.5f, 0.25f, 0, 0},
        {"A subnormal (2^-20) x 1024", 9.5367431640625e-7f, 1024.0f, 0, 0},
        {"B subnormal (2^-20) x 1024", 1024.0f, 9.5367431640625e-7f, 0, 0},
        {"A min subnormal 2^-24 x 2^10", 5.9604644775390625e-8f, 1024.0f, 0, 0},
        {"both subnormal 2^-15 * 2^-15", 3.0517578125e-5f, 3.0517578125e-5f, 0, 0},
        {"cancellation: 1*1 + (-1)*1 + 2^-20*1 (k=0,1 in slot 0,1; tiny in slot 8)", 1.0f, 1.0f, 0, 0},
    };
    for (auto& c : cases) {
        for (int i = 0; i < 16; ++i) ha[i] = hb[i] = 0;
        ha[0] = c.a0; hb[0] = c.b0;
        if (c.name[0] == 'c') { ha[0] = 1; hb[0] = 1; ha[1] = -1; hb[1] = 1; ha[8] = 9.5367431640625e-7f; hb[8] = 1; }
        hipMemcpy(da, ha, 64, hipMemcpyHostToDevice); hipMemcpy(db, hb, 64, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dout);
        hipMemcpy(&r, dout, 4, hipMemcpyDeviceToHost);
        double expect = 0; for (int i = 0; i < 16; ++i) expect += (double)(float)(_Float16)ha[i] * (double)(float)(_Float16)hb[i];
        printf("%-75s got %.9g expect %.9g\n", c.name, r, expect);
    }
    // accumulation exactness: 16 products of magnitude ~1 with low-order bits, compare with f64 sum rounded once
    double worst = 0;
    for (int t = 0; t < 200; ++t) {
        srand(t);
        for (int i = 0; i < 16; ++i) { ha[i] = (float)(_Float16)((rand() % 2001 - 1000) / 997.0f); hb[i] = (float)(_Float16)((rand() % 2001 - 1000) / 613.0f); }
        hipMemcpy(da, ha, 64, hipMemcpyHostToDevice); hipMemcpy(db, hb, 64, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dout);
        hipMemcpy(&r, dout, 4, hipMemcpyDeviceToHost);
        double e = 0, mag = 0; for (int i = 0; i < 16; ++i) { e += (double)ha[i] * hb[i]; mag += fabs((double)ha[i] * hb[i]); }
        double err = fabs(r - e) / mag; if (err > worst) worst = err;
    }
    printf("accumulation of 16 exact products: worst |err| / sum|terms| = %.3g (2^-24 = 5.96e-8)\n", worst);
    return 0;
}
