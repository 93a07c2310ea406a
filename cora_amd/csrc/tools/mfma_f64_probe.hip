// Probe of the gfx950 FP64 pipes: layout check + throughput of v_mfma_f64_16x16x4_f64
// and v_fma_f64, alone and interleaved.  Diagnostic tool, not part of the library.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void layout_kernel(const double* A, const double* B, double* C) {
    // A[16][4], B[4][16] row-major; C[16][16]
    int lane = threadIdx.x;
    double a = A[(lane & 15) * 4 + (lane >> 4)];
    double b = B[(lane >> 4) * 16 + (lane & 15)];
    d4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; r++) C[((lane >> 4) + 4 * r) * 16 + (lane & 15)] = acc[r];
}

template <int NACC, int NVALU>
__global__ void __launch_bounds__(256) rate_kernel(double* out, int iters, double x, unsigned long long* stamps) {
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    d4 acc[NACC > 0 ? NACC : 1];
    for (int i = 0; i < NACC; i++) acc[i] = (d4){0, 0, 0, 0};
    double a = threadIdx.x * 1e-3 + 1.0, b = 1.0 - threadIdx.x * 1e-4;
    double v[8];
    for (int i = 0; i < 8; i++) v[i] = x + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) {
            if (NACC > 0) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NVALU; j++) v[j & 7] = fma(v[j & 7], x, b);
        }
        if (NACC == 0) {
#pragma unroll
            for (int j = 0; j < NVALU; j++) v[j & 7] = fma(v[j & 7], x, b);
        }
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) { stamps[0] = c1 - c0; stamps[1] = r1 - r0; }
    double s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int NVALU>
int run(const char* name, int blocks, int threads, int iters) {
    double* out;
    CK(hipMalloc(&out, sizeof(double) * blocks * threads));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned long long* st; CK(hipMalloc(&st, 16)); unsigned long long hst[2];
    rate_kernel<NACC, NVALU><<<blocks, threads>>>(out, iters, 0.999, st);
    CK(hipDeviceSynchronize());
    hipEventRecord(e0);
    rate_kernel<NACC, NVALU><<<blocks, threads>>>(out, iters, 0.999, st);
    hipEventRecord(e1);
    CK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, e0, e1);
    CK(hipMemcpy(hst, st, 16, hipMemcpyDeviceToHost));
    double ghz = (double)hst[0] / ((double)hst[1] * 10.0);  // realtime ticks at 100 MHz
    double waves = (double)blocks * threads / 64;
    double nm = NACC > 0 ? (double)NACC : 1.0;
    double mfma_flops = (NACC > 0) ? waves * iters * NACC * 2048.0 : 0;
    double valu_flops = waves * iters * nm * NVALU * 128.0;
    printf("%-28s blocks=%d thr=%d: %.3f ms  MFMA %.2f TF  VALU %.2f TF  (cyc/MFMA/SIMD %.1f, clk %.2f GHz, cyc/DFMA/SIMD %.2f)\n", name, blocks, threads, ms,
           mfma_flops / ms / 1e9, valu_flops / ms / 1e9,
           NACC > 0 ? (double)hst[0] / (iters * NACC * (waves / (256.0 * 4))) : 0.0, ghz,
           NVALU > 0 ? (double)hst[0] / (iters * nm * NVALU * (waves / (256.0 * 4))) : 0.0);
    hipFree(out);
    return 0;
}

int main() {
    // ---- layout check with asymmetric integer data
    std::vector<double> A(64), B(64), C(256), R(256, 0.0);
    for (int i = 0; i < 16; i++) for (int k = 0; k < 4; k++) A[i * 4 + k] = i * 7 + k * 3 + 1;
    for (int k = 0; k < 4; k++) for (int j = 0; j < 16; j++) B[k * 16 + j] = (k + 1) * 100 + j * j;
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) for (int k = 0; k < 4; k++) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    double *dA, *dB, *dC;
    CK(hipMalloc(&dA, 64 * 8)); CK(hipMalloc(&dB, 64 * 8)); CK(hipMalloc(&dC, 256 * 8));
    CK(hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice));
    layout_kernel<<<1, 64>>>(dA, dB, dC);
    CK(hipMemcpy(C.data(), dC, 256 * 8, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 256; i++) if (C[i] != R[i]) bad++;
    printf("layout check: %d mismatches of 256\n", bad);

    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("device %s CUs=%d clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    int it = 200000;
    run<4, 0>("mfma x4acc, 1 wave/SIMD", 256, 256, it);
    run<4, 0>("mfma x4acc, 2 wave/SIMD", 512, 256, it);
    run<8, 0>("mfma x8acc, 1 wave/SIMD", 256, 256, it);
    run<4, 0>("mfma x4acc, 4 wave/SIMD", 1024, 256, it);
    run<2, 0>("mfma x2acc, 1 wave/SIMD", 256, 256, it);
    run<1, 0>("mfma x1acc (dep chain)", 256, 256, it);
    run<0, 16>("valu dfma only 1w", 256, 256, it);
    run<0, 16>("valu dfma only 2w", 512, 256, it);
    run<0, 16>("valu dfma only 4w", 1024, 256, it);
    run<4, 2>("mfma x4 + 2 dfma each", 256, 256, it);
    run<4, 4>("mfma x4 + 4 dfma each", 256, 256, it);
    run<4, 8>("mfma x4 + 8 dfma each", 256, 256, it);
    run<4, 16>("mfma x4 + 16 dfma each", 256, 256, it);
    run<4, 4>("mfma x4 + 4 dfma, 2w/SIMD", 512, 256, it);
    run<4, 8>("mfma x4 + 8 dfma, 2w/SIMD", 512, 256, it);
    return 0;
}
