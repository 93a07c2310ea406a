// Issue cost (cycles per wave-instruction and SIMD) of the VALU instructions the Philox / Box-Muller chain of K3 is
// made of, alone (1 and 2 waves per SIMD) and next to FP64 MFMAs.  Diagnostic tool, not part of the library:
//   make -C cora_amd/csrc valuprobe && cora_amd/csrc/tools/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// KIND: which instruction; 8 independent chains so that latency is hidden even with one wave per SIMD
enum { K_MAD64 = 0, K_XOR, K_BITOP3, K_FMA64, K_CVT_U32, K_LDEXP, K_RSQ, K_MOV64, K_CNDMASK, K_MULHI, K_MULLO, K_ADD64, K_FREXP, K_RNDNE, K_ALIGNBIT, K_PL32SWAP, K_PL16SWAP, K_FMA64_CHAIN, K_N };
static const char *names[K_N] = {"v_mad_u64_u32", "v_xor_b32", "v_bitop3_b32", "v_fma_f64", "v_cvt_f64_u32", "v_ldexp_f64", "v_rsq_f64",
                                 "v_mov_b64", "v_cndmask_b32", "v_mul_hi_u32", "v_mul_lo_u32", "v_add_f64", "v_frexp_mant_f64", "v_rndne_f64", "v_alignbit_b32", "v_permlane32_swap", "v_permlane16_swap", "v_fma_f64 (one dependent chain)"};

template <int KIND, int NMFMA>
__global__ void __launch_bounds__(256) probe(double *out, int iters, unsigned long long *stamps) {
    unsigned long long a[8];
    double d[8];
    unsigned u[8];
    for (int i = 0; i < 8; i++) {
        a[i] = threadIdx.x * 0x9E3779B97F4A7C15ull + i;
        d[i] = 1.0 + 1e-3 * threadIdx.x + i;
        u[i] = threadIdx.x * 2654435761u + i;
    }
    d4 acc[NMFMA > 0 ? NMFMA : 1];
    for (int i = 0; i < NMFMA; i++) acc[i] = (d4){0, 0, 0, 0};
    const double ma = 1.0 + threadIdx.x * 1e-3, mb = 1.0 - threadIdx.x * 1e-4;
    const unsigned m0 = 0xD2511F53u;
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NMFMA; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, acc[i], 0, 0, 0);
#pragma unroll
        for (int rep = 0; rep < 4; rep++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (KIND == K_MAD64) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(a[i]) : "v"(u[i]), "v"(m0) : "vcc");
                if (KIND == K_XOR) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(u[i]) : "v"(u[i]), "v"(m0));
                if (KIND == K_BITOP3) asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(u[i]) : "v"(u[i]), "v"(m0), "v"(u[(i + 1) & 7]));
                if (KIND == K_FMA64) asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(d[i]) : "v"(d[i]), "v"(ma), "v"(mb));
                if (KIND == K_CVT_U32) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[i]) : "v"(u[i]));
                if (KIND == K_LDEXP) asm volatile("v_ldexp_f64 %0, %1, 3" : "=v"(d[i]) : "v"(d[i]));
                if (KIND == K_RSQ) asm volatile("v_rsq_f64 %0, %1" : "=v"(d[i]) : "v"(d[i]));
                if (KIND == K_MOV64) asm volatile("v_mov_b64 %0, %1" : "=v"(d[i]) : "v"(d[(i + 1) & 7]));
                if (KIND == K_CNDMASK) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(u[i]) : "v"(u[i]), "v"(m0) : "vcc");
                if (KIND == K_MULHI) asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(u[i]) : "v"(u[i]), "v"(m0));
                if (KIND == K_MULLO) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(u[i]) : "v"(u[i]), "v"(m0));
                if (KIND == K_ADD64) asm volatile("v_add_f64 %0, %1, %2" : "=v"(d[i]) : "v"(d[i]), "v"(ma));
                if (KIND == K_FREXP) asm volatile("v_frexp_mant_f64 %0, %1" : "=v"(d[i]) : "v"(d[i]));
                if (KIND == K_RNDNE) asm volatile("v_rndne_f64 %0, %1" : "=v"(d[i]) : "v"(d[i]));
                if (KIND == K_ALIGNBIT) asm volatile("v_alignbit_b32 %0, %1, %2, 12" : "=v"(u[i]) : "v"(u[i]), "v"(m0));
                if (KIND == K_PL32SWAP) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u[i]), "+v"(u[(i + 1) & 7]));
                if (KIND == K_PL16SWAP) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(u[i]), "+v"(u[(i + 1) & 7]));
                if (KIND == K_FMA64_CHAIN) asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(d[0]) : "v"(d[0]), "v"(ma), "v"(mb));
            }
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) stamps[0] = c1 - c0;
    double s = 0;
    for (int i = 0; i < 8; i++) s += d[i] + (double)u[i] + (double)a[i];
    for (int i = 0; i < NMFMA; i++) s += acc[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int NMFMA>
static int run(int wps) {
    double *out;
    unsigned long long *st, h = 0;
    const int blocks = 256 * wps, iters = 2000;
    CK(hipMalloc(&out, sizeof(double) * blocks * 256));
    CK(hipMalloc(&st, 8));
    probe<KIND, NMFMA><<<blocks, 256>>>(out, iters, st);
    CK(hipDeviceSynchronize());
    probe<KIND, NMFMA><<<blocks, 256>>>(out, iters, st);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(&h, st, 8, hipMemcpyDeviceToHost));
    // s_memtime ticks at 100 MHz on gfx950?  report the raw count per iteration and per instruction as well
    const double per_iter = (double)h / iters;
    printf("%-18s mfma/iter %d  waves/SIMD %d : %8.2f ticks/iter  -> %6.3f ticks per instr per wave (32 instr/iter)%s\n", names[KIND], NMFMA, wps,
           per_iter, (per_iter - 0.0) / 32.0, NMFMA ? "  [incl. MFMAs]" : "");
    hipFree(out);
    hipFree(st);
    return 0;
}

template <int KIND>
static int all() {
    if (run<KIND, 0>(1)) return 1;
    if (run<KIND, 0>(2)) return 1;
    if (run<KIND, 4>(1)) return 1;
    if (run<KIND, 4>(2)) return 1;
    return 0;
}

int main() {
    // calibration: 4 MFMAs per iteration alone = 256 SIMD cycles per iteration at one wave per SIMD
    if (run<K_XOR, 4>(1)) return 1;
    all<K_MAD64>();
    all<K_MULHI>();
    all<K_MULLO>();
    all<K_XOR>();
    all<K_BITOP3>();
    all<K_ALIGNBIT>();
    all<K_CNDMASK>();
    all<K_MOV64>();
    all<K_FMA64>();
    all<K_ADD64>();
    all<K_CVT_U32>();
    all<K_LDEXP>();
    all<K_FREXP>();
    all<K_RNDNE>();
    all<K_RSQ>();
    all<K_PL32SWAP>();
    all<K_PL16SWAP>();
    all<K_FMA64_CHAIN>();
    return 0;
}
