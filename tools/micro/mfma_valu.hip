// Micro-benchmark: does an f32 MFMA stream block the VALU of the same SIMD?  A work-group of 8 waves = 2 per SIMD:
// waves 0-3 issue independent v_mfma_f32_32x32x2_f32 back to back, waves 4-7 issue independent v_fma_f32.
// Each kind is timed alone and together (cycle counter).  If the two pipes were independent the times would not add.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// what: bit 0 = MFMA waves active, bit 1 = VALU waves active
__global__ __launch_bounds__(512) void k(int what, int iters, unsigned long long* out, float* sink) {
    const int wave = threadIdx.x >> 6;
    const bool is_mfma = wave < 4;
    unsigned long long c0 = 0, c1 = 0;
    float s = 0.f;
    __syncthreads();
    if (is_mfma && (what & 1)) {
        f32x16 a0 = {0}, a1 = {0};
        float x = threadIdx.x * 1e-3f;
        c0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, 1e-3f, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(1e-3f, x, a1, 0, 0, 0);
            }
        }
        c1 = __builtin_readcyclecounter();
        s = a0[0] + a1[5];
    } else if (!is_mfma && (what & 2)) {
        float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f, v4 = 4.f, v5 = 5.f, v6 = 6.f, v7 = 7.f;
        c0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                v0 = __builtin_fmaf(v0, 1.0001f, 0.5f); v1 = __builtin_fmaf(v1, 1.0001f, 0.5f);
                v2 = __builtin_fmaf(v2, 1.0001f, 0.5f); v3 = __builtin_fmaf(v3, 1.0001f, 0.5f);
                v4 = __builtin_fmaf(v4, 1.0001f, 0.5f); v5 = __builtin_fmaf(v5, 1.0001f, 0.5f);
                v6 = __builtin_fmaf(v6, 1.0001f, 0.5f); v7 = __builtin_fmaf(v7, 1.0001f, 0.5f);
            }
        }
        c1 = __builtin_readcyclecounter();
        s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    }
    else if (!is_mfma && (what & 4)) {          // dependent v_fma chain
        float v = threadIdx.x;
        c0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 64; ++u) v = __builtin_fmaf(v, 1.0001f, 0.5f);
        }
        c1 = __builtin_readcyclecounter();
        s = v;
    } else if (!is_mfma && (what & 8)) {          // dependent SALU chain (uniform integer ops)
        int a = iters | 1;
        c0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 64; ++u) a = a * 3 + (a >> 3);
        }
        c1 = __builtin_readcyclecounter();
        s = (float)a;
    }
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[wave] = c1 - c0;
    if (s == 12345.678f) sink[0] = s;
}

int main() {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 64); hipMalloc(&sink, 64);
    const int iters = 200;
    const char* names[] = {"", "MFMA waves only", "VALU waves only", "both", "dep VALU only", "dep VALU + MFMA", "", "", "dep SALU only", "dep SALU + MFMA"};
    for (int what : {1, 2, 3, 4, 5, 8, 9}) {
        hipMemset(out, 0, 64);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, what, iters, out, sink);
        hipDeviceSynchronize();
        unsigned long long h[8]; hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
        printf("%-16s: MFMA wave %.1f cyc per mfma (16/iter), VALU wave %.2f cyc per v_fma (64/iter)\n", names[what],
               (double)h[0] / (iters * 16.0), (double)h[4] / (iters * 64.0));
        if (what >= 4) printf("      (per op in the 64-op chain: %.2f cycles; SALU chain has 3 ops per step)\n", (double)h[4] / (iters * 64.0));
    }
    return 0;
}
