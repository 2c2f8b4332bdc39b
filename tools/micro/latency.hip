// Micro-benchmark: per-instruction latencies seen by ONE wave per SIMD (the situation of the small encoder GEMMs).
// Build: hipcc -O3 --offload-arch=gfx950 -o build_ab/latency tools/micro/latency.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Out { unsigned long long clk, real; float sink; };

// mode 0: dependent LDS reads; 1: dependent MFMA chain; 2: dependent global loads (pointer chase, L2 resident);
// 3: barriers; 4: independent MFMAs (2 accumulators); 5: dependent VALU fma chain
__global__ __launch_bounds__(256) void lat_kernel(int mode, int iters, const int* chase, Out* out) {
    __shared__ int lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (i * 33 + 7) & 4095;
    __syncthreads();
    unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    float sink = 0.f;
    if (mode == 0) {
        int idx = threadIdx.x;
        for (int i = 0; i < iters; ++i) idx = lds[idx];
        sink = (float)idx;
    } else if (mode == 1) {
        f32x16 acc = {0};
        float a = threadIdx.x * 1e-3f, b = 1e-3f;
        for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        sink = acc[0] + acc[7];
    } else if (mode == 2) {
        int idx = (blockIdx.x * 256 + threadIdx.x) & 65535;
        for (int i = 0; i < iters; ++i) idx = chase[idx];
        sink = (float)idx;
    } else if (mode == 3) {
        for (int i = 0; i < iters; ++i) __syncthreads();
    } else if (mode == 4) {
        f32x16 acc = {0}, acc2 = {0};
        float a = threadIdx.x * 1e-3f, b = 1e-3f;
        for (int i = 0; i < iters; i += 2) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc2, 0, 0, 0);
        }
        sink = acc[0] + acc2[7];
    } else if (mode == 5) {
        float v = threadIdx.x;
        for (int i = 0; i < iters; ++i) v = __builtin_fmaf(v, 1.0001f, 0.5f);
        sink = v;
    }
    unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    if (threadIdx.x == 0) { out[blockIdx.x].clk = c1 - c0; out[blockIdx.x].real = r1 - r0; out[blockIdx.x].sink = sink; }
    else if (sink == 12345.678f) out[blockIdx.x].sink = sink;
}

int main() {
    const int N = 65536;
    std::vector<int> h(N);
    for (int i = 0; i < N; ++i) h[i] = (int)(((long)i * 4097 + 12345) & (N - 1));
    int* chase; Out* out;
    CHECK(hipMalloc(&chase, N * 4)); CHECK(hipMemcpy(chase, h.data(), N * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&out, 4096 * sizeof(Out)));
    const char* names[] = {"dependent ds_read", "dependent mfma 32x32x2", "dependent global load (L2)", "s_barrier (4 waves)", "2 independent mfma chains", "dependent v_fma"};
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int grid : {24, 256, 1024}) {
        for (int mode = 0; mode < 6; ++mode) {
            int iters = 512;
            for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(lat_kernel, dim3(grid), dim3(256), 0, 0, mode, iters, chase, out);
            CHECK(hipEventRecord(e0));
            for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL(lat_kernel, dim3(grid), dim3(256), 0, 0, mode, iters, chase, out);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            Out o; CHECK(hipMemcpy(&o, out, sizeof(Out), hipMemcpyDeviceToHost));
            printf("grid %4d  %-28s: %7.1f clk/iter  %7.2f ns/iter (wall_clock 100MHz)  -> %.2f GHz ; kernel %.2f us\n", grid, names[mode],
                   (double)o.clk / iters, (double)o.real * 10.0 / iters, (double)o.clk / ((double)o.real * 10.0), ms * 1000 / 20);
        }
    }
    // empty-kernel launch rate
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(lat_kernel, dim3(24), dim3(256), 0, 0, 3, 0, chase, out);
    CHECK(hipEventRecord(e0));
    for (int rep = 0; rep < 200; ++rep) hipLaunchKernelGGL(lat_kernel, dim3(24), dim3(256), 0, 0, 3, 0, chase, out);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("empty kernel (24 WGs, LDS init only): %.2f us per launch back-to-back\n", ms * 1000 / 200);
    return 0;
}
