// Micro-benchmark: what does a tiny kernel chain cost when it shares the GPU with the persistent conv kernel?
// Stream A runs conv2 forward (B=512, 32->64, 64x64) back to back; stream B runs a chain of small kernels, each
// exercising one resource.  Compare per-kernel time alone vs beside the conv.
// Build: hipcc -O3 --offload-arch=gfx950 -Iinclude -o build_ab/corun tools/micro/corun.hip -L<pkg> -lbbbp_hip -Wl,-rpath,<pkg>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "bbbp_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool PRIO>
__global__ __launch_bounds__(256) void small_kernel(int mode, int iters, const int* chase, float* out) {
    if (PRIO) __builtin_amdgcn_s_setprio(3);
    __shared__ int lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = (i * 33 + 7) & 1023;
    __syncthreads();
    float sink = 0.f;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    if (mode == 0) { int idx = threadIdx.x; for (int i = 0; i < iters; ++i) idx = lds[idx]; sink = (float)idx; }
    else if (mode == 1) { f32x16 acc = {0}; float a = threadIdx.x * 1e-3f; for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, 1e-3f, acc, 0, 0, 0); sink = acc[0] + acc[7]; }
    else if (mode == 2) { int idx = (blockIdx.x * 256 + threadIdx.x) & 65535; for (int i = 0; i < iters; ++i) idx = chase[idx]; sink = (float)idx; }
    else if (mode == 3) { float v = threadIdx.x; for (int i = 0; i < iters; ++i) v = __builtin_fmaf(v, 1.0001f, 0.5f); sink = v; }
    else if (mode == 4) { f32x4 acc = {0}; float a = threadIdx.x * 1e-3f; for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, 1e-3f, acc, 0, 0, 0); sink = acc[0] + acc[3]; }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[2] = (float)(c1 - c0); out[3] = (float)(r1 - r0); }
    if (sink == 12345.678f) out[0] = sink;
}

int main() {
    const int B = 512;
    float *x, *w, *bias, *y; uint8_t* mask; void* ws;
    size_t wsb = bbbp_conv3x3_workspace_bytes(B, 32, 64, 64, 64);
    hipMalloc(&x, (size_t)B * 32 * 64 * 64 * 4); hipMalloc(&w, 64 * 32 * 9 * 4); hipMalloc(&bias, 64 * 4);
    hipMalloc(&y, (size_t)B * 64 * 32 * 32 * 4); hipMalloc(&mask, (size_t)B * 64 * 32 * 32); hipMalloc(&ws, wsb ? wsb : 16);
    hipMemset(x, 0, (size_t)B * 32 * 64 * 64 * 4); hipMemset(w, 0, 64 * 32 * 9 * 4); hipMemset(bias, 0, 64 * 4);
    int* chase; float* out;
    std::vector<int> h(65536);
    for (int i = 0; i < 65536; ++i) h[i] = (int)(((long)i * 4097 + 12345) & 65535);
    hipMalloc(&chase, 65536 * 4); hipMemcpy(chase, h.data(), 65536 * 4, hipMemcpyHostToDevice); hipMalloc(&out, 64);
    hipStream_t sa, sb; hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    hipEvent_t e0, e1, c0, c1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&c0); hipEventCreate(&c1);
    const char* names[] = {"ds_read chain", "mfma 32x32x2 chain", "L2 load chain", "v_fma chain", "mfma 16x16x4 chain"};
    const int NCHAIN = 100;
    // conv alone
    for (int r = 0; r < 2; ++r) bbbp_conv3x3_relu_pool_fwd(sa, x, w, bias, y, mask, B, 32, 64, 64, 64, ws, wsb);
    hipEventRecord(c0, sa);
    for (int r = 0; r < 4; ++r) bbbp_conv3x3_relu_pool_fwd(sa, x, w, bias, y, mask, B, 32, 64, 64, 64, ws, wsb);
    hipEventRecord(c1, sa); hipEventSynchronize(c1);
    float cms; hipEventElapsedTime(&cms, c0, c1);
    printf("conv2 fwd alone: %.1f us per launch\n", cms * 1000 / 4);
    for (int grid : {96}) for (int prio = 0; prio < 2; ++prio) for (int mode = 0; mode < 5; ++mode) for (int iters : {0, 256}) {
        if (iters == 0 && mode != 0) continue;
        float t[2], convt = 0, cyc[2], ghz[2];
        for (int bg = 0; bg < 2; ++bg) {
            hipDeviceSynchronize();
            if (bg) { hipEventRecord(c0, sa); for (int r = 0; r < 6; ++r) bbbp_conv3x3_relu_pool_fwd(sa, x, w, bias, y, mask, B, 32, 64, 64, 64, ws, wsb); hipEventRecord(c1, sa); }
            // let the conv get going before the chain starts
            if (bg) { for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(small_kernel<false>, dim3(1), dim3(256), 0, sb, 3, 2000, chase, out); }
            hipEventRecord(e0, sb);
            for (int r = 0; r < NCHAIN; ++r) {
                if (prio) hipLaunchKernelGGL(small_kernel<true>, dim3(grid), dim3(256), 0, sb, mode, iters, chase, out);
                else hipLaunchKernelGGL(small_kernel<false>, dim3(grid), dim3(256), 0, sb, mode, iters, chase, out);
            }
            hipEventRecord(e1, sb);
            hipDeviceSynchronize();
            hipEventElapsedTime(&t[bg], e0, e1);
            float hc[4]; hipMemcpy(hc, out, 16, hipMemcpyDeviceToHost);
            cyc[bg] = hc[2]; ghz[bg] = hc[3] > 0 ? hc[2] / (hc[3] * 10.f) : 0.f;
            if (bg) hipEventElapsedTime(&convt, c0, c1);
        }
        printf("grid %3d prio %d %-20s iters %3d: alone %6.2f us/kernel, beside conv %6.2f us/kernel (6 convs took %.0f us); body %.0f cyc @ %.2f GHz alone, %.0f cyc @ %.2f GHz beside\n", grid, prio,
               iters ? names[mode] : "empty", iters, t[0] * 1000 / NCHAIN, t[1] * 1000 / NCHAIN, convt * 1000, cyc[0], ghz[0], cyc[1], ghz[1]);
    }
    return 0;
}
