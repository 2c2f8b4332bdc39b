// Is the direct GEMM's K loop bound by HOW its loads are spread (16 rows x 64 B per wave instruction) rather than by bytes?
// Same number of 16-byte loads and MFMAs per wave; pattern 0 = the direct kernel's (lane (i, kq): row i, 16 B at k-quad kq),
// pattern 1 = fully contiguous (lane l: 16 B at 16 * l).  Timed alone and beside conv2 forward.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "bbbp_hip.h"
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

template <int PATTERN>
__global__ __launch_bounds__(256) void probe(const float* A, const float* B, float* C, int K, int nchunks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane & 15, kq = lane >> 4;
    const int tile = blockIdx.x * 4 + wave;
    const float *pa, *pb; long step;
    if (PATTERN == 0) { pa = A + (long)((tile * 16 + q) % 512) * K + 4 * kq; pb = B + (long)((tile * 7 + q) % 496) * K + 4 * kq; step = 16; }
    else { pa = A + (long)(tile % 32) * 16 * K + 4 * lane; pb = B + (long)(tile % 31) * 16 * K + 4 * lane; step = 256; }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x4 ra[4], rb[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) { ra[d] = *reinterpret_cast<const f32x4u*>(pa + step * d); rb[d] = *reinterpret_cast<const f32x4u*>(pb + step * d); }
    for (int c0 = 0; c0 < nchunks; c0 += 4) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[d][j], rb[d][j], acc, 0, 0, 0);
            const int c = min(c0 + d + 4, nchunks - 1);
            ra[d] = *reinterpret_cast<const f32x4u*>(pa + step * (PATTERN == 0 ? c : c % 10)); rb[d] = *reinterpret_cast<const f32x4u*>(pb + step * (PATTERN == 0 ? c : c % 10));
        }
    }
    if (acc[0] == 12345.f) C[tile] = acc[0] + acc[3];
}
__global__ void spin(int iters, float* out) { float v = threadIdx.x; for (int i = 0; i < iters; ++i) v = __builtin_fmaf(v, 1.0001f, 0.5f); if (v == 12345.f) out[0] = v; }

int main() {
    const int Bsz = 512, K = 167;
    float *x, *w, *bias, *y; uint8_t* mask; void* ws;
    size_t wsb = bbbp_conv3x3_workspace_bytes(Bsz, 32, 64, 64, 64);
    hipMalloc(&x, (size_t)Bsz * 32 * 64 * 64 * 4); hipMalloc(&w, 64 * 32 * 9 * 4); hipMalloc(&bias, 64 * 4);
    hipMalloc(&y, (size_t)Bsz * 64 * 32 * 32 * 4); hipMalloc(&mask, (size_t)Bsz * 64 * 32 * 32); hipMalloc(&ws, wsb ? wsb : 16);
    hipMemset(x, 0, (size_t)Bsz * 32 * 64 * 64 * 4); hipMemset(w, 0, 64 * 32 * 9 * 4); hipMemset(bias, 0, 64 * 4);
    float *A, *Bm, *C, *out; hipMalloc(&A, 4 << 20); hipMalloc(&Bm, 4 << 20); hipMalloc(&C, 1 << 20); hipMalloc(&out, 64);
    hipMemset(A, 0, 4 << 20); hipMemset(Bm, 0, 4 << 20);
    hipStream_t sa, sb; hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int N = 60, grid = 256;      // 1024 waves, like the QKV GEMM
    for (int pattern = 0; pattern < 2; ++pattern) for (int bg = 0; bg < 2; ++bg) {
        hipDeviceSynchronize();
        if (bg) for (int r = 0; r < 6; ++r) bbbp_conv3x3_relu_pool_fwd(sa, x, w, bias, y, mask, Bsz, 32, 64, 64, 64, ws, wsb);
        if (bg) for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, sb, 3000, out);
        hipEventRecord(e0, sb);
        for (int r = 0; r < N; ++r) {
            if (pattern == 0) hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(256), 0, sb, A, Bm, C, K, 10);
            else hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(256), 0, sb, A, Bm, C, K, 10);
        }
        hipEventRecord(e1, sb); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("pattern %d (%s) %s: %.2f us per launch\n", pattern, pattern ? "contiguous 1 KB per instruction" : "16 rows x 64 B per instruction", bg ? "beside conv2 fwd" : "alone", ms * 1000 / N);
    }
    return 0;
}
