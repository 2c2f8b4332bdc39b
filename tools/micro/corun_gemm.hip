// What do the encoder's actual launches cost beside the persistent conv kernel?  Stream A: conv2 forward back to back;
// stream B: a chain of one library call repeated (GEMM shapes of the F=167 encoder, LayerNorm, softmax).
// Build: hipcc -O3 --offload-arch=gfx950 -Iinclude -o build_ab/corun_gemm tools/micro/corun_gemm.hip -L<pkg> -lbbbp_hip -Wl,-rpath,...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <functional>
#include "bbbp_hip.h"

__global__ void spin(int iters, float* out) { float v = threadIdx.x; for (int i = 0; i < iters; ++i) v = __builtin_fmaf(v, 1.0001f, 0.5f); if (v == 12345.f) out[0] = v; }

int main() {
    const int B = 512, F = 167, DFF = 2048;
    float *x, *w, *bias, *y; uint8_t* mask; void* ws;
    size_t wsb = bbbp_conv3x3_workspace_bytes(B, 32, 64, 64, 64);
    hipMalloc(&x, (size_t)B * 32 * 64 * 64 * 4); hipMalloc(&w, 64 * 32 * 9 * 4); hipMalloc(&bias, 64 * 4);
    hipMalloc(&y, (size_t)B * 64 * 32 * 32 * 4); hipMalloc(&mask, (size_t)B * 64 * 32 * 32); hipMalloc(&ws, wsb ? wsb : 16);
    hipMemset(x, 0, (size_t)B * 32 * 64 * 64 * 4); hipMemset(w, 0, 64 * 32 * 9 * 4); hipMemset(bias, 0, 64 * 4);
    float *a, *b, *c, *g, *mean, *rstd, *out; void* gws;
    hipMalloc(&a, 8 << 20); hipMalloc(&b, 8 << 20); hipMalloc(&c, 8 << 20); hipMalloc(&g, 1 << 20); hipMalloc(&mean, 4096); hipMalloc(&rstd, 4096); hipMalloc(&out, 64);
    hipMalloc(&gws, 64 << 20);
    hipMemset(a, 0, 8 << 20); hipMemset(b, 0, 8 << 20); hipMemset(g, 0, 1 << 20);
    hipStream_t sa, sb; hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    hipEvent_t e0, e1, c0, c1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&c0); hipEventCreate(&c1);
    struct Case { const char* name; std::function<int()> call; };
    Case cases[] = {
        {"QKV  NT 512x501x167", [&] { return bbbp_gemm_f32(sb, 0, 1, B, 3 * F, F, 1.f, a, F, b, F, c, 3 * F, g, nullptr, 0, 0, 1, 0, 0, 0, 0, gws, 64 << 20); }},
        {"QK^T NT 512x512x167", [&] { return bbbp_gemm_f32(sb, 0, 1, B, B, F, 1.f, a, 3 * F, a + F, 3 * F, c, B, nullptr, nullptr, 0, 0, 1, 0, 0, 0, 0, gws, 64 << 20); }},
        {"PV   NN 512x167x512", [&] { return bbbp_gemm_f32(sb, 0, 0, B, F, B, 1.f, a, B, b, 3 * F, c, F, nullptr, nullptr, 0, 0, 1, 0, 0, 0, 0, gws, 64 << 20); }},
        {"out  NT 512x167x167", [&] { return bbbp_gemm_f32(sb, 0, 1, B, F, F, 1.f, a, F, b, F, c, F, g, nullptr, 0, 0, 1, 0, 0, 0, 0, gws, 64 << 20); }},
        {"FFN1 NT 512x2048x167", [&] { return bbbp_gemm_f32(sb, 0, 1, B, DFF, F, 1.f, a, F, b, F, c, DFF, g, nullptr, 0, 1, 1, 0, 0, 0, 0, gws, 64 << 20); }},
        {"FFN2 NT 512x167x2048", [&] { return bbbp_gemm_f32(sb, 0, 1, B, F, DFF, 1.f, a, DFF, b, DFF, c, F, g, nullptr, 0, 0, 1, 0, 0, 0, 0, gws, 64 << 20); }},
        {"layernorm 512x167 (dropout 0.1)", [&] { return bbbp_layernorm_fwd(sb, a, b, c, g, g, mean, rstd, B, F, 1e-5f, 0.1f, 1234ull); }},
        {"softmax 512x512 (dropout 0.1)", [&] { return bbbp_softmax_fwd(sb, a, c, (long)B, B, 0.1f, 99ull); }},
        {"dropout 512x2048", [&] { return bbbp_dropout(sb, a, c, (long)B * DFF, 0.1f, 7ull); }},
    };
    const int N = 60;
    for (auto& cs : cases) {
        float t[2];
        for (int bg = 0; bg < 2; ++bg) {
            hipDeviceSynchronize();
            if (bg) for (int r = 0; r < 6; ++r) bbbp_conv3x3_relu_pool_fwd(sa, x, w, bias, y, mask, B, 32, 64, 64, 64, ws, wsb);
            if (bg) for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, sb, 3000, out);     // let the conv get going
            for (int r = 0; r < 3; ++r) if (cs.call()) { printf("call failed: %s\n", bbbp_last_error()); return 1; }
            hipEventRecord(e0, sb);
            for (int r = 0; r < N; ++r) cs.call();
            hipEventRecord(e1, sb);
            hipDeviceSynchronize();
            hipEventElapsedTime(&t[bg], e0, e1);
        }
        printf("%-34s alone %6.2f us/call   beside conv2 fwd %6.2f us/call\n", cs.name, t[0] * 1000 / N, t[1] * 1000 / N);
    }
    return 0;
}
