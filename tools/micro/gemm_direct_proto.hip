// Feasibility prototype: latency-oriented f32 GEMM without LDS (operands fetched straight into MFMA 16x16x4 layout).
// NT only: C[M][N] = A[M][K] * B[N][K]^T.  Build: hipcc -O3 --offload-arch=gfx950 -o build_ab/gemm_direct tools/micro/gemm_direct_proto.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

struct P { const float *A, *B; float* C; int M, N, K, lda, ldb, ldc; };

template <int TM, int TN, int D>
__global__ __launch_bounds__(256) void gemm_direct_nt(P p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane & 15, kq = lane >> 4;
    const int m_base = blockIdx.y * (32 * TM) + (wave >> 1) * 16 * TM;
    const int n_base = blockIdx.x * (32 * TN) + (wave & 1) * 16 * TN;
    const float* pa[TM];
    const float* pb[TN];
#pragma unroll
    for (int u = 0; u < TM; ++u) pa[u] = p.A + (long)min(m_base + 16 * u + q, p.M - 1) * p.lda + 4 * kq;
#pragma unroll
    for (int u = 0; u < TN; ++u) pb[u] = p.B + (long)min(n_base + 16 * u + q, p.N - 1) * p.ldb + 4 * kq;
    const int nfull = p.K / 16;
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 ra[D][TM], rb[D][TN];
    auto fetch = [&](int d, int c) __attribute__((always_inline)) {
        const int cc = min(c, max(nfull - 1, 0)) * 16;
#pragma unroll
        for (int u = 0; u < TM; ++u) ra[d][u] = *reinterpret_cast<const f32x4u*>(pa[u] + cc);
#pragma unroll
        for (int u = 0; u < TN; ++u) rb[d][u] = *reinterpret_cast<const f32x4u*>(pb[u] + cc);
    };
    if (nfull > 0) {
#pragma unroll
        for (int d = 0; d < D; ++d) fetch(d, d);
        for (int c0 = 0; c0 < nfull; c0 += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                if (c0 + d < nfull) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int um = 0; um < TM; ++um)
#pragma unroll
                            for (int un = 0; un < TN; ++un)
                                acc[um][un] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[d][um][j], rb[d][un][j], acc[um][un], 0, 0, 0);
                }
                fetch(d, c0 + d + D);
            }
        }
    }
    // K tail, element-wise with clamped addresses
    const int kt = nfull * 16;
    if (kt < p.K) {
        float ta[TM][4], tb[TN][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = kt + 4 * kq + j;
            const bool ok = k < p.K;
            const int off = ok ? kt + j : 0;        // pa already carries + 4*kq
#pragma unroll
            for (int u = 0; u < TM; ++u) { float v = pa[u][ok ? off : -4 * kq]; ta[u][j] = ok ? v : 0.f; }
#pragma unroll
            for (int u = 0; u < TN; ++u) { float v = pb[u][ok ? off : -4 * kq]; tb[u][j] = ok ? v : 0.f; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int um = 0; um < TM; ++um)
#pragma unroll
                for (int un = 0; un < TN; ++un) acc[um][un] = __builtin_amdgcn_mfma_f32_16x16x4f32(ta[um][j], tb[un][j], acc[um][un], 0, 0, 0);
    }
#pragma unroll
    for (int um = 0; um < TM; ++um)
#pragma unroll
        for (int un = 0; un < TN; ++un) {
            const int n = n_base + 16 * un + q;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m_base + 16 * um + 4 * kq + r;
                if (m < p.M && n < p.N) p.C[(long)m * p.ldc + n] = acc[um][un][r];
            }
        }
}

template <int TM, int TN, int D>
float run(const P& p, int reps) {
    dim3 grid((p.N + 32 * TN - 1) / (32 * TN), (p.M + 32 * TM - 1) / (32 * TM));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((gemm_direct_nt<TM, TN, D>), grid, dim3(256), 0, 0, p);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((gemm_direct_nt<TM, TN, D>), grid, dim3(256), 0, 0, p);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000 / reps;
}

int main() {
    struct Case { int M, N, K; } cases[] = {{512, 167, 167}, {512, 501, 167}, {512, 512, 167}, {512, 2048, 167}, {512, 167, 2048}, {512, 167, 512}};
    for (auto& c : cases) {
        std::vector<float> hA((size_t)c.M * c.K), hB((size_t)c.N * c.K), hC((size_t)c.M * c.N);
        srand(1);
        for (auto& v : hA) v = (rand() % 2001 - 1000) * 1e-3f;
        for (auto& v : hB) v = (rand() % 2001 - 1000) * 1e-3f;
        float *A, *B, *C;
        hipMalloc(&A, hA.size() * 4); hipMalloc(&B, hB.size() * 4); hipMalloc(&C, hC.size() * 4);     // exact sizes: overruns would fault
        hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice); hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
        P p{A, B, C, c.M, c.N, c.K, c.K, c.K, c.N};
        float t11 = run<1, 1, 8>(p, 50);
        hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
        double err = 0;
        for (int m = 0; m < c.M; m += 37) for (int n = 0; n < c.N; ++n) {
            double s = 0; for (int k = 0; k < c.K; ++k) s += (double)hA[(size_t)m * c.K + k] * hB[(size_t)n * c.K + k];
            err = fmax(err, fabs(s - hC[(size_t)m * c.N + n]));
        }
        float t12 = run<1, 2, 4>(p, 50), t22 = run<2, 2, 4>(p, 50), t21 = run<2, 1, 4>(p, 50), t11d4 = run<1, 1, 4>(p, 50), t22d2 = run<2, 2, 2>(p, 50);
        hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
        double err2 = 0;
        for (int m = 0; m < c.M; m += 37) for (int n = 0; n < c.N; ++n) {
            double s = 0; for (int k = 0; k < c.K; ++k) s += (double)hA[(size_t)m * c.K + k] * hB[(size_t)n * c.K + k];
            err2 = fmax(err2, fabs(s - hC[(size_t)m * c.N + n]));
        }
        printf("M=%d N=%d K=%d: 1x1(D8) %.2f us, 1x1(D4) %.2f, 1x2 %.2f, 2x1 %.2f, 2x2 %.2f, 2x2(D2) %.2f   max err %.2e / %.2e\n", c.M, c.N, c.K, t11, t11d4, t12, t21, t22, t22d2, err, err2);
        hipFree(A); hipFree(B); hipFree(C);
    }
    return 0;
}
