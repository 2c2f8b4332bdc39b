// HBM-bound row / column kernels of the BBBP hot path for gfx950: LayerNorm (+residual, +dropout),
// row softmax (+dropout), BatchNorm1d, bias/activation backward, the attention-fusion combine,
// dropout, MSE and AdamW.  They replace the ATen elementwise / reduction ops behind
// nn.LayerNorm, softmax, nn.BatchNorm1d, nn.Dropout, nn.MSELoss and optim.AdamW on the path
// (a3, a9, a10, a11, a13).  One wave (64 lanes) owns a row and reduces with DPP
// shuffles; column statistics use 64-column x 16-row-lane blocks with a fixed-order LDS tree, so every
// result is bit-reproducible run to run (no float atomics).
#include "common.h"
#include "bbbp_hip.h"
#include <stdlib.h>
#include <mutex>

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over the last dim.  z = dropout(x) + r is written back over x (saved for backward);
// y = (z - mean) * rstd * gamma + beta.  One wave per row.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(float* x, const float* r, float* y, const float* gamma,
                                                           const float* beta, float* mean_out, float* rstd_out,
                                                           int rows, int cols, float eps, float p, uint64_t seed_in, const unsigned long long* seed_base) {
    const uint64_t seed = effective_seed(seed_in, seed_base);
    BBBP_HIGH_PRIO();
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* xr = x + (long)row * cols;
    const float* rr = r ? r + (long)row * cols : nullptr;
    const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) {
        float v = xr[c];
        if (p > 0.f) v *= dropout_scale(seed, (uint64_t)row * cols + c, p, inv_keep);
        if (rr) v += rr[c];
        xr[c] = v;
        s += v;
    }
    const float mean = wave_sum(s) / cols;
    float q = 0.f;
    for (int c = lane; c < cols; c += 64) { float d = xr[c] - mean; q += d * d; }
    const float rstd = rsqrtf(wave_sum(q) / cols + eps);
    float* yr = y + (long)row * cols;
    for (int c = lane; c < cols; c += 64) yr[c] = (xr[c] - mean) * rstd * gamma[c] + beta[c];
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

// dz = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma.  dz is the gradient of the
// residual input; dx (optional, when dropout was applied to x) = dz * keep-scale.
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* dy, const float* z, const float* gamma,
                                                           const float* mean, const float* rstd, float* dz, float* dx,
                                                           int rows, int cols, float p, uint64_t seed_in, const unsigned long long* seed_base) {
    const uint64_t seed = effective_seed(seed_in, seed_base);
    BBBP_HIGH_PRIO();
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* dyr = dy + (long)row * cols;
    const float* zr = z + (long)row * cols;
    const float mu = mean[row], rs = rstd[row];
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < cols; c += 64) {
        float g = dyr[c] * gamma[c];
        s1 += g;
        s2 += g * (zr[c] - mu) * rs;
    }
    s1 = wave_sum(s1) / cols;
    s2 = wave_sum(s2) / cols;
    const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
    for (int c = lane; c < cols; c += 64) {
        float g = dyr[c] * gamma[c];
        float v = rs * (g - s1 - (zr[c] - mu) * rs * s2);
        dz[(long)row * cols + c] = v;
        if (dx) dx[(long)row * cols + c] = p > 0.f ? v * dropout_scale(seed, (uint64_t)row * cols + c, p, inv_keep) : v;
    }
}

// Wide rows (the F = 2048 encoder: 8 KB per row).  One wave per row left 512 waves on 1024 SIMDs walking 2048 columns three times
// with a full Philox block per ELEMENT: 39 / 45 us per call for 16 MB of traffic (0.3 TB/s, profiles/r02_kernel_stats_config4.csv).
// Here a 256-thread work-group owns a row and keeps it in registers (NV float4 per thread, cols <= 1024 * NV): every byte is
// read once with 16-byte loads, one Philox block serves the four elements of a float4 (the same stream as dropout_scale: element
// idx draws component idx & 3 of block idx >> 2), the two row statistics are two block reductions in a fixed order.
__device__ __forceinline__ float block_sum4(float v, float* red) {     // 4 waves; every thread gets the sum, same order everywhere
    v = wave_sum(v);
    __syncthreads();                           // the previous use of `red` has been read
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((red[0] + red[1]) + red[2]) + red[3];
}
__device__ __forceinline__ float4 dropout_scale4(uint64_t seed, uint64_t idx4, float p, float inv_keep) {
    const uint4 r = philox4(seed, idx4);
    const float k = 1.0f / 16777216.0f;
    return make_float4((float)(r.x >> 8) * k >= p ? inv_keep : 0.f, (float)(r.y >> 8) * k >= p ? inv_keep : 0.f,
                       (float)(r.z >> 8) * k >= p ? inv_keep : 0.f, (float)(r.w >> 8) * k >= p ? inv_keep : 0.f);
}
template <int NV>
__global__ __launch_bounds__(256) void layernorm_fwd_wide_kernel(float* x, const float* r, float* y, const float* gamma,
                                                                const float* beta, float* mean_out, float* rstd_out,
                                                                int rows, int cols, float eps, float p, uint64_t seed_in, const unsigned long long* seed_base) {
    const uint64_t seed = effective_seed(seed_in, seed_base);
    BBBP_HIGH_PRIO();
    __shared__ float red[4];
    const int t = threadIdx.x, nq = cols >> 2;
    const long row = blockIdx.x;
    float4* xr = reinterpret_cast<float4*>(x + row * cols);
    const float4* rr = r ? reinterpret_cast<const float4*>(r + row * cols) : nullptr;
    const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int q = t + i * 256;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < nq) {
            v[i] = xr[q];
            if (p > 0.f) {
                const float4 k = dropout_scale4(seed, ((uint64_t)row * cols >> 2) + q, p, inv_keep);
                v[i].x *= k.x; v[i].y *= k.y; v[i].z *= k.z; v[i].w *= k.w;
            }
            if (rr) { const float4 a = rr[q]; v[i].x += a.x; v[i].y += a.y; v[i].z += a.z; v[i].w += a.w; }
            xr[q] = v[i];
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    const float mean = block_sum4(s, red) / cols;
    float qd = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (t + i * 256 < nq) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            qd += (a * a + b * b) + (c * c + d * d);
        }
    const float rstd = rsqrtf(block_sum4(qd, red) / cols + eps);
    float4* yr = reinterpret_cast<float4*>(y + row * cols);
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    const float4* b4 = reinterpret_cast<const float4*>(beta);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int q = t + i * 256;
        if (q < nq) {
            const float4 g = g4[q], b = b4[q];
            yr[q] = make_float4((v[i].x - mean) * rstd * g.x + b.x, (v[i].y - mean) * rstd * g.y + b.y,
                                (v[i].z - mean) * rstd * g.z + b.z, (v[i].w - mean) * rstd * g.w + b.w);
        }
    }
    if (t == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}
template <int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_wide_kernel(const float* dy, const float* z, const float* gamma,
                                                                const float* mean, const float* rstd, float* dz, float* dx,
                                                                int rows, int cols, float p, uint64_t seed_in, const unsigned long long* seed_base) {
    const uint64_t seed = effective_seed(seed_in, seed_base);
    BBBP_HIGH_PRIO();
    __shared__ float red[4];
    const int t = threadIdx.x, nq = cols >> 2;
    const long row = blockIdx.x;
    const float4* dyr = reinterpret_cast<const float4*>(dy + row * cols);
    const float4* zr = reinterpret_cast<const float4*>(z + row * cols);
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    const float mu = mean[row], rs = rstd[row];
    float4 g[NV], xh[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int q = t + i * 256;
        g[i] = xh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < nq) {
            const float4 d = dyr[q], w = g4[q], zz = zr[q];
            g[i] = make_float4(d.x * w.x, d.y * w.y, d.z * w.z, d.w * w.w);
            xh[i] = make_float4((zz.x - mu) * rs, (zz.y - mu) * rs, (zz.z - mu) * rs, (zz.w - mu) * rs);
            s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
            s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
        }
    }
    s1 = block_sum4(s1, red) / cols;
    s2 = block_sum4(s2, red) / cols;
    const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
    float4* dzr = reinterpret_cast<float4*>(dz + row * cols);
    float4* dxr = dx ? reinterpret_cast<float4*>(dx + row * cols) : nullptr;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int q = t + i * 256;
        if (q < nq) {
            const float4 v = make_float4(rs * (g[i].x - s1 - xh[i].x * s2), rs * (g[i].y - s1 - xh[i].y * s2),
                                         rs * (g[i].z - s1 - xh[i].z * s2), rs * (g[i].w - s1 - xh[i].w * s2));
            dzr[q] = v;
            if (dxr) {
                if (p > 0.f) {
                    const float4 k = dropout_scale4(seed, ((uint64_t)row * cols >> 2) + q, p, inv_keep);
                    dxr[q] = make_float4(v.x * k.x, v.y * k.y, v.z * k.z, v.w * k.w);
                } else {
                    dxr[q] = v;
                }
            }
        }
    }
}
// the wide form serves rows of 1024..4096 columns in whole float4s (gfx950 global loads need 4-byte alignment only)
inline int ln_wide_nv(int cols) {
    static const int on = [] { const char* e = getenv("BBBP_LN_WIDE"); return e ? atoi(e) : 1; }();
    return (on && cols >= 1024 && cols <= 4096 && cols % 4 == 0) ? (cols + 1023) / 1024 : 0;
}

// column sums over rows of (a) dy * xhat and (b) dy:  LayerNorm dgamma / dbeta.
// Column reductions use 16-column x 64-row-lane blocks: enough blocks to spread a 167-wide tensor over the chip.
// 512 threads = 2 waves per SIMD x <= 32 VGPRs: small enough to co-reside with a 224-VGPR x 2 conv weight-gradient
// work-group (64 registers per lane are free per SIMD); 1024-thread blocks could not be placed while it runs.
constexpr int CW = 16, RL = 32;
__global__ __launch_bounds__(512) void layernorm_param_grad_kernel(const float* dy, const float* z, const float* mean,
                                                                   const float* rstd, float* dgamma, float* dbeta,
                                                                   int rows, int cols) {
    BBBP_HIGH_PRIO();
    __shared__ float s1[RL][CW], s2[RL][CW];
    const int cl = threadIdx.x % CW, rl = threadIdx.x / CW;
    const int c = blockIdx.x * CW + cl;
    float a = 0.f, b = 0.f;
    if (c < cols)
        for (int r = rl; r < rows; r += RL) {
            float g = dy[(long)r * cols + c];
            a += g * (z[(long)r * cols + c] - mean[r]) * rstd[r];
            b += g;
        }
    s1[rl][cl] = a; s2[rl][cl] = b;
    __syncthreads();
    if (rl == 0 && c < cols) {
        float ta = 0.f, tb = 0.f;
#pragma unroll 8
        for (int i = 0; i < RL; ++i) { ta += s1[i][cl]; tb += s2[i][cl]; }
        dgamma[c] = ta; dbeta[c] = tb;
    }
}

// ---------------------------------------------------------------------------------------------
// softmax over the last dim, one wave per row, in place; optional dropout copy pd = dropout(p).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_fwd_kernel(float* x, float* pd, long rows, int cols, float p, uint64_t seed_in, const unsigned long long* seed_base) {
    const uint64_t seed = effective_seed(seed_in, seed_base);
    BBBP_HIGH_PRIO();
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* xr = x + row * cols;
    float m = -INFINITY;
    for (int c = lane; c < cols; c += 64) m = fmaxf(m, xr[c]);
    m = wave_max(m);
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) { float e = __expf(xr[c] - m); xr[c] = e; s += e; }
    const float inv = 1.f / wave_sum(s);
    const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
    for (int c = lane; c < cols; c += 64) {
        float v = xr[c] * inv;
        xr[c] = v;
        if (pd) pd[row * cols + c] = v * dropout_scale(seed, (uint64_t)row * cols + c, p, inv_keep);
    }
}

// ds = P * (dP - sum(dP * P)), dP = dpd * keep-scale.  In place over dpd.
__global__ __launch_bounds__(256) void softmax_bwd_kernel(float* dpd, const float* prob, long rows, int cols, float p, uint64_t seed_in, const unsigned long long* seed_base) {
    const uint64_t seed = effective_seed(seed_in, seed_base);
    BBBP_HIGH_PRIO();
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* dr = dpd + row * cols;
    const float* pr = prob + row * cols;
    const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) {
        float d = dr[c];
        if (p > 0.f) { d *= dropout_scale(seed, (uint64_t)row * cols + c, p, inv_keep); dr[c] = d; }
        s += d * pr[c];
    }
    s = wave_sum(s);
    for (int c = lane; c < cols; c += 64) dr[c] = pr[c] * (dr[c] - s);
}

// y = x * keep-scale (forward and backward of nn.Dropout share this kernel and the seed)
__global__ __launch_bounds__(256) void dropout_kernel(const float* x, float* y, long n, float p, uint64_t seed_in, const unsigned long long* seed_base) {
    const uint64_t seed = effective_seed(seed_in, seed_base);
    BBBP_HIGH_PRIO();
    const float inv_keep = 1.f / (1.f - p);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        y[i] = x[i] * dropout_scale(seed, (uint64_t)i, p, inv_keep);
}

// ---------------------------------------------------------------------------------------------
// BatchNorm1d over [rows, cols]; a block owns 64 columns and all rows.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void batchnorm_fwd_kernel(const float* x, float* y, const float* gamma, const float* beta,
                                                            float* running_mean, float* running_var, float* save_mean,
                                                            float* save_rstd, int rows, int cols, float eps, float momentum,
                                                            int training) {
    BBBP_HIGH_PRIO();
    __shared__ float red[16][64];
    __shared__ float smean[64], srstd[64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    if (training) {
        float a = 0.f;
        if (c < cols) for (int r = rl; r < rows; r += 16) a += x[(long)r * cols + c];
        red[rl][cl] = a;
        __syncthreads();
        if (rl == 0) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) t += red[i][cl];
            smean[cl] = t / rows;
        }
        __syncthreads();
        const float mu = smean[cl];
        float q = 0.f;
        if (c < cols) for (int r = rl; r < rows; r += 16) { float d = x[(long)r * cols + c] - mu; q += d * d; }
        __syncthreads();
        red[rl][cl] = q;
        __syncthreads();
        if (rl == 0) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) t += red[i][cl];
            float var = t / rows;
            srstd[cl] = rsqrtf(var + eps);
            if (c < cols) {
                save_mean[c] = mu; save_rstd[c] = srstd[cl];
                running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * ((float)rows / (float)(rows - 1));
            }
        }
        __syncthreads();
    } else {
        if (rl == 0 && c < cols) {
            smean[cl] = running_mean[c];
            srstd[cl] = rsqrtf(running_var[c] + eps);
            save_mean[c] = smean[cl]; save_rstd[c] = srstd[cl];
        }
        __syncthreads();
    }
    if (c < cols) {
        const float mu = smean[cl], rs = srstd[cl], g = gamma[c], b = beta[c];
        for (int r = rl; r < rows; r += 16) y[(long)r * cols + c] = (x[(long)r * cols + c] - mu) * rs * g + b;
    }
}

__global__ __launch_bounds__(1024) void batchnorm_bwd_kernel(const float* dy, const float* x, const float* gamma,
                                                            const float* save_mean, const float* save_rstd, float* dx,
                                                            float* dgamma, float* dbeta, int rows, int cols, int training,
                                                            int relu_gate) {
    BBBP_HIGH_PRIO();
    __shared__ float r1[16][64], r2[16][64];
    __shared__ float t1[64], t2[64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const float mu = c < cols ? save_mean[c] : 0.f, rs = c < cols ? save_rstd[c] : 0.f;
    float a = 0.f, b = 0.f;
    if (c < cols)
        for (int r = rl; r < rows; r += 16) {
            float g = dy[(long)r * cols + c];
            a += g * (x[(long)r * cols + c] - mu) * rs;
            b += g;
        }
    r1[rl][cl] = a; r2[rl][cl] = b;
    __syncthreads();
    if (rl == 0) {
        float ta = 0.f, tb = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { ta += r1[i][cl]; tb += r2[i][cl]; }
        t1[cl] = ta; t2[cl] = tb;
        if (c < cols) { dgamma[c] = ta; dbeta[c] = tb; }
    }
    __syncthreads();
    if (c < cols) {
        const float g = gamma[c], sa = t1[cl] / rows, sb = t2[cl] / rows;
        for (int r = rl; r < rows; r += 16) {
            float d = dy[(long)r * cols + c];
            const float xv = x[(long)r * cols + c];
            float v = training ? g * rs * (d - sb - (xv - mu) * rs * sa) : d * g * rs;
            if (relu_gate) v = xv > 0.f ? v : 0.f;       // x is the output of a ReLU: its backward rides along
            dx[(long)r * cols + c] = v;
        }
    }
}

// Pieces of a BatchNorm1d whose batch is sharded over ranks (exact-global-batch mode, distributed.py): local column
// moments about a given centre, and the input gradient from GLOBAL sums.  The collectives in between run on the host side.
__global__ __launch_bounds__(512) void column_moments_kernel(const float* x, const float* centre, float* s1, float* s2,
                                                            int rows, int cols) {
    BBBP_HIGH_PRIO();
    __shared__ float a1[32][16], a2[32][16];
    const int cl = threadIdx.x % 16, rl = threadIdx.x / 16;
    const int c = blockIdx.x * 16 + cl;
    const float m = (centre && c < cols) ? centre[c] : 0.f;
    float a = 0.f, b = 0.f;
    if (c < cols)
        for (int r = rl; r < rows; r += 32) { float d = x[(long)r * cols + c] - m; a += d; b += d * d; }
    a1[rl][cl] = a; a2[rl][cl] = b;
    __syncthreads();
    if (rl == 0 && c < cols) {
        float ta = 0.f, tb = 0.f;
#pragma unroll 8
        for (int i = 0; i < 32; ++i) { ta += a1[i][cl]; tb += a2[i][cl]; }
        s1[c] = ta; s2[c] = tb;
    }
}

// dx = gamma * rstd * (dy - sum_dy / n - xhat * sum_dy_xhat / n) with sums over the GLOBAL batch of n rows
__global__ __launch_bounds__(256) void batchnorm_bwd_apply_kernel(const float* dy, const float* x, const float* gamma,
                                                                 const float* mean, const float* rstd, const float* sum_dy,
                                                                 const float* sum_dy_xhat, float* dx, long total, int cols,
                                                                 float inv_n) {
    BBBP_HIGH_PRIO();
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int c = (int)(i % cols);
        float xh = (x[i] - mean[c]) * rstd[c];
        dx[i] = gamma[c] * rstd[c] * (dy[i] - sum_dy[c] * inv_n - xh * sum_dy_xhat[c] * inv_n);
    }
}

// ---------------------------------------------------------------------------------------------
// dy <- dy * act'(y) in place (y = the activation's OUTPUT), db[n] = column sums of the result.
// act: 0 none, 1 relu (y > 0), 2 tanh (1 - y^2).  Leading dims allow column slices.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void bias_act_bwd_kernel(float* dy, int lddy, const float* y, int ldy, float* db,
                                                           int rows, int cols, int act, float scale) {
    BBBP_HIGH_PRIO();
    __shared__ float red[RL][CW];
    const int cl = threadIdx.x % CW, rl = threadIdx.x / CW;
    const int c = blockIdx.x * CW + cl;
    float a = 0.f;
    if (c < cols)
        for (int r = rl; r < rows; r += RL) {
            float g = dy[(long)r * lddy + c];
            if (act == 1) g = y[(long)r * ldy + c] > 0.f ? g * scale : 0.f;
            else if (act == 2) { float t = y[(long)r * ldy + c]; g *= (1.f - t * t) * scale; }
            if (act) dy[(long)r * lddy + c] = g;
            a += g;
        }
    red[rl][cl] = a;
    __syncthreads();
    if (rl == 0 && c < cols && db) {
        float t = 0.f;
#pragma unroll 8
        for (int i = 0; i < RL; ++i) t += red[i][cl];
        db[c] = t;
    }
}

// ---------------------------------------------------------------------------------------------
// attention fusion combine (reference MultiHeadAttentionFusion.forward, ...20250113.py:60-65):
// logits_h = hid_h . w2_h + b2_h ; a = softmax_h(logits) ; out = sum_h a_h * combined.
// hid: [NH][rows][HD] (tanh outputs), w2: NH pointers.  One wave per row; NH <= 8.
// ---------------------------------------------------------------------------------------------
struct FusionPtrs { const float* w2[8]; const float* b2[8]; };

__global__ __launch_bounds__(256) void fusion_combine_fwd_kernel(const float* combined, const float* hid, FusionPtrs fp,
                                                                float* out, float* attn, int rows, int dim, int hd, int nh) {
    BBBP_HIGH_PRIO();
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float a[8];
    float m = -INFINITY;
    for (int h = 0; h < nh; ++h) {
        const float* hr = hid + ((long)h * rows + row) * hd;
        float s = 0.f;
        for (int c = lane; c < hd; c += 64) s += hr[c] * fp.w2[h][c];
        a[h] = wave_sum(s) + fp.b2[h][0];
        m = fmaxf(m, a[h]);
    }
    float den = 0.f;
    for (int h = 0; h < nh; ++h) { a[h] = __expf(a[h] - m); den += a[h]; }
    for (int h = 0; h < nh; ++h) { a[h] /= den; if (lane == 0) attn[(long)row * nh + h] = a[h]; }
    for (int c = lane; c < dim; c += 64) {
        float x = combined[(long)row * dim + c], s = 0.f;
        for (int h = 0; h < nh; ++h) s += a[h] * x;
        out[(long)row * dim + c] = s;
    }
}

// dcombined = dout * sum_h a_h ; dlogit_h = a_h * (t - sum_k a_k t), t = dout . combined ;
// dpre[h][row][:] = dlogit_h * w2_h * (1 - hid^2)   (through the Tanh)
__global__ __launch_bounds__(256) void fusion_combine_bwd_kernel(const float* dout, const float* combined, const float* hid,
                                                                const float* attn, FusionPtrs fp, float* dcombined,
                                                                float* dlogit, float* dpre, int rows, int dim, int hd, int nh) {
    BBBP_HIGH_PRIO();
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float a[8], asum = 0.f;
    for (int h = 0; h < nh; ++h) { a[h] = attn[(long)row * nh + h]; asum += a[h]; }
    float t = 0.f;
    for (int c = lane; c < dim; c += 64) {
        float g = dout[(long)row * dim + c];
        t += g * combined[(long)row * dim + c];
        dcombined[(long)row * dim + c] = g * asum;
    }
    t = wave_sum(t);
    // every head sees the same d(out)/d(a_h) . dout = t, so sum_k a_k t = t * asum
    for (int h = 0; h < nh; ++h) {
        float dl = a[h] * (t - t * asum);
        if (lane == 0) dlogit[(long)h * rows + row] = dl;
        const float* hr = hid + ((long)h * rows + row) * hd;
        float* dp = dpre + ((long)h * rows + row) * hd;
        for (int c = lane; c < hd; c += 64) { float y = hr[c]; dp[c] = dl * fp.w2[h][c] * (1.f - y * y); }
    }
}

// ---------------------------------------------------------------------------------------------
// MSE: loss = mean((pred - y)^2); dpred = 2 (pred - y) / n * gscale.  Single block, fixed order.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mse_kernel(const float* pred, const float* y, float* loss, float* dpred, int n, float gscale) {
    BBBP_HIGH_PRIO();
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        float d = pred[i] - y[i];
        s += d * d;
        if (dpred) dpred[i] = 2.f * d / n * gscale;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0 && loss) loss[0] = red[0] / n;
}

// ---------------------------------------------------------------------------------------------
// AdamW, torch.optim.AdamW arithmetic (decoupled decay, no amsgrad): one flat launch.
// ---------------------------------------------------------------------------------------------
// One element of the update.  Every rounding is spelled out (no contraction left to the compiler), for two reasons: all launch forms of the
// step -- flat, sliced, multi-tensor, vectorised or not -- produce the same bits for the same element, and the sequence is the one
// torch.optim.AdamW executes on the CPU in float32 (the reference's optimizer, R:172; probed op by op against torch 2.10's single-tensor
// path: exp_avg.lerp_ is one fused multiply-add, exp_avg_sq.mul_().addcmul_() rounds value * g, then fuses (value g) g + beta2 v,
// addcdiv_ rounds (-step_size m), divides, adds): tests/test_gpu_round4.py holds the kernel to <= 1 ulp on < 0.1 % of the elements
// against torch's own step.
struct AdamHyper { float decay, omb1, beta2, omb2, step_size, bc2_sqrt, eps, gscale; int low_prio; };

__device__ __forceinline__ void adamw_element(float& p, float g, float& m, float& v, const AdamHyper& h) {
    // (HIP's __fmul_rn / __fadd_rn are plain operators the compiler may still contract: the pragma is what pins the roundings)
#pragma clang fp contract(off)
    const float gi = g * h.gscale;
    const float pi = p * h.decay;                                                   // param.mul_(1 - lr * weight_decay)
    const float mi = __builtin_fmaf(h.omb1, gi - m, m);                             // exp_avg.lerp_(grad, 1 - beta1)
    const float t = h.omb2 * gi;
    const float vb = v * h.beta2;
    const float vi = __builtin_fmaf(t, gi, vb);                                     // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    const float sq = __builtin_sqrtf(vi) / h.bc2_sqrt;
    const float denom = sq + h.eps;                                                 // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
    const float num = -h.step_size * mi;
    const float q = num / denom;
    p = pi + q;                                                                     // param.addcdiv_(exp_avg, denom, value=-step_size)
    m = mi; v = vi;
}

// BBBP_ADAMW_BACKGROUND=n (experiment, default 0 = off): launches of the step use at most n work-groups and no raised wave priority -- an
// update that runs BESIDE compute (the optimizer pipelined into the backward pass) should trickle through HBM, not evict the GEMMs.
int adamw_background() {
    static const int v = [] { const char* e = getenv("BBBP_ADAMW_BACKGROUND"); return e ? atoi(e) : 0; }();
    return v;
}

// The float32 constants of a step, derived in DOUBLE from double hyper-parameters exactly as torch.optim.AdamW derives its Python floats
// (1 - lr * weight_decay, 1 - beta1, 1 - beta2, lr / (1 - beta1^t), (1 - beta2^t) ** 0.5) and rounded to float32 once, where torch's
// kernels round them: float hyper-parameters at this boundary (rounds 1-3) put 1 - 0.999f = 0.00099998713 where torch has 0.001f.
AdamHyper adam_hyper(double lr, double beta1, double beta2, double eps, double weight_decay, int step, double grad_scale) {
    const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
    return AdamHyper{(float)(1.0 - lr * weight_decay), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2),
                     (float)(lr / bc1), (float)pow(bc2, 0.5), (float)eps, (float)grad_scale, adamw_background() > 0 ? 1 : 0};
}

__global__ __launch_bounds__(256) void adamw_kernel(float* p, const float* g, float* m, float* v, long n, AdamHyper h) {
    if (!h.low_prio) BBBP_HIGH_PRIO();
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float pi = p[i], mi = m[i], vi = v[i];
        adamw_element(pi, g[i], mi, vi, h);
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}

// ---- multi-tensor form (round 4): parameters and moments are ONE flat buffer, the gradients are separate tensors (what autograd leaves behind when
// the model is a per-op composition: 180 tensors in the wide/deep variant = 180 launches and 2.7 ms of host time per step before this).  `offs`
// (nt + 1 element offsets into the flat buffer, offs[0] = 0, offs[nt] = n) and `grads` (nt device pointers) live in device memory; the hyper-
// parameters come by value or, when `hyper_dev` is set, from eight floats in device memory (a captured step replays with the values of the
// day).  Same adamw_element as adamw_kernel: bit-identical to nt single launches.
__global__ void adamw_hyper_store_kernel(float* out, AdamHyper h) {
    if (threadIdx.x == 0) { out[0] = h.decay; out[1] = h.omb1; out[2] = h.beta2; out[3] = h.omb2; out[4] = h.step_size; out[5] = h.bc2_sqrt; out[6] = h.eps; out[7] = h.gscale; }
}

__global__ __launch_bounds__(256) void adamw_multi_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, long n,
                                                         const long* __restrict__ offs, const float* const* __restrict__ grads, int nt,
                                                         AdamHyper h, const float* __restrict__ hyper_dev) {
    BBBP_HIGH_PRIO();
    if (hyper_dev) {
        h.decay = hyper_dev[0]; h.omb1 = hyper_dev[1]; h.beta2 = hyper_dev[2]; h.omb2 = hyper_dev[3];
        h.step_size = hyper_dev[4]; h.bc2_sqrt = hyper_dev[5]; h.eps = hyper_dev[6]; h.gscale = hyper_dev[7];
    }
    const long n4 = n & ~3L;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
        int lo = 0, hi = nt;                                   // the tensor holding element i: offs[lo] <= i < offs[lo + 1]
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (offs[mid] <= i) lo = mid; else hi = mid; }
        long base = offs[lo], end = offs[lo + 1];
        if (i < n4 && i + 4 <= end) {
            const float* gp = grads[lo] + (i - base);
            float4 pv = *reinterpret_cast<const float4*>(p + i), mv = *reinterpret_cast<const float4*>(m + i), vv = *reinterpret_cast<const float4*>(v + i);
            float g0, g1, g2, g3;
            if ((reinterpret_cast<uintptr_t>(gp) & 15) == 0) { const float4 gv = *reinterpret_cast<const float4*>(gp); g0 = gv.x; g1 = gv.y; g2 = gv.z; g3 = gv.w; }
            else { g0 = gp[0]; g1 = gp[1]; g2 = gp[2]; g3 = gp[3]; }
            adamw_element(pv.x, g0, mv.x, vv.x, h); adamw_element(pv.y, g1, mv.y, vv.y, h);
            adamw_element(pv.z, g2, mv.z, vv.z, h); adamw_element(pv.w, g3, mv.w, vv.w, h);
            *reinterpret_cast<float4*>(p + i) = pv; *reinterpret_cast<float4*>(m + i) = mv; *reinterpret_cast<float4*>(v + i) = vv;
        } else {
            for (long j = i; j < i + 4 && j < n; ++j) {
                while (j >= end) { ++lo; base = offs[lo]; end = offs[lo + 1]; }          // also steps over empty tensors
                float pj = p[j], mj = m[j], vj = v[j];
                adamw_element(pj, grads[lo][j - base], mj, vj, h);
                p[j] = pj; m[j] = mj; v[j] = vj;
            }
        }
    }
}

__global__ __launch_bounds__(256) void scale_kernel(float* x, long n, float s) {
    BBBP_HIGH_PRIO();
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) x[i] *= s;
}

inline int grid_for(long n, int background = 0) {
    long g = (n + 255) / 256;
    long cap = background > 0 ? background : (long)bbbp_num_cus() * 8;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

#define ST static_cast<hipStream_t>(stream)

extern "C" int bbbp_layernorm_fwd(void* stream, float* x_inout_z, const float* residual, float* y, const float* gamma,
                                  const float* beta, float* mean, float* rstd, int rows, int cols, float eps,
                                  float dropout_p, uint64_t seed) {
    BBBP_CHECK_ARG(rows >= 0 && cols > 0, "layernorm: bad shape %d x %d", rows, cols);
    BBBP_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "layernorm: bad dropout %f", dropout_p);
    if (rows == 0) return BBBP_OK;
#define BBBP_LN_FWD_WIDE(NV) hipLaunchKernelGGL(layernorm_fwd_wide_kernel<NV>, dim3(rows), dim3(256), g_bbbp_small_lds_pad, ST, x_inout_z, residual, \
                                                y, gamma, beta, mean, rstd, rows, cols, eps, dropout_p, seed, g_bbbp_seed_base)
    switch (ln_wide_nv(cols)) {
        case 1: BBBP_LN_FWD_WIDE(1); break;
        case 2: BBBP_LN_FWD_WIDE(2); break;
        case 3: BBBP_LN_FWD_WIDE(3); break;
        case 4: BBBP_LN_FWD_WIDE(4); break;
        default:
            hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), g_bbbp_small_lds_pad, ST, x_inout_z, residual, y, gamma, beta,
                               mean, rstd, rows, cols, eps, dropout_p, seed, g_bbbp_seed_base);
    }
#undef BBBP_LN_FWD_WIDE
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}

extern "C" int bbbp_layernorm_bwd(void* stream, const float* dy, const float* z, const float* gamma, const float* mean,
                                  const float* rstd, float* dz, float* dx, float* dgamma, float* dbeta, int rows, int cols,
                                  float dropout_p, uint64_t seed) {
    BBBP_CHECK_ARG(rows >= 0 && cols > 0, "layernorm bwd: bad shape %d x %d", rows, cols);
    // dz == NULL skips the input gradient, dgamma == NULL skips the parameter gradients (the engine runs the two
    // halves on different streams: the parameter gradients are off the critical dependency chain)
    if (rows > 0 && dz) {
#define BBBP_LN_BWD_WIDE(NV) hipLaunchKernelGGL(layernorm_bwd_wide_kernel<NV>, dim3(rows), dim3(256), g_bbbp_small_lds_pad, ST, dy, z, gamma, mean, \
                                                rstd, dz, dx, rows, cols, dropout_p, seed, g_bbbp_seed_base)
        switch (ln_wide_nv(cols)) {
            case 1: BBBP_LN_BWD_WIDE(1); break;
            case 2: BBBP_LN_BWD_WIDE(2); break;
            case 3: BBBP_LN_BWD_WIDE(3); break;
            case 4: BBBP_LN_BWD_WIDE(4); break;
            default:
                hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), g_bbbp_small_lds_pad, ST, dy, z, gamma, mean, rstd, dz, dx,
                                   rows, cols, dropout_p, seed, g_bbbp_seed_base);
        }
#undef BBBP_LN_BWD_WIDE
        BBBP_CHECK_LAUNCH();
    }
    if (dgamma) {
        hipLaunchKernelGGL(layernorm_param_grad_kernel, dim3(cdiv(cols, CW)), dim3(CW * RL), g_bbbp_small_lds_pad, ST, dy, z, mean, rstd, dgamma,
                           dbeta, rows, cols);
        BBBP_CHECK_LAUNCH();
    }
    return BBBP_OK;
}

extern "C" int bbbp_softmax_fwd(void* stream, float* x_inout, float* dropped_out, long rows, int cols, float dropout_p,
                                uint64_t seed) {
    BBBP_CHECK_ARG(rows >= 0 && cols > 0, "softmax: bad shape");
    if (rows == 0) return BBBP_OK;
    hipLaunchKernelGGL(softmax_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), g_bbbp_small_lds_pad, ST, x_inout,
                       dropout_p > 0.f ? dropped_out : nullptr, rows, cols, dropout_p, seed, g_bbbp_seed_base);
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}

extern "C" int bbbp_softmax_bwd(void* stream, float* dprob_inout, const float* prob, long rows, int cols, float dropout_p,
                                uint64_t seed) {
    BBBP_CHECK_ARG(rows >= 0 && cols > 0, "softmax bwd: bad shape");
    if (rows == 0) return BBBP_OK;
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), g_bbbp_small_lds_pad, ST, dprob_inout, prob, rows, cols,
                       dropout_p, seed, g_bbbp_seed_base);
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}

extern "C" int bbbp_dropout(void* stream, const float* x, float* y, long n, float p, uint64_t seed) {
    BBBP_CHECK_ARG(p >= 0.f && p < 1.f, "dropout: bad p %f", p);
    if (n == 0) return BBBP_OK;
    if (p == 0.f) {
        if (x != y) BBBP_CHECK_HIP(hipMemcpyAsync(y, x, n * sizeof(float), hipMemcpyDeviceToDevice, ST));
        return BBBP_OK;
    }
    hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(n)), dim3(256), g_bbbp_small_lds_pad, ST, x, y, n, p, seed, g_bbbp_seed_base);
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}

extern "C" int bbbp_batchnorm1d_fwd(void* stream, const float* x, float* y, const float* gamma, const float* beta,
                                    float* running_mean, float* running_var, float* save_mean, float* save_rstd, int rows,
                                    int cols, float eps, float momentum, int training) {
    BBBP_CHECK_ARG(cols > 0 && rows >= 0, "batchnorm: bad shape");
    // same failure mode as nn.BatchNorm1d on a single-row training batch (tiny-batch tails)
    BBBP_CHECK_ARG(!(training && rows <= 1), "Expected more than 1 value per channel when training, got input size [%d, %d]", rows, cols);
    if (rows == 0) return BBBP_OK;
    hipLaunchKernelGGL(batchnorm_fwd_kernel, dim3(cdiv(cols, 64)), dim3(1024), g_bbbp_small_lds_pad, ST, x, y, gamma, beta, running_mean,
                       running_var, save_mean, save_rstd, rows, cols, eps, momentum, training);
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}

extern "C" int bbbp_batchnorm1d_bwd(void* stream, const float* dy, const float* x, const float* gamma, const float* save_mean,
                                    const float* save_rstd, float* dx, float* dgamma, float* dbeta, int rows, int cols,
                                    int training) {
    BBBP_CHECK_ARG(cols > 0 && rows >= 0, "batchnorm bwd: bad shape");
    hipLaunchKernelGGL(batchnorm_bwd_kernel, dim3(cdiv(cols, 64)), dim3(1024), g_bbbp_small_lds_pad, ST, dy, x, gamma, save_mean, save_rstd, dx,
                       dgamma, dbeta, rows, cols, training, 0);
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}

// the same, followed by the backward of the ReLU that produced x (dx = 0 where x <= 0): nn.Sequential(..., ReLU, BatchNorm1d)
extern "C" int bbbp_batchnorm1d_bwd_relu(void* stream, const float* dy, const float* x, const float* gamma, const float* save_mean,
                                         const float* save_rstd, float* dx, float* dgamma, float* dbeta, int rows, int cols,
                                         int training) {
    BBBP_CHECK_ARG(cols > 0 && rows >= 0, "batchnorm bwd: bad shape");
    hipLaunchKernelGGL(batchnorm_bwd_kernel, dim3(cdiv(cols, 64)), dim3(1024), g_bbbp_small_lds_pad, ST, dy, x, gamma, save_mean, save_rstd, dx,
                       dgamma, dbeta, rows, cols, training, 1);
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}

extern "C" int bbbp_column_moments(void* stream, const float* x, const float* centre, float* sum_out, float* sumsq_out, int rows,
                                   int cols) {
    BBBP_CHECK_ARG(rows >= 0 && cols > 0 && x && sum_out && sumsq_out, "column_moments: bad arguments");
    hipLaunchKernelGGL(column_moments_kernel, dim3(cdiv(cols, 16)), dim3(512), g_bbbp_small_lds_pad, ST, x, centre, sum_out, sumsq_out,
                       rows, cols);
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}

extern "C" int bbbp_batchnorm1d_bwd_apply(void* stream, const float* dy, const float* x, const float* gamma, const float* mean,
                                          const float* rstd, const float* sum_dy, const float* sum_dy_xhat, float* dx, int rows,
                                          int cols, long n_global) {
    BBBP_CHECK_ARG(rows >= 0 && cols > 0 && n_global >= 1, "batchnorm bwd_apply: bad arguments");
    if (rows == 0) return BBBP_OK;
    long total = (long)rows * cols;
    hipLaunchKernelGGL(batchnorm_bwd_apply_kernel, dim3(grid_for(total)), dim3(256), g_bbbp_small_lds_pad, ST, dy, x, gamma, mean, rstd,
                       sum_dy, sum_dy_xhat, dx, total, cols, 1.0f / (float)n_global);
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}

extern "C" int bbbp_bias_act_bwd(void* stream, float* dy_inout, int lddy, const float* y, int ldy, float* dbias, int rows,
                                 int cols, int act, float scale) {
    BBBP_CHECK_ARG(cols > 0 && rows >= 0 && act >= 0 && act <= 2, "bias_act_bwd: bad args");
    BBBP_CHECK_ARG(act == 0 || y, "bias_act_bwd: activation output required");
    hipLaunchKernelGGL(bias_act_bwd_kernel, dim3(cdiv(cols, CW)), dim3(CW * RL), g_bbbp_small_lds_pad, ST, dy_inout, lddy, y, ldy, dbias, rows, cols,
                       act, scale);
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}

extern "C" int bbbp_fusion_combine_fwd(void* stream, const float* combined, const float* hid, const float* const* w2,
                                       const float* const* b2, float* out, float* attn, int rows, int dim, int hidden,
                                       int num_heads) {
    BBBP_CHECK_ARG(num_heads >= 1 && num_heads <= 8, "fusion: num_heads %d not in [1, 8]", num_heads);
    if (rows == 0) return BBBP_OK;
    FusionPtrs fp;
    for (int h = 0; h < 8; ++h) { fp.w2[h] = h < num_heads ? w2[h] : nullptr; fp.b2[h] = h < num_heads ? b2[h] : nullptr; }
    hipLaunchKernelGGL(fusion_combine_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), g_bbbp_small_lds_pad, ST, combined, hid, fp, out, attn, rows,
                       dim, hidden, num_heads);
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}

extern "C" int bbbp_fusion_combine_bwd(void* stream, const float* dout, const float* combined, const float* hid,
                                       const float* attn, const float* const* w2, float* dcombined, float* dlogit,
                                       float* dpre, int rows, int dim, int hidden, int num_heads) {
    BBBP_CHECK_ARG(num_heads >= 1 && num_heads <= 8, "fusion bwd: num_heads %d not in [1, 8]", num_heads);
    if (rows == 0) return BBBP_OK;
    FusionPtrs fp;
    for (int h = 0; h < 8; ++h) { fp.w2[h] = h < num_heads ? w2[h] : nullptr; fp.b2[h] = nullptr; }
    hipLaunchKernelGGL(fusion_combine_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), g_bbbp_small_lds_pad, ST, dout, combined, hid, attn, fp,
                       dcombined, dlogit, dpre, rows, dim, hidden, num_heads);
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}

extern "C" int bbbp_mse(void* stream, const float* pred, const float* target, float* loss, float* dpred, int n, float grad_scale) {
    BBBP_CHECK_ARG(n > 0, "mse: empty input");
    hipLaunchKernelGGL(mse_kernel, dim3(1), dim3(256), g_bbbp_small_lds_pad, ST, pred, target, loss, dpred, n, grad_scale);
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}

// ---- a slice of the update on a side stream (round 4: the serial stretch between two steps) -------------------------------------
// The optimizer step is HBM-bound (28 bytes per parameter: 0.065 ms at F = 167) and nothing of the model runs beside it.  62 % of its bytes
// are the image-FC weight, which the NEXT forward pass does not read before its third kernel (0.8 ms in).  bbbp_adamw_step_deferred updates
// everything else on the caller's stream and that slice on a library-owned stream, behind an event; bbbp_mixed_forward waits for the event
// right before the image FC (any other entry point that touches parameters waits at its start: bbbp_param_wait).  Same kernel, same
// element-wise arithmetic: the parameters are bit-identical to the one-launch step's.
namespace {
struct DeferredSlice { hipStream_t stream = nullptr; hipEvent_t ready = nullptr, grads = nullptr; bool pending = false; const char* lo = nullptr; const char* hi = nullptr; };
DeferredSlice g_deferred[64];
std::mutex g_deferred_mu;
}  // namespace

// make `st` wait for a pending deferred slice.  `ptr` != null: only when ptr lies inside the slice (the caller is about to read that tensor);
// null: unconditionally (the caller may read any parameter).  Returns 1 when a wait was enqueued, 0 when there was nothing to wait for.
int bbbp_param_wait(hipStream_t st, const void* ptr) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    std::lock_guard<std::mutex> lock(g_deferred_mu);
    DeferredSlice& d = g_deferred[dev];
    if (!d.pending) return 0;
    if (ptr && !(static_cast<const char*>(ptr) >= d.lo && static_cast<const char*>(ptr) < d.hi)) return 0;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &cap);
    if (cap != hipStreamCaptureStatusNone) { (void)hipEventSynchronize(d.ready); d.pending = false; return 0; }   // never wait for an outside event inside a capture
    (void)hipStreamWaitEvent(st, d.ready, 0);
    d.pending = false;
    return 1;
}
extern "C" int bbbp_param_sync(void* stream) { (void)bbbp_param_wait(static_cast<hipStream_t>(stream), nullptr); return BBBP_OK; }
// the side stream of the current device (null before the first deferred step): a caller whose allocator is stream-ordered must know that the
// gradient buffer is in use there (torch: tensor.record_stream)
extern "C" void* bbbp_param_stream(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(g_deferred_mu);
    return g_deferred[dev].stream;
}
// is a deferred slice pending that does NOT contain ptr?  (forward: then wait at the start instead of at the image FC)
bool bbbp_param_pending_elsewhere(const void* ptr) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    std::lock_guard<std::mutex> lock(g_deferred_mu);
    const DeferredSlice& d = g_deferred[dev];
    return d.pending && !(static_cast<const char*>(ptr) >= d.lo && static_cast<const char*>(ptr) < d.hi);
}

extern "C" int bbbp_adamw_step(void* stream, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n,
                               double lr, double beta1, double beta2, double eps, double weight_decay, int step, double grad_scale) {
    BBBP_CHECK_ARG(step >= 1, "adamw: step is 1-based, got %d", step);
    if (n == 0) return BBBP_OK;
    (void)bbbp_param_wait(static_cast<hipStream_t>(stream), nullptr);     // a deferred slice of an earlier step is ordered before this update
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n, adamw_background())), dim3(256), g_bbbp_small_lds_pad, ST, param, grad, exp_avg, exp_avg_sq, n,
                       adam_hyper(lr, beta1, beta2, eps, weight_decay, step, grad_scale));
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}

// the eight derived floats the kernels compute with (decay, 1-beta1, beta2, 1-beta2, lr / bias-correction-1, sqrt(bias-correction-2), eps,
// gradient scale) stored to device memory in stream order (`hyper_dev` below): the values travel as kernel arguments, so the host may call
// this again for the next step at once
extern "C" int bbbp_adamw_hyper_store(void* stream, float* hyper_dev, double lr, double beta1, double beta2, double eps, double weight_decay, int step,
                                      double grad_scale) {
    BBBP_CHECK_ARG(step >= 1 && hyper_dev, "adamw_hyper_store: step is 1-based (got %d), hyper_dev must not be null", step);
    hipLaunchKernelGGL(adamw_hyper_store_kernel, dim3(1), dim3(64), 0, ST, hyper_dev, adam_hyper(lr, beta1, beta2, eps, weight_decay, step, grad_scale));
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}

// One launch for a flat parameter / moment buffer whose gradients are `n_tensors` separate device tensors.  `table_dev` (device memory, owned
// by the caller, read on `stream`): long offsets[n_tensors + 1] followed by const float* grads[n_tensors].  `hyper_dev` (nullable): eight
// floats in device memory as bbbp_adamw_hyper_store writes them -- then lr ... grad_scale are ignored and `step` is not checked.
extern "C" int bbbp_adamw_step_multi(void* stream, float* param, float* exp_avg, float* exp_avg_sq, long n, const void* table_dev, int n_tensors,
                                     double lr, double beta1, double beta2, double eps, double weight_decay, int step, double grad_scale,
                                     const float* hyper_dev) {
    BBBP_CHECK_ARG(hyper_dev || step >= 1, "adamw_multi: step is 1-based, got %d", step);
    BBBP_CHECK_ARG(n >= 0 && n_tensors >= 1, "adamw_multi: n = %ld, n_tensors = %d", n, n_tensors);
    if (n == 0) return BBBP_OK;
    BBBP_CHECK_ARG(param && exp_avg && exp_avg_sq && table_dev, "adamw_multi: null pointer");
    BBBP_CHECK_ARG(((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(exp_avg) | reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(table_dev) & 7) == 0, "adamw_multi: flat buffers must be 16-byte aligned, the table 8-byte aligned");
    (void)bbbp_param_wait(static_cast<hipStream_t>(stream), nullptr);
    const long* offs = static_cast<const long*>(table_dev);
    const float* const* grads = reinterpret_cast<const float* const*>(offs + n_tensors + 1);
    const AdamHyper h = hyper_dev ? AdamHyper{} : adam_hyper(lr, beta1, beta2, eps, weight_decay, step, grad_scale);
    hipLaunchKernelGGL(adamw_multi_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), g_bbbp_small_lds_pad, ST, param, exp_avg, exp_avg_sq, n, offs, grads,
                       n_tensors, h, hyper_dev);
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}

extern "C" int bbbp_scale(void* stream, float* x, long n, float s) {
    if (n == 0) return BBBP_OK;
    hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n)), dim3(256), g_bbbp_small_lds_pad, ST, x, n, s);
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}

extern "C" int bbbp_adamw_step_deferred(void* stream, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, long lo, long hi,
                                        double lr, double beta1, double beta2, double eps, double weight_decay, int step, double grad_scale) {
    BBBP_CHECK_ARG(step >= 1, "adamw: step is 1-based, got %d", step);
    BBBP_CHECK_ARG(n >= 0 && lo >= 0 && lo <= hi && hi <= n, "adamw_deferred: slice [%ld, %ld) of %ld", lo, hi, n);
    BBBP_CHECK_ARG(n == 0 || (param && grad && exp_avg && exp_avg_sq), "adamw_deferred: null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    (void)bbbp_param_wait(st, nullptr);                       // an earlier deferred slice first
    int dev = 0;
    BBBP_CHECK_HIP(hipGetDevice(&dev));
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &cap);
    if (hi == lo || dev < 0 || dev >= 64 || cap != hipStreamCaptureStatusNone)
        return bbbp_adamw_step(stream, param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale);
    DeferredSlice& d = g_deferred[dev];
    {
        std::lock_guard<std::mutex> lock(g_deferred_mu);
        if (!d.stream) {
            BBBP_CHECK_HIP(hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking));
            BBBP_CHECK_HIP(hipEventCreateWithFlags(&d.ready, hipEventDisableTiming));
            BBBP_CHECK_HIP(hipEventCreateWithFlags(&d.grads, hipEventDisableTiming));
        }
    }
    const AdamHyper h = adam_hyper(lr, beta1, beta2, eps, weight_decay, step, grad_scale);
    auto launch = [&](hipStream_t s, long a, long b) {
        if (b > a)
            hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(b - a)), dim3(256), g_bbbp_small_lds_pad, s, param + a, grad + a, exp_avg + a, exp_avg_sq + a,
                               b - a, h);
    };
    // the gradients are final at this point of the caller's stream: the side stream starts there
    BBBP_CHECK_HIP(hipEventRecord(d.grads, st));
    BBBP_CHECK_HIP(hipStreamWaitEvent(d.stream, d.grads, 0));
    launch(d.stream, lo, hi);
    BBBP_CHECK_LAUNCH();
    BBBP_CHECK_HIP(hipEventRecord(d.ready, d.stream));
    launch(st, 0, lo);
    launch(st, hi, n);
    BBBP_CHECK_LAUNCH();
    {
        std::lock_guard<std::mutex> lock(g_deferred_mu);
        d.pending = true; d.lo = reinterpret_cast<const char*>(param + lo); d.hi = reinterpret_cast<const char*>(param + hi);
    }
    return BBBP_OK;
}
