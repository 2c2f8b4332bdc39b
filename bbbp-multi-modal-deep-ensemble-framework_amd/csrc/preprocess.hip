// Input pipeline at the tensor boundary, on the GPU: what the reference does per molecule on
// the host in Descriptors/multi_input_data_preprocess_maccs_opt_IsolationForest_fixed_1.py
//   :56-71   Image.open().convert('RGB') -> transforms.Resize((128,128)) -> ToTensor() -> flatten      (PIL + torchvision)
//   :86-101  StandardScaler().fit_transform on every chunk of 100 rows of hstack([MACCS u8, image f32])  (scikit-learn)
// Both are HBM-bound byte/float streaming and both are restated EXACTLY:
//   * the resize is Pillow's two-pass (horizontal, then vertical) antialiased bilinear resampling in its 8-bit
//     fixed-point form (coefficients rounded to 22 fractional bits, accumulate in int32, round, clip to u8 after each
//     pass); the coefficient tables are built on the host with Pillow's formulas (preprocess.py) and the kernels do
//     integer arithmetic only => bit-identical bytes;
//   * ToTensor is u8 -> f32 / 255 in CHW order (one IEEE division, same rounding as torch);
//   * the scaler accumulates in float64 over the rows IN ORDER (numpy's axis-0 reduction order), uses scikit-learn's
//     corrected two-pass variance and writes float32(float64(x) - mean) then float32(float64(.) / scale).
// PNG decoding stays on the host (byte-serial entropy decoding; not a GPU job).
#include "common.h"
#include "bbbp_hip.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;      // Pillow: src/libImaging/Resample.c

__device__ __forceinline__ uint8_t clip8(int v) {
    v >>= PRECISION_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: src [N][Hs][Ws][3] u8 -> tmp [N][Hs][Wo][3] u8.  One thread per output byte.
__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* src, uint8_t* tmp, const int* bounds, const int* kk,
                                                      int ksize, long total, int Hs, int Ws, int Wo) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        int c = (int)(idx % 3);
        int xo = (int)((idx / 3) % Wo);
        long row = idx / (3L * Wo);                  // n * Hs + y
        int xmin = bounds[2 * xo], cnt = bounds[2 * xo + 1];
        const int* k = kk + (long)xo * ksize;
        const uint8_t* s = src + (row * Ws + xmin) * 3 + c;
        int ss = 1 << (PRECISION_BITS - 1);
        for (int x = 0; x < cnt; ++x) ss += (int)s[3 * x] * k[x];
        tmp[idx] = clip8(ss);
    }
}

// vertical pass + ToTensor: tmp [N][Hs][Wo][3] u8 -> dst8 [N][Ho][Wo][3] u8 (optional) and dstf [N][3][Ho][Wo] f32 / 255.
__global__ __launch_bounds__(256) void resize_v_totensor_kernel(const uint8_t* tmp, uint8_t* dst8, float* dstf, const int* bounds,
                                                               const int* kk, int ksize, long total, int Hs, int Ho, int Wo) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        int c = (int)(idx % 3);
        int xo = (int)((idx / 3) % Wo);
        int yo = (int)((idx / (3L * Wo)) % Ho);
        long n = idx / (3L * Wo * Ho);
        int ymin = bounds[2 * yo], cnt = bounds[2 * yo + 1];
        const int* k = kk + (long)yo * ksize;
        const uint8_t* s = tmp + ((n * Hs + ymin) * Wo + xo) * 3 + c;
        int ss = 1 << (PRECISION_BITS - 1);
        for (int y = 0; y < cnt; ++y) ss += (int)s[(long)y * Wo * 3] * k[y];
        uint8_t v = clip8(ss);
        if (dst8) dst8[idx] = v;
        dstf[((n * 3 + c) * Ho + yo) * Wo + xo] = (float)v / 255.0f;
    }
}

// StandardScaler.fit_transform over the rows of one chunk, one thread per column.  Columns [0, F) come from the u8
// fingerprint matrix, columns [F, F + I) from the f32 image matrix (the reference hstacks them: u8 -> f32 exactly).
__global__ __launch_bounds__(256) void standardize_chunk_kernel(const uint8_t* fp, const float* img, float* fp_out, float* img_out,
                                                               double* mean_out, double* scale_out, int n, int F, int I) {
    int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= F + I) return;
    const bool is_fp = col < F;
    const int c = is_fp ? col : col - F;
    auto at = [&](int r) -> double { return is_fp ? (double)fp[(long)r * F + c] : (double)img[(long)r * I + c]; };
    double sum = 0.0;
    for (int r = 0; r < n; ++r) sum += at(r);
    const double mean = sum / n;
    double corr = 0.0, sq = 0.0;
    for (int r = 0; r < n; ++r) { double t = at(r) - mean; corr += t; sq += t * t; }
    double var = (sq - corr * corr / n) / n;
    double scale = sqrt(var);
    // sklearn _handle_zeros_in_scale: (near-)constant features keep their values
    if (scale < 10.0 * 2.220446049250313e-16) scale = 1.0;
    if (mean_out) { mean_out[col] = mean; scale_out[col] = scale; }
    for (int r = 0; r < n; ++r) {
        float centered = (float)(at(r) - mean);                 // X -= mean_ on a float32 array
        float v = (float)((double)centered / scale);            // X /= scale_
        if (is_fp) fp_out[(long)r * F + c] = v; else img_out[(long)r * I + c] = v;
    }
}

inline int grid_for(long n) {
    long g = (n + 255) / 256;
    long cap = (long)bbbp_num_cus() * 8;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int bbbp_resize_bilinear_totensor(void* stream, const uint8_t* src, uint8_t* tmp, uint8_t* dst_u8, float* dst_chw,
                                             const int* bounds_x, const int* kk_x, int ksize_x, const int* bounds_y,
                                             const int* kk_y, int ksize_y, int N, int Hs, int Ws, int Ho, int Wo) {
    BBBP_CHECK_ARG(N >= 0 && Hs > 0 && Ws > 0 && Ho > 0 && Wo > 0 && ksize_x > 0 && ksize_y > 0, "resize: bad sizes");
    if (N == 0) return BBBP_OK;          // an empty batch has no buffers
    BBBP_CHECK_ARG(src && tmp && dst_chw && bounds_x && kk_x && bounds_y && kk_y, "resize: null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    long t1 = (long)N * Hs * Wo * 3, t2 = (long)N * Ho * Wo * 3;
    hipLaunchKernelGGL(resize_h_kernel, dim3(grid_for(t1)), dim3(256), 0, st, src, tmp, bounds_x, kk_x, ksize_x, t1, Hs, Ws, Wo);
    BBBP_CHECK_LAUNCH();
    hipLaunchKernelGGL(resize_v_totensor_kernel, dim3(grid_for(t2)), dim3(256), 0, st, tmp, dst_u8, dst_chw, bounds_y, kk_y, ksize_y,
                       t2, Hs, Ho, Wo);
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}

extern "C" int bbbp_standardize_chunk(void* stream, const uint8_t* fingerprint_u8, const float* image, float* fingerprint_out,
                                      float* image_out, double* mean_out, double* scale_out, int rows, int F, int I) {
    BBBP_CHECK_ARG(rows >= 1 && F >= 0 && I >= 0 && F + I > 0, "standardize: bad sizes rows=%d F=%d I=%d", rows, F, I);
    BBBP_CHECK_ARG((F == 0 || (fingerprint_u8 && fingerprint_out)) && (I == 0 || (image && image_out)), "standardize: null pointer");
    hipLaunchKernelGGL(standardize_chunk_kernel, dim3(cdiv(F + I, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       fingerprint_u8, image, fingerprint_out, image_out, mean_out, scale_out, rows, F, I);
    BBBP_CHECK_LAUNCH();
    return BBBP_OK;
}
