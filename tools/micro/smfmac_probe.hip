// Layout probe for v_smfmac_f32_32x32x32_bf16 (gfx950 2:4 structured-sparse MFMA): which dense-B slot does compressed-A element e of lane
// (row i, k-block kb) with 2-bit index p multiply?  A has ONE nonzero (= 1.0); B[lane][e'] = code(kb', e') = 1 + 16 kb' + e' for every column,
// so every element of D row i equals the code of the matched B slot.  Prints, per (kb, e, p, abid), the matched (kb', e').
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef __bf16 b16 __attribute__((ext_vector_type(16)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int ABID>
__global__ void probe(float* out, int kb, int e, int p, int idx_shift) {
    const int lane = threadIdx.x;
    b8 a; b16 b; f16v c;
    for (int i = 0; i < 8; ++i) a[i] = (__bf16)0.f;
    for (int i = 0; i < 16; ++i) { b[i] = (__bf16)(float)(1 + 16 * (lane >> 5) + i); c[i] = 0.f; }
    if (lane == 5 + 32 * kb) a[e] = (__bf16)1.f;                 // row 5
    // index register: 2 bits per compressed element; element e's field at bit 2 e (+ idx_shift)
    const int idx = (p << (2 * e)) << idx_shift;
    c = __builtin_amdgcn_smfmac_f32_32x32x32_bf16(a, b, c, idx, 0, ABID);
    for (int i = 0; i < 16; ++i) out[lane * 16 + i] = c[i];
}

int main() {
    float* d; hipMalloc(&d, 64 * 16 * sizeof(float));
    float h[64 * 16];
    for (int abid = 0; abid < 2; ++abid)
        for (int shift = 0; shift <= 16; shift += 16)
            for (int kb = 0; kb < 2; ++kb)
                for (int e = 0; e < 8; ++e)
                    for (int p = 0; p < 4; ++p) {
                        if (abid == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, d, kb, e, p, shift);
                        else hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64), 0, 0, d, kb, e, p, shift);
                        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
                        // find the nonzero values and where they are
                        float v = 0.f; int cnt = 0, first = -1;
                        for (int i = 0; i < 64 * 16; ++i) if (h[i] != 0.f) { if (first < 0) first = i; v = h[i]; ++cnt; }
                        const int code = (int)v - 1;
                        printf("abid %d shift %2d | A lane(row 5, kb %d) e %d idx %d -> B kb' %d e' %2d  (k = %2d)  nonzeros %d first at lane %d reg %d\n",
                               abid, shift, kb, e, p, code >> 4, code & 15, code, cnt, first / 16, first % 16);
                    }
    return 0;
}
