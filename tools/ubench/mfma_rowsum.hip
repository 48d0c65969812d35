// mfma_rowsum.hip -- checks the lane <-> element maps the linear kernel's wave reduction relies on:
// D[f][j] += sum_k term_f[16 k + j]  with A = (lane % 16 == f), B = term_f, v_mfma_f32_16x16x4_f32.
// hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_rowsum.hip -o tools/ubench/mfma_rowsum && tools/ubench/mfma_rowsum
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out)
{
    const int lane = threadIdx.x;
    f4 c = {0, 0, 0, 0};
    float sel = (lane & 15) == 0 ? 1.0f : 0.0f;       // (NOT inline asm: the compiler's hazard recognizer does not see a VALU write inside an asm
                                                       //  block and omits the wait states an MFMA reading the register needs -- measured here)
#pragma unroll
    for (int f = 0; f < 11; ++f) {
        if (f > 0) sel = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sel), 0x111, 0xF, 0xF, true));
        const float term = (float)(1000 * (f + 1) + lane);       // term_f[lane]
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(sel, term, c, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
}
int main()
{
    float* d; (void)hipMalloc(&d, 256 * 4);
    k<<<1, 64>>>(d);
    float h[256]; (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int f = 0; f < 16; ++f)
        for (int j = 0; j < 16; ++j) {
            const int lane = 16 * (f >> 2) + j, r = f & 3;
            float want = 0;
            if (f < 11) for (int kk = 0; kk < 4; ++kk) want += (float)(1000 * (f + 1) + 16 * kk + j);
            if (h[lane * 4 + r] != want) { if (bad < 10) printf("D[%d][%d] = %g, want %g\n", f, j, h[lane * 4 + r], want); ++bad; }
        }
    printf("%s (%d mismatches)\n", bad ? "MISMATCH" : "ok", bad);
    return bad != 0;
}
