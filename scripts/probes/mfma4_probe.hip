// Operand layout of v_mfma_f32_4x4x1_16b_f32 on gfx950: D[l][r] = A[la] * B[lb] -> prints (la, lb) per (lane, reg).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
using f32x4 = __attribute__((ext_vector_type(4))) float;
__global__ void k(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}
int main() {
    float ha[64], hb[64], hd[256];
    for (int i = 0; i < 64; ++i) { ha[i] = 1.f + i; hb[i] = 128.f * (1 + i); }
    float *da, *db, *dd;
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 1024);
    hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(da, db, dd);
    hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            int fa = -1, fb = -1;
            for (int x = 0; x < 64 && fa < 0; ++x)
                for (int y = 0; y < 64; ++y)
                    if (hd[l * 4 + r] == ha[x] * hb[y]) { fa = x; fb = y; break; }
            // hypothesis: la = 4*(l/4) + r, lb = l
            if (fa != 4 * (l / 4) + r || fb != l) ok = 0;
            if (l < 8 || l == 21 || l == 63) printf("lane %d reg %d: A lane %d, B lane %d\n", l, r, fa, fb);
        }
    printf("hypothesis D[l][r] = A[4*(l/4)+r] * B[l]: %s\n", ok ? "HOLDS" : "FAILS");
    return 0;
}
