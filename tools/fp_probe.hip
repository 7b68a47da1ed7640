// fp_probe: are f32 div / sqrt / 1/sqrt bit-identical between gfx950 device code and x86 host code?
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
__global__ void k(const float* a, const float* b, float* o, int n) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float x = a[i], y = b[i];
    o[i * 4 + 0] = x / y;
    o[i * 4 + 1] = sqrtf(y);
    o[i * 4 + 2] = 1.0f / __builtin_sqrtf(y);
    o[i * 4 + 3] = x * (1.0f / sqrtf(y));
}
int main() {
    const int n = 1 << 20;
    std::vector<float> a(n), b(n), o(n * 4);
    srand(1);
    for (int i = 0; i < n; i++) { a[i] = (rand() / (float)RAND_MAX) * 2.f - 1.f; b[i] = (rand() / (float)RAND_MAX) * 1.5f + 1e-3f; }
    float *da, *db, *dout;
    (void)hipMalloc(&da, n * 4); (void)hipMalloc(&db, n * 4); (void)hipMalloc(&dout, n * 16);
    (void)hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(da, db, dout, n);
    (void)hipMemcpy(o.data(), dout, n * 16, hipMemcpyDeviceToHost);
    int bad[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; i++) {
        volatile float x = a[i], y = b[i];
        volatile float s = sqrtf(y);
        volatile float r = 1.0f / s;
        float ref[4] = {x / y, s, r, x * r};
        for (int j = 0; j < 4; j++) if (memcmp(&ref[j], &o[i * 4 + j], 4)) { if (bad[j] < 3) printf("j=%d x=%a y=%a gpu=%a cpu=%a\n", j, a[i], b[i], o[i*4+j], ref[j]); bad[j]++; }
    }
    printf("mismatches div=%d sqrt=%d rsqrt=%d mulrsqrt=%d of %d\n", bad[0], bad[1], bad[2], bad[3], n);
    return 0;
}
