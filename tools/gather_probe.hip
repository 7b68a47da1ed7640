// gather_probe — what does a scattered texel gather cost on this chip?  (tools only; not part of the library)
// Each lane reads `taps` 8-byte pairs from pseudo-random 64-byte-aligned places of a large buffer (larger than L2 + MALL),
// the pattern of an unmipped, heavily minified bilinear fetch.  Prints achieved lines/s; run under rocprofv3 --pmc
// FETCH_SIZE to see how many bytes the fabric counters attribute to one such access.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_gather(const uint2* __restrict__ src, uint32_t n_chunks_mask, uint32_t per_thread, uint32_t* out, uint32_t stride_chunks) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    uint32_t h = i * 2654435761u + 12345u;
    uint32_t acc = 0;
    for (uint32_t k = 0; k < per_thread; k += 4) {
        uint2 v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            h = h * 1664525u + 1013904223u;
            const uint32_t chunk = (h >> 4) & n_chunks_mask;                 // 64-byte chunk index
            v[j] = src[(size_t)chunk * 8u * stride_chunks + ((h >> 1) & 7u)];  // one 8-byte pair inside it
        }
#pragma unroll
        for (int j = 0; j < 4; j++) acc += v[j].x ^ v[j].y;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char** argv) {
    const size_t bytes = (argc > 1 ? (size_t)atoll(argv[1]) : 2048) << 20;     // MiB, power of two
    const uint32_t per_thread = argc > 2 ? atoi(argv[2]) : 16;
    const uint32_t n_threads = 8u << 20;
    uint2* src; uint32_t* out;
    CHK(hipMalloc(&src, bytes)); CHK(hipMalloc(&out, 4));
    CHK(hipMemset(src, 1, bytes));
    const uint32_t mask = (uint32_t)(bytes / 64) - 1u;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_gather, dim3(n_threads / 256), dim3(256), 0, 0, src, mask, per_thread, out, 1u);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        const double taps = (double)n_threads * per_thread;
        printf("buffer %zu MiB: %.0f M taps in %.3f ms = %.1f G taps/s = %.2f TB/s at 64 B/tap, %.2f TB/s at 128 B/tap\n", bytes >> 20, taps / 1e6, ms,
               taps / ms / 1e6, taps * 64 / ms / 1e9, taps * 128 / ms / 1e9);
    }
    return 0;
}
