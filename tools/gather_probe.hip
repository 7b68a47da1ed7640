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

// mode 1: the lean kernel's key read — 8 bytes per lane, consecutive lanes consecutive addresses (one 512-byte run per wavefront), streamed once
__global__ __launch_bounds__(256) void k_stream8(const uint2* __restrict__ src, size_t n, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u) { const uint2 v = src[i]; acc += v.x ^ v.y; }
    if (acc == 0x12345678u) out[0] = acc;
}
// mode 2: the lean kernel's triangle records — 208 contiguous bytes brought in by 13 lanes x 16 bytes, records at pseudo-random 208-byte slots
__global__ __launch_bounds__(256) void k_records(const uint4* __restrict__ src, uint32_t n_records_mask, uint32_t per_wave, uint32_t* out) {
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
    uint32_t h = wave * 2654435761u + 777u, acc = 0;
    for (uint32_t k = 0; k < per_wave; k++) {
        h = h * 1664525u + 1013904223u;
        const uint32_t rec = (h >> 3) & n_records_mask;
        if (lane < 13u) { const uint4 v = src[(size_t)rec * 13u + lane]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char** argv) {
    if (argc > 3 && atoi(argv[3]) == 1) {
        const size_t bytes = (size_t)atoll(argv[1]) << 20;
        uint2* src; uint32_t* out;
        CHK(hipMalloc(&src, bytes)); CHK(hipMalloc(&out, 4)); CHK(hipMemset(src, 1, bytes));
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        for (int rep = 0; rep < 3; rep++) {
            CHK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_stream8, dim3(16384), dim3(256), 0, 0, src, bytes / 8, out);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            printf("stream8 %zu MiB in %.3f ms = %.2f TB/s; true bytes per dispatch %zu\n", bytes >> 20, ms, bytes / ms / 1e9, bytes);
        }
        return 0;
    }
    if (argc > 3 && atoi(argv[3]) == 2) {
        const size_t bytes = (size_t)atoll(argv[1]) << 20;
        const uint32_t per_wave = atoi(argv[2]);
        uint32_t n_rec = 1; while ((size_t)n_rec * 2 * 208 <= bytes) n_rec *= 2;
        uint4* src; uint32_t* out;
        CHK(hipMalloc(&src, (size_t)n_rec * 208)); CHK(hipMalloc(&out, 4)); CHK(hipMemset(src, 1, (size_t)n_rec * 208));
        const uint32_t n_waves = 1u << 17;
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        for (int rep = 0; rep < 3; rep++) {
            CHK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_records, dim3(n_waves / 4), dim3(256), 0, 0, src, n_rec - 1u, per_wave, out);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            const double recs = (double)n_waves * per_wave;
            printf("records: table %zu MiB, %.1f M records of 208 B in %.3f ms = %.2f TB/s of payload; payload bytes per dispatch %.0f, as 64-B lines touched %.0f, as 128-B lines %.0f\n",
                   ((size_t)n_rec * 208) >> 20, recs / 1e6, ms, recs * 208 / ms / 1e9, recs * 208, recs * 256, recs * 320);
        }
        return 0;
    }
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
        printf("taps per dispatch %.0f; ", taps);
        printf("buffer %zu MiB: %.0f M taps in %.3f ms = %.1f G taps/s = %.2f TB/s at 64 B/tap, %.2f TB/s at 128 B/tap\n", bytes >> 20, taps / 1e6, ms,
               taps / ms / 1e6, taps * 64 / ms / 1e9, taps * 128 / ms / 1e9);
    }
    return 0;
}
