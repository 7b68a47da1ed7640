// Queue probe (MI355X): when does a machine-filling kernel on one stream start while another machine-filling kernel on a second stream is in its
// tail?  The overlapped frame pipeline wants the opaque pass of frame i + 1 (stream B) to move into the slots the opaque pass of frame i
// (stream A) leaves; the kernel trace showed it starting only after A had run dry.  Device-side timestamps (s_memrealtime, 100 MHz) of the first
// and last workgroup of every kernel, for a few arrangements of the same work.
// build: hipcc --offload-arch=gfx950 -O2 tools/queue_probe.hip -o tools/queue_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)

__device__ inline unsigned long long now() { return __builtin_amdgcn_s_memrealtime(); }

struct Stamp { unsigned long long first, last; };

// every workgroup spins `ticks` of the 100 MHz clock; 26 KB of LDS per workgroup: six workgroups per CU, like the opaque kernel's occupancy
__global__ __launch_bounds__(256) void k_busy(Stamp* st, uint32_t ticks) {
    extern __shared__ uint32_t lds[];
    const unsigned long long t0 = now();
    if (threadIdx.x == 0) { lds[0] = 1; if ((blockIdx.x & 63u) == 0u) atomicMin(&st->first, t0); }
    while (now() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
    if (threadIdx.x == 0 && (blockIdx.x & 63u) == 0u) atomicMax(&st->last, now());
}
// variant: VGPR-limited occupancy (64 live accumulators -> ~80 VGPRs, six waves per SIMD, no LDS) and, with `out`, 2 KB of non-temporal stores
// per workgroup and round (the opaque kernel writes 66 MB per frame)
__global__ __launch_bounds__(256) void k_busy_vgpr(Stamp* st, uint32_t ticks, float* out, float seed) {
    const unsigned long long t0 = now();
    if (threadIdx.x == 0 && (blockIdx.x & 63u) == 0u) atomicMin(&st->first, t0);
    float acc[64];
#pragma unroll
    for (int i = 0; i < 64; i++) acc[i] = seed * (float)(i + threadIdx.x);
    uint32_t round = 0;
    while (now() - t0 < ticks) {
#pragma unroll
        for (int i = 0; i < 64; i++) acc[i] = fmaf(acc[i], 1.0001f, seed);
        if (out) __builtin_nontemporal_store(acc[round & 63u], out + ((size_t)blockIdx.x * 64u + (round & 63u)) * 256u + threadIdx.x);
        round++;
    }
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < 64; i++) sum += acc[i];
    if (sum == 12345.678f && out) out[0] = sum;
    if (threadIdx.x == 0 && (blockIdx.x & 63u) == 0u) atomicMax(&st->last, now());
}
__global__ void k_tiny(Stamp* st) { const unsigned long long t = now(); st->first = t; st->last = t; }
__global__ void k_spin_one(Stamp* st, uint32_t ticks) { const unsigned long long t0 = now(); st->first = t0; while (now() - t0 < ticks) __builtin_amdgcn_s_sleep(8); st->last = now(); }
__global__ void k_signal(Stamp* st, uint32_t* flag, uint32_t v) { st->first = now(); __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); st->last = now(); }
__global__ void k_wait(Stamp* st, const uint32_t* flag, uint32_t v) {
    st->first = now();
    uint32_t polls = 0;
    while ((int32_t)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - v) < 0 && ++polls < (1u << 20)) __builtin_amdgcn_s_sleep(8);
    st->last = now();
}

int main(int argc, char** argv) {
    const int scenario = argc > 1 ? atoi(argv[1]) : 0;
    const int reps = 4;
    hipStream_t sG, sA, sB;
    CHK(hipStreamCreateWithFlags(&sG, hipStreamNonBlocking));
    CHK(hipStreamCreateWithFlags(&sA, hipStreamNonBlocking));
    CHK(hipStreamCreateWithFlags(&sB, hipStreamNonBlocking));
    Stamp* st; uint32_t* flags;
    const int n_st = 64;
    CHK(hipMalloc(&st, n_st * sizeof(Stamp)));
    CHK(hipMalloc(&flags, 64));
    std::vector<Stamp> init(n_st, Stamp{~0ull, 0ull}), got(n_st);
    const uint32_t grid = 34816, wg_ticks = 1500;      // 15 us per workgroup: 34816 / (256 CUs x 6) = 23 rounds = ~340 us
    const size_t lds = 26 * 1024;
    float* big = nullptr;
    if (scenario & 4) CHK(hipMalloc(&big, (size_t)grid * 64 * 256 * sizeof(float)));
    auto busy = [&](Stamp* sp, hipStream_t s) {
        if (scenario & 12) hipLaunchKernelGGL(k_busy_vgpr, dim3(grid), dim3(256), 0, s, sp, wg_ticks, big, 1e-9f);
        else hipLaunchKernelGGL(k_busy, dim3(grid), dim3(256), lds, s, sp, wg_ticks);
    };
    hipEvent_t ev;
    CHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    for (int rep = 0; rep < reps; rep++) {
        CHK(hipMemcpy(st, init.data(), n_st * sizeof(Stamp), hipMemcpyHostToDevice));
        CHK(hipMemset(flags, 0, 64));
        CHK(hipDeviceSynchronize());
        std::vector<std::string> names(n_st);
        int k = 0;
        auto slot = [&](const char* n) { names[k] = n; return st + k++; };
        // frame i: its opaque pass K1 on stream A, then the two small kernels that follow it
        busy(slot("A busy1"), sA);
        hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, sA, slot("A tiny(todo)"));
        hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, sA, slot("A signal"), flags + 1, 1u);
        if (scenario & 1) CHK(hipEventRecord(ev, sA));
        // "geometry" of frame i + 1 on stream G: one workgroup spinning 250 us (a dependency chain, not a load), then the signal
        hipLaunchKernelGGL(k_spin_one, dim3(1), dim3(64), 0, sG, slot("G chain"), 25000u);
        hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, sG, slot("G signal"), flags + 0, 1u);
        // frame i + 1's opaque pass K2
        hipStream_t s2 = (scenario & 2) ? sG : sB;      // scenarios 2, 3: K2 in order behind the geometry on G instead of behind a gate on B
        if (!(scenario & 2)) hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, sB, slot("B gate"), flags + 0, 1u);
        busy(slot("busy2"), s2);
        CHK(hipDeviceSynchronize());
        CHK(hipMemcpy(got.data(), st, n_st * sizeof(Stamp), hipMemcpyDeviceToHost));
        if (rep < 2) continue;
        const unsigned long long t0 = got[0].first;
        printf("scenario %d rep %d:", scenario, rep);
        for (int i = 0; i < k; i++) printf("  %s [%.1f, %.1f]", names[i].c_str(), (double)(got[i].first - t0) / 100.0, (double)(got[i].last - t0) / 100.0);
        printf("\n");
    }
    return 0;
}
