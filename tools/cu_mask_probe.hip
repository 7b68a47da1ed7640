// Which CUs does a stream created with hipExtStreamCreateWithCUMask run on?  (The order of the mask's bits over XCDs / shader engines is not
// documented anywhere this repo can read.)  Usage: cu_mask_probe.bin <hex words, least significant first, comma separated>
//   e.g.  cu_mask_probe.bin ffffffff,ffffffff,ffffffff,ffffffff     -> bits 0..127
// Prints, per XCD, how many distinct (se, sh, cu) places ran a workgroup, and the total.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ __launch_bounds__(64) void k_where(uint32_t* out, unsigned long long ticks) {
    const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u;      // HW_REG_XCC_ID[3:0]
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);            // HW_REG_HW_ID
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);                      // long enough for the whole grid to be resident together
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}

int main(int argc, char** argv) {
    std::vector<uint32_t> mask;
    if (argc > 1) { char* s = strdup(argv[1]); for (char* p = strtok(s, ","); p; p = strtok(nullptr, ",")) mask.push_back((uint32_t)strtoul(p, nullptr, 16)); }
    hipStream_t st;
    if (mask.empty()) CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    else CHECK(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
    const int n = 8192;
    uint32_t* d; CHECK(hipMalloc(&d, n * 8));
    int khz = 100000; hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
    hipLaunchKernelGGL(k_where, dim3(n), dim3(64), 0, st, d, (unsigned long long)khz / 20);   // 50 us per workgroup
    CHECK(hipStreamSynchronize(st));
    std::vector<uint32_t> h(2 * n); CHECK(hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost));
    std::set<uint32_t> places[16];
    for (int i = 0; i < n; i++) places[h[2 * i] & 15u].insert((h[2 * i + 1] >> 8) & 0xFFu);      // cu_id[11:8], sh_id[12], se_id[15:13]
    size_t total = 0;
    for (int x = 0; x < 16; x++) if (!places[x].empty()) {
        printf("xcd %d: %zu CUs:", x, places[x].size());
        for (uint32_t p : places[x]) printf(" %x", p);
        printf("\n"); total += places[x].size();
    }
    printf("total %zu CUs\n", total);
    return 0;
}
