// valu_probe: per-wave issue cost and per-SIMD throughput of the VALU instructions the shading kernels are made of (gfx950).
// One workgroup per CU; W waves per SIMD (W = 1, 2, 4, 8); every wave runs N x 32 independent instructions of one kind between two
// s_memtime stamps.  Prints SIMD cycles per wave-instruction = elapsed / (N * 32 * W): the number that prices an instruction in a
// kernel with W resident waves.  build: hipcc --offload-arch=gfx950 -O3 -o tools/valu_probe.bin tools/valu_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

template <int KIND>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, int n, float seed) {
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {seed, seed}, p1 = p0 + 1.0f, p2 = p0 + 2.0f, p3 = p0 + 3.0f;
    double d0 = seed, d1 = seed + 1, d2 = seed + 2, d3 = seed + 3;
    unsigned u0 = (unsigned)seed + threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;
    unsigned long long q0 = u0, q1 = u1;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i++) {
        if (KIND == 0) { REP8(asm volatile("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %4, %4, %4, %5\n v_fma_f32 %6, %6, %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (KIND == 1) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %0, %1\n v_pk_fma_f32 %1, %1, %1, %2\n v_pk_fma_f32 %2, %2, %2, %3\n v_pk_fma_f32 %3, %3, %3, %0" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));) }
        if (KIND == 2) { REP8(asm volatile("v_cvt_f32_ubyte1_e32 %0, %4\n v_cvt_f32_ubyte2_e32 %1, %5\n v_cvt_f32_ubyte0_e32 %2, %6\n v_cvt_f32_ubyte3_e32 %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(u0), "v"(u1), "v"(u2), "v"(u3));) }
        if (KIND == 3) { REP8(asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %1, %1, %2\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %3, %3, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
        if (KIND == 4) { REP8(asm volatile("v_mad_u32_u24 %0, %0, %1, %2\n v_mad_u32_u24 %1, %1, %2, %3\n v_mad_u32_u24 %2, %2, %3, %0\n v_mad_u32_u24 %3, %3, %0, %1" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
        if (KIND == 5) { REP8(asm volatile("v_fma_f64 %0, %0, %0, %1\n v_fma_f64 %1, %1, %1, %2\n v_fma_f64 %2, %2, %2, %3\n v_fma_f64 %3, %3, %3, %0" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
        if (KIND == 6) { REP8(asm volatile("v_lshl_add_u64 %0, %0, 2, %1\n v_lshl_add_u64 %1, %1, 2, %0\n v_lshl_add_u64 %0, %0, 2, %1\n v_lshl_add_u64 %1, %1, 2, %0" : "+v"(q0), "+v"(q1));) }
        if (KIND == 7) { REP8(asm volatile("v_rcp_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rsq_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (KIND == 8) { REP8(asm volatile("v_bfe_u32 %0, %0, 0, %1\n v_lshl_or_b32 %1, %1, %2, %3\n v_bfe_u32 %2, %2, 0, %3\n v_lshl_or_b32 %3, %3, %0, %1" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
        if (KIND == 9) { REP8(asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_add_f32 %1, %1, %2\n v_pk_mul_f32 %2, %2, %3\n v_pk_add_f32 %3, %3, %0" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));) }
        if (KIND == 10) { REP8(asm volatile("v_cvt_f64_f32 %0, %2\n v_cvt_f32_f64 %3, %1\n v_cvt_f64_f32 %1, %3\n v_cvt_f32_f64 %2, %0" : "+v"(d0), "+v"(d1), "+v"(a0), "+v"(a1));) }
        if (KIND == 11) { REP8(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
        if (KIND == 12) { REP8(asm volatile("v_cvt_f16_f32 %0, %0\n v_cvt_f32_f16 %0, %0\n v_cvt_f16_f32 %1, %1\n v_cvt_f32_f16 %1, %1" : "+v"(a0), "+v"(a1));) }
        if (KIND == 13) { REP8(asm volatile("v_exp_f32 %0, %0\n v_log_f32 %1, %1\n v_sin_f32 %2, %2\n v_cos_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (KIND == 14) { REP8(asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0\n v_div_fmas_f32 %1, %1, %2, %3\n v_div_fixup_f32 %2, %2, %3, %0\n v_div_scale_f32 %3, vcc, %3, %0, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc");) }
        if (KIND == 20) { REP8(asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %1, %1, %2\n v_add_f32 %2, %2, %3\n v_add_f32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (KIND == 21) { REP8(asm volatile("v_mul_f32 %0, %0, %1\n v_mul_f32 %1, %1, %2\n v_mul_f32 %2, %2, %3\n v_mul_f32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (KIND == 22) { REP8(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
        if (KIND == 23) { REP8(asm volatile("v_and_b32 %0, %0, %1\n v_or_b32 %1, %1, %2\n v_xor_b32 %2, %2, %3\n v_and_b32 %3, %3, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
        if (KIND == 24) { REP8(asm volatile("v_lshlrev_b32 %0, 3, %0\n v_lshrrev_b32 %1, 3, %1\n v_lshlrev_b32 %2, 5, %2\n v_lshrrev_b32 %3, 7, %3" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
        if (KIND == 25) { REP8(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
        if (KIND == 26) { REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc");) }
        if (KIND == 27) { REP8(asm volatile("v_max_f32 %0, %0, %1\n v_min_f32 %1, %1, %2\n v_max_f32 %2, %2, %3\n v_min_f32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (KIND == 28) { REP8(asm volatile("v_fmac_f32 %0, %1, %2\n v_fmac_f32 %1, %2, %3\n v_fmac_f32 %2, %3, %0\n v_fmac_f32 %3, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (KIND == 29) { REP8(asm volatile("v_floor_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_floor_f32 %2, %2\n v_cvt_f32_u32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (KIND == 30) { REP8(asm volatile("v_ldexp_f32 %0, %0, %4\n v_ldexp_f32 %1, %1, %4\n v_ldexp_f32 %2, %2, %4\n v_ldexp_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(u0));) }
        if (KIND == 31) { REP8(asm volatile("v_fma_f32 %0, %1, %2, %3\n v_fma_f32 %1, %2, %3, %0\n v_fma_f32 %2, %3, %0, %1\n v_fma_f32 %3, %0, %1, %2" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (KIND == 33) { REP8(asm volatile("v_add3_u32 %0, %1, %2, %3\n v_lshl_add_u32 %1, %2, 2, %0\n v_add3_u32 %2, %3, %0, %1\n v_lshl_add_u32 %3, %0, 2, %2" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
        if (KIND == 34) { REP8(asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0");) }
        if (KIND == 35) { REP8(asm volatile("v_sub_f32 %0, 1.0, %0\n v_sub_f32 %1, 1.0, %1\n v_mul_f32 %2, 0x3b808081, %2\n v_mul_f32 %3, 2.0, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (KIND == 40) { REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[4:5]\n v_cndmask_b32_e64 %1, %1, %2, s[4:5]\n v_cndmask_b32_e64 %2, %2, %3, s[4:5]\n v_cndmask_b32_e64 %3, %3, %0, s[4:5]" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : : "s4", "s5");) }
        if (KIND == 41) { REP8(asm volatile("v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %5, %6, vcc\n v_cndmask_b32 %2, %6, %7, vcc\n v_cndmask_b32 %3, %7, %4, vcc" : "=v"(u0), "=v"(u1), "=v"(u2), "=v"(u3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));) }
        if (KIND == 42) { REP8(asm volatile("v_cmp_lt_f32 vcc, %4, %5\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_lt_f32 vcc, %6, %7\n v_cndmask_b32 %2, %2, %3, vcc" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");) }
        if (KIND == 43) { REP8(asm volatile("v_cmp_lt_f32 s[4:5], %4, %5\n v_cmp_lt_f32 s[6:7], %6, %7\n v_cndmask_b32_e64 %0, %0, %1, s[4:5]\n v_cndmask_b32_e64 %2, %2, %3, s[6:7]" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "s4", "s5", "s6", "s7");) }
        if (KIND == 44) { REP8(asm volatile("v_max3_f32 %0, %0, %1, %2\n v_med3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %0\n v_med3_f32 %3, %3, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (KIND == 45) { REP8(asm volatile("v_fma_f32 %0, %0, %0, s4\n v_fma_f32 %1, %1, s5, %1\n v_mul_f32 %2, s6, %2\n v_add_f32 %3, s7, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (KIND == 46) { REP8(asm volatile("v_fma_f32 %0, |%0|, %0, -%1\n v_fma_f32 %1, %1, %1, %2 clamp\n v_mul_f32_e64 %2, %2, %3 clamp\n v_add_f32_e64 %3, |%3|, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (KIND == 47) { REP8(asm volatile("v_perm_b32 %0, %0, %1, %2\n v_alignbit_b32 %1, %1, %2, 8\n v_perm_b32 %2, %2, %3, %0\n v_alignbit_b32 %3, %3, %0, 16" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
        if (KIND == 51) { REP8(asm volatile("v_perm_b32 %0, %0, %1, %2\n v_perm_b32 %1, %1, %2, %3\n v_perm_b32 %2, %2, %3, %0\n v_perm_b32 %3, %3, %0, %1" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
        if (KIND == 52) { REP8(asm volatile("v_dot2_u32_u16 %0, %0, %1, %2\n v_dot2_u32_u16 %1, %1, %2, %3\n v_dot2_u32_u16 %2, %2, %3, %0\n v_dot2_u32_u16 %3, %3, %0, %1" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
        if (KIND == 53) { REP8(asm volatile("v_dot4_u32_u8 %0, %0, %1, %2\n v_dot4_u32_u8 %1, %1, %2, %3\n v_dot4_u32_u8 %2, %2, %3, %0\n v_dot4_u32_u8 %3, %3, %0, %1" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
        if (KIND == 54) { REP8(asm volatile("v_mad_u32_u16 %0, %0, %1, %2\n v_mad_u32_u16 %1, %1, %2, %3\n v_mad_u32_u16 %2, %2, %3, %0\n v_mad_u32_u16 %3, %3, %0, %1" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
        if (KIND == 55) { REP8(asm volatile("v_dot2_f32_f16 %0, %4, %5, %0\n v_dot2_f32_f16 %1, %5, %6, %1\n v_dot2_f32_f16 %2, %6, %7, %2\n v_dot2_f32_f16 %3, %7, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(u0), "v"(u1), "v"(u2), "v"(u3));) }
        if (KIND == 48) { REP8(asm volatile("v_cvt_f32_u32 %0, %4\n v_cvt_f32_i32 %1, %5\n v_cvt_u32_f32 %6, %2\n v_cvt_f32_u32 %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
        if (KIND == 49) { REP8(asm volatile("v_sub_u32 %0, %0, %1\n v_subrev_u32 %1, %1, %2\n v_min_u32 %2, %2, %3\n v_max_i32 %3, %3, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
        if (KIND == 50) { REP8(asm volatile("v_readfirstlane_b32 s4, %0\n v_readfirstlane_b32 s5, %1\n v_readfirstlane_b32 s6, %2\n v_readfirstlane_b32 s7, %3" : : "v"(u0), "v"(u1), "v"(u2), "v"(u3) : "s4", "s5", "s6", "s7");) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + (float)(d0 + d1 + d2 + d3) + (float)(u0 ^ u1 ^ u2 ^ u3) + (float)(q0 ^ q1);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = (t1 - t0) + (s == 12345.678f ? 1 : 0);
}

template <int KIND>
void run(const char* name, unsigned long long* dout) {
    const int n = 2000, blocks = 256;
    printf("%-40s", name);
    for (int w : {1, 2, 4, 8}) {     // waves per SIMD = workgroup of w * 256 threads, one workgroup per CU (256 blocks: one per CU if dispatched evenly)
        if (w * 256 > 1024) {       // 8 waves/SIMD needs two 1024-thread workgroups per CU
            hipLaunchKernelGGL(k<KIND>, dim3(blocks * 2), dim3(1024), 0, 0, dout, n, 1.0f);
        } else {
            hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(w * 256), 0, 0, dout, n, 1.0f);
        }
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks * 2 * 16);
        (void)hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> v;
        const int nb = w == 8 ? blocks * 2 : blocks, nw = w == 8 ? 16 : w * 4;
        for (int b = 0; b < nb; b++) for (int j = 0; j < nw; j++) v.push_back((double)h[b * 16 + j]);
        std::sort(v.begin(), v.end());
        const double med = v[v.size() / 2];
        // s_memtime counts at a fixed 100 MHz?  (guide: "tick = shader cycle"); report ticks per instruction per wave and per SIMD
        printf("  w=%d: %6.2f/wave %5.2f/simd", w, med / (n * 32.0), med / (n * 32.0 * w));
    }
    printf("\n");
}

int main() {
    unsigned long long* dout;
    (void)hipMalloc(&dout, 512 * 16 * 8);
    (void)hipMemset(dout, 0, 512 * 16 * 8);
    printf("ticks per wave-instruction (per wave / per SIMD with w waves resident per SIMD)\n");
    run<0>("v_fma_f32", dout);
    run<1>("v_pk_fma_f32", dout);
    run<9>("v_pk_mul_f32 / v_pk_add_f32", dout);
    run<2>("v_cvt_f32_ubyteN", dout);
    run<3>("v_mul_lo_u32", dout);
    run<4>("v_mad_u32_u24", dout);
    run<5>("v_fma_f64", dout);
    run<10>("v_cvt_f64_f32 / v_cvt_f32_f64", dout);
    run<6>("v_lshl_add_u64", dout);
    run<7>("v_rcp_f32 / v_rsq_f32", dout);
    run<13>("v_exp/log/sin/cos_f32", dout);
    run<8>("v_bfe_u32 / v_lshl_or_b32", dout);
    run<11>("v_mov_b32", dout);
    run<12>("v_cvt_f16_f32 / v_cvt_f32_f16", dout);
    run<14>("v_div_scale / fmas / fixup", dout);
    run<20>("v_add_f32", dout);
    run<21>("v_mul_f32", dout);
    run<35>("v_sub/mul_f32 with inline/literal const", dout);
    run<28>("v_fmac_f32 (VOP2)", dout);
    run<31>("v_fma_f32 3 distinct srcs", dout);
    run<27>("v_max_f32 / v_min_f32", dout);
    run<22>("v_add_u32", dout);
    run<23>("v_and/or/xor_b32", dout);
    run<24>("v_lshlrev/lshrrev_b32", dout);
    run<33>("v_add3_u32 / v_lshl_add_u32", dout);
    run<25>("v_cndmask_b32", dout);
    run<26>("v_cmp_lt_f32", dout);
    run<29>("v_floor/cvt_i32/cvt_f32_u32", dout);
    run<30>("v_ldexp_f32", dout);
    run<34>("s_nop (loop overhead)", dout);
    run<40>("v_cndmask_b32_e64 sgpr mask", dout);
    run<41>("v_cndmask_b32 vcc, independent", dout);
    run<42>("v_cmp vcc + v_cndmask vcc pairs", dout);
    run<43>("v_cmp sgpr + v_cndmask sgpr pairs", dout);
    run<44>("v_max3_f32 / v_med3_f32", dout);
    run<45>("v_fma/mul/add with SGPR operand", dout);
    run<46>("fma/mul/add with abs/neg/clamp modifiers", dout);
    run<47>("v_perm_b32 / v_alignbit_b32", dout);
    run<51>("v_perm_b32", dout);
    run<52>("v_dot2_u32_u16", dout);
    run<53>("v_dot4_u32_u8", dout);
    run<54>("v_mad_u32_u16", dout);
    run<55>("v_dot2_f32_f16", dout);
    run<48>("v_cvt_f32_u32/i32, v_cvt_u32_f32", dout);
    run<49>("v_sub_u32 / v_min_u32 / v_max_i32", dout);
    run<50>("v_readfirstlane_b32", dout);
    return 0;
}
