// Microbenchmark: VALU issue rate of one CU-resident wave mix on gfx950 -- how many cycles a wave64 VALU instruction
// of the kinds this library's kernels are made of occupies a SIMD.  One workgroup per CU, W waves per SIMD; every wave
// runs N independent chains of one instruction type; cycles from s_memtime.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHAINS 8
#define ITERS 4096

template <int KIND>
__global__ void __launch_bounds__(1024) k_issue(unsigned long long* out, unsigned seed) {
    unsigned v[CHAINS];
    float f[CHAINS];
    double d2[CHAINS];
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) v[i] = seed + threadIdx.x * 7 + i, f[i] = (float)(seed + i), d2[i] = (double)i;
    unsigned long long t0, t1;
    const unsigned long long mask64 = 0x5555555555555555ull ^ seed;
    asm volatile("s_mov_b64 vcc, %0" ::"s"(mask64) : "vcc");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) {
            if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) % CHAINS]));
            if (KIND == 1) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(v[i]) : "v"(v[(i + 1) % CHAINS]));
            if (KIND == 2) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) % CHAINS]));
            if (KIND == 3) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(v[(i + 1) % CHAINS]));
            if (KIND == 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(f[(i + 1) % CHAINS]));
            if (KIND == 5) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(v[i]) : "v"(v[(i + 1) % CHAINS]));
            if (KIND == 6) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(v[i]));
            if (KIND == 7) asm volatile("v_alignbit_b32 %0, %0, %1, 5" : "+v"(v[i]) : "v"(v[(i + 1) % CHAINS]));
            if (KIND == 8) asm volatile("v_add_u32_e64 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) % CHAINS]));            // VOP3 encoding, two sources
            if (KIND == 9) asm volatile("v_add_u32 %0, 0x12345, %0" : "+v"(v[i]));                                      // VOP2 + 32-bit literal (8 bytes)
            if (KIND == 10) asm volatile("v_and_b32 %0, 0xff00ff, %0" : "+v"(v[i]));                                    // VOP2 + literal
            if (KIND == 11) asm volatile("v_add_u32 %0, %1, %0" : "+v"(v[i]) : "s"(seed));                              // VOP2, SGPR source
            if (KIND == 12) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(v[i]) : "v"(v[(i + 1) % CHAINS]));       // VOP3, two distinct registers
            if (KIND == 13) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d2[i]) : "v"(d2[(i + 1) % CHAINS]));          // packed fp32 (64-bit operands)
            if (KIND == 14) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(v[(i + 1) % CHAINS]) : );   // VOP2 + VCC (set once)
            if (KIND == 16) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[(i + 1) % CHAINS]), "s"(mask64));  // SGPR-pair mask
            if (KIND == 17) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(v[(i + 1) % CHAINS]) : "vcc");  // pair: counts as 2
            if (KIND == 18) asm volatile("v_min_u32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) % CHAINS]));
            if (KIND == 19) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(v[i]));
            if (KIND == 20) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) % CHAINS]));
            if (KIND == 21) asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(v[i]));
            if (KIND == 22) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(v[(i + 1) % CHAINS]));
            if (KIND == 23) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(v[(i + 1) % CHAINS]));
            if (KIND == 15) asm volatile("v_dot2_i32_i16 %0, %0, %1, %0" : "+v"(v[i]) : "v"(v[(i + 1) % CHAINS]));
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) acc += v[i] + (unsigned)f[i] + (unsigned)d2[i];
    if (threadIdx.x % 64 == 0) out[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2] = t1 - t0, out[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2 + 1] = acc;
}

template <int KIND>
double run(int waves_per_simd, unsigned long long* d) {
    const int threads = waves_per_simd * 4 * 64;
    hipLaunchKernelGGL((k_issue<KIND>), dim3(256), dim3(threads), 0, 0, d, 1u);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k_issue<KIND>), dim3(256), dim3(threads), 0, 0, d, 1u);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 16 * 2);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0;
    const int nw = 256 * threads / 64;
    for (int i = 0; i < nw; ++i) sum += (double)h[i * 2];
    const double cycles = sum / nw;                        // per wave
    return cycles / ((double)ITERS * CHAINS) / waves_per_simd;  // SIMD cycles per wave-instruction
}

int main() {
    unsigned long long* d;
    hipMalloc(&d, 256 * 16 * 2 * 8);
    const char* names[24] = {"v_add_u32", "v_mad_u32_u24", "v_mul_hi_u32_u24", "v_perm_b32", "v_fma_f32", "v_lshl_add_u32", "v_cvt_f32_i32", "v_alignbit_b32", "v_add_u32_e64", "v_add_u32 literal", "v_and_b32 literal", "v_add_u32 sgpr", "v_mad_u32_u24 a,b,a", "v_pk_add_f32", "v_cndmask_b32 vcc", "v_dot2_i32_i16", "v_cndmask_b32 sgprs", "v_cmp+v_cndmask /2", "v_min_u32", "v_lshrrev_b32 imm", "v_mul_u32_u24", "v_bfe_u32", "v_add3_u32", "v_and_or_b32"};
    printf("SIMD cycles per wave64 instruction (s_memtime cycles / instructions / waves per SIMD), 1 workgroup per CU\n");
    printf("%-20s %8s %8s %8s %8s\n", "instruction", "1 w/SIMD", "2", "3", "4");
#define ROW(K) printf("%-20s %8.2f %8.2f %8.2f %8.2f\n", names[K], run<K>(1, d), run<K>(2, d), run<K>(3, d), run<K>(4, d));
    ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7) ROW(8) ROW(9) ROW(10) ROW(11) ROW(12) ROW(13) ROW(14) ROW(15) ROW(16) ROW(17) ROW(18) ROW(19) ROW(20) ROW(21) ROW(22) ROW(23)
    return 0;
}
