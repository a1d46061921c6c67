// Microbenchmark: what a v_mfma_i32_16x16x64_i8 costs the issuing wave on gfx950, alone and between VALU instructions,
// with independent and with chained accumulators, at 1 and 2 waves per SIMD (one workgroup per CU; cycles from s_memtime).
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_issue.hip -o mfma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
#define ITERS 2048

// KIND 0: 8 independent accumulators, C = previous value of the same accumulator (8 chains round robin)
// KIND 1: one chain (every MFMA depends on the one before)
// KIND 2: 8 chains, 4 VALU (v_xor) between MFMAs
// KIND 3: 8 chains, 8 VALU between MFMAs
// KIND 4: VALU only: the 4 v_xor of KIND 2 without the MFMA (baseline)
// KIND 5: 2 chains round robin
// KIND 6: 4 chains round robin
template <int KIND>
__global__ void __launch_bounds__(512) k_mfma(unsigned long long* out, unsigned seed) {
    v4i acc[8], a, b;
    unsigned x[8];
    a = v4i{(int)seed, 1, 2, 3}, b = v4i{(int)threadIdx.x, 5, 6, 7};
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = v4i{i, i, i, i}, x[i] = seed + i;
    unsigned long long t0, t1;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < ITERS; ++it) {
        asm volatile("" : "+v"(a), "+v"(b));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = KIND == 1 ? 0 : (KIND == 5 ? i & 1 : (KIND == 6 ? i & 3 : i));
            // the builtin, not inline assembly: the compiler then inserts the wait states a dependent MFMA needs
            if (KIND != 4) acc[c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[c], 0, 0, 0);
            if (KIND == 2 || KIND == 3 || KIND == 4) {
#pragma unroll
                for (int k = 0; k < (KIND == 3 ? 8 : 4); ++k) asm volatile("v_xor_b32 %0, 0x80808080, %0" : "+v"(x[(i + k) & 7]));
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3] + x[i];
    if (threadIdx.x % 64 == 0) out[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2] = t1 - t0, out[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2 + 1] = s;
}

template <int KIND>
double run(int waves_per_simd, unsigned long long* d) {
    const int threads = waves_per_simd * 4 * 64;
    for (int r = 0; r < 2; ++r) {
        hipLaunchKernelGGL((k_mfma<KIND>), dim3(256), dim3(threads), 0, 0, d, 1u);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(256 * 8 * 2);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0;
    const int nw = 256 * threads / 64;
    for (int i = 0; i < nw; ++i) sum += (double)h[i * 2];
    return sum / nw / ((double)ITERS * 8);  // wave cycles per loop step (one MFMA + its VALU)
}

int main() {
    unsigned long long* d;
    hipMalloc(&d, 256 * 8 * 2 * 8);
    printf("wave cycles per step (s_memtime), v_mfma_i32_16x16x64_i8, one workgroup per CU\n");
    printf("%-44s %10s %10s\n", "step", "1 w/SIMD", "2 w/SIMD");
    printf("%-44s %10.2f %10.2f\n", "MFMA, 8 chains round robin", run<0>(1, d), run<0>(2, d));
    printf("%-44s %10.2f %10.2f\n", "MFMA, 4 chains round robin", run<6>(1, d), run<6>(2, d));
    printf("%-44s %10.2f %10.2f\n", "MFMA, 2 chains round robin", run<5>(1, d), run<5>(2, d));
    printf("%-44s %10.2f %10.2f\n", "MFMA, one chain", run<1>(1, d), run<1>(2, d));
    printf("%-44s %10.2f %10.2f\n", "MFMA + 4 v_xor, 8 chains", run<2>(1, d), run<2>(2, d));
    printf("%-44s %10.2f %10.2f\n", "MFMA + 8 v_xor, 8 chains", run<3>(1, d), run<3>(2, d));
    printf("%-44s %10.2f %10.2f\n", "4 v_xor alone", run<4>(1, d), run<4>(2, d));
    return 0;
}
