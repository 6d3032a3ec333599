// Development probe (GPU box): per-instruction constants the register-resident sweep depends on.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/probes/ubench.hip -o scripts/probes/bin/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(1024) void k_mfma(double* out, unsigned long long* t, int mode)
{
    __shared__ double lds[2048];
    const int lane = threadIdx.x % 64;
    double4_t a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    double x = 1.0 + lane * 1e-9, y = 0.5;
    lds[threadIdx.x] = x;
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter();
    if (mode == 0) {          // 4 independent accumulators, back to back
#pragma unroll 1
        for (int i = 0; i < 250; i++) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
        }
    } else if (mode == 1) {   // one accumulator: dependent chain
#pragma unroll 1
        for (int i = 0; i < 250; i++) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        }
    } else if (mode == 2) {   // mfma -> valu reads the result -> operand of the next mfma
#pragma unroll 1
        for (int i = 0; i < 1000; i++) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
            x = a0[0] * 1e-30 + 1.0;
        }
    } else if (mode == 3) {   // dependent f64 fma chain
#pragma unroll 1
        for (int i = 0; i < 250; i++) {
            x = __builtin_fma(x, y, 0.25); x = __builtin_fma(x, y, 0.25); x = __builtin_fma(x, y, 0.25); x = __builtin_fma(x, y, 0.25);
        }
    } else if (mode == 4) {   // independent f64 fmas
        double x1 = x, x2 = x + 1, x3 = x + 2, x4 = x + 3;
#pragma unroll 1
        for (int i = 0; i < 250; i++) {
            x1 = __builtin_fma(x1, y, 0.25); x2 = __builtin_fma(x2, y, 0.25); x3 = __builtin_fma(x3, y, 0.25); x4 = __builtin_fma(x4, y, 0.25);
        }
        x = x1 + x2 + x3 + x4;
    } else if (mode == 5) {   // lds write -> barrier -> read -> use (the exchange of the sweep), per round
#pragma unroll 1
        for (int i = 0; i < 1000; i++) {
            lds[(threadIdx.x + 64) % blockDim.x] = x;
            __syncthreads();
            x = lds[threadIdx.x] + 1e-9;
        }
    } else if (mode == 6) {   // barrier alone
#pragma unroll 1
        for (int i = 0; i < 1000; i++) { __syncthreads(); asm volatile("" ::: "memory"); }
    } else if (mode == 7) {   // lds read latency chain (address depends on the value read)
        int idx = threadIdx.x;
        reinterpret_cast<int*>(lds)[threadIdx.x * 2] = (threadIdx.x + 1) % blockDim.x;
        __syncthreads();
#pragma unroll 1
        for (int i = 0; i < 1000; i++) idx = reinterpret_cast<volatile int*>(lds)[idx * 2];
        x = idx;
    } else if (mode == 8) {   // rcp f64 chain
#pragma unroll 1
        for (int i = 0; i < 1000; i++) x = __builtin_amdgcn_rcp(x) + 0.5;
    } else if (mode == 9) {   // readfirstlane round trip
#pragma unroll 1
        for (int i = 0; i < 1000; i++) { const int v = __builtin_amdgcn_readfirstlane(__double2hiint(x)); x = __hiloint2double(v, __double2loint(x)) * 1.0000001; }
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    out[threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + x;
    if (threadIdx.x == 0) t[0] = c1 - c0;
}

int main()
{
    double* d; unsigned long long* t;
    hipMalloc(&d, 8 * 256); hipMalloc(&t, 8);
    const char* names[] = {"mfma f64 16x16x4, 4 independent accumulators (per instruction)", "mfma f64, one accumulator (per instruction)",
                           "mfma -> valu use -> mfma (per round)", "f64 fma dependent chain (per fma)", "f64 fma, 4 independent (per fma)",
                           "lds write -> barrier -> read -> use (per round)", "barrier alone", "lds dependent read (per read)", "v_rcp_f64 + add chain (per round)",
                           "readfirstlane round trip + mul (per round)"};
    for (int threads : {256, 512, 1024})
        for (int mode = 0; mode < 7; mode++) {
            hipLaunchKernelGGL(k_mfma, dim3(1), dim3(threads), 0, 0, d, t, mode);
            hipLaunchKernelGGL(k_mfma, dim3(1), dim3(threads), 0, 0, d, t, mode);
            unsigned long long c; hipMemcpy(&c, t, 8, hipMemcpyDeviceToHost);
            printf("%3d threads  %-70s %.1f cycles\n", threads, names[mode], c / 1000.0);
        }
    return 0;
}
