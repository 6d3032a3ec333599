// Probe of the v_mfma_f64_16x16x4_f64 operand/accumulator register layout on gfx950 (development aid).
// hipcc --offload-arch=gfx950 -O2 scripts/mfma_f64_layout.hip -o scripts/mfma_layout.bin && ./scripts/mfma_layout.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
// every lane supplies a = av[l], b = bv[l]; raw accumulator registers come back as out[l*4 + r]
__global__ void k(const double* av, const double* bv, double* out)
{
    const int l = threadIdx.x;
    double4_t c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(av[l], bv[l], c, 0, 0, 0);
    for (int r = 0; r < 4; r++) out[l * 4 + r] = c[r];
}
int main()
{
    double ha[64], hb[64], ho[256];
    double *da, *db, *dout;
    (void)hipMalloc(&da, sizeof ha); (void)hipMalloc(&db, sizeof hb); (void)hipMalloc(&dout, sizeof ho);
    // hypothesis for the inputs: lane l holds A[l%16][l/16] and B[l/16][l%16].  Tag A[i][k] = (i+1) + 100*(k+1) ... use
    // k-separable primes so each D entry identifies (i, j): A[i][k] = (i+1) * p_k, B[k][j] = (j+1) * q_k with
    // sum_k p_k q_k = 1*1 + 0 + 0 + 0 (only k = 0 active) first, then all k to check the k pairing.
    for (int pass = 0; pass < 2; pass++) {
        for (int l = 0; l < 64; l++) {
            const int i = l % 16, kk = l / 16;
            const double pk = pass == 0 ? (kk == 0 ? 1.0 : 0.0) : 1.0 + kk;
            const double qk = pass == 0 ? (kk == 0 ? 1.0 : 0.0) : 1.0 / (1.0 + kk);
            ha[l] = (i + 1) * pk;
            hb[l] = (i + 1) * 100.0 * qk;
        }
        (void)hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dout);
        (void)hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
        const double scale = pass == 0 ? 100.0 : 400.0;   // pass 1: sum_k p_k q_k = 4
        printf("pass %d: lane -> (i,j) of its 4 accumulator values [D = (i+1)(j+1)*%g]\n", pass, scale);
        for (int l = 0; l < 64; l += (l < 4 ? 1 : 15)) {
            printf("  lane %2d:", l);
            for (int r = 0; r < 4; r++) {
                const double v = ho[l * 4 + r] / scale;
                int fi = -1, fj = -1;
                for (int i = 1; i <= 16; i++) for (int j = 1; j <= 16; j++) if (fabs(v - i * j) < 1e-9) { fi = i - 1; fj = j - 1; }
                printf("  r%d=%g", r, v);
                (void)fi; (void)fj;
            }
            printf("\n");
        }
    }
    return 0;
}
