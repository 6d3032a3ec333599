// Development probe (GPU box): the block Gauss-Jordan dense inverse of crsolve.hip against A A^-1 = I, with timing.
#include "../../stereo_orb_slam_amd/csrc/crsolve.hip"
#include <cstdio>
#include <vector>
using namespace soslam;
int main()
{
    for (int nb : {1, 3, 12}) {
        const int n = 60 * nb;
        std::vector<double> G((size_t)n * n), A((size_t)n * n), Ai((size_t)n * n);
        unsigned long long s = 88172645463325252ull;
        auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s % 2000001) / 1e6 - 1.0; };
        for (auto& g : G) g = rnd();
        for (int i = 0; i < n; i++)
            for (int j = 0; j <= i; j++) {
                double v = i == j ? 0.05 * n : 0.0;
                for (int k = 0; k < n; k += 7) v += G[(size_t)i * n + k] * G[(size_t)j * n + k];
                A[(size_t)i * n + j] = A[(size_t)j * n + i] = v;
            }
        double *dA, *dE, *dS;
        hipMalloc(&dA, A.size() * 8); hipMalloc(&dE, 2 * 3600 * 8); hipMalloc(&dS, 8);
        hipMemset(dS, 0, 8);
        float best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            launch_dense_spd_inverse60(nullptr, dA, nb, dE, dS);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        hipMemcpy(Ai.data(), dA, A.size() * 8, hipMemcpyDeviceToHost);
        double st; hipMemcpy(&st, dS, 8, hipMemcpyDeviceToHost);
        double worst = 0;
        for (int i = 0; i < n; i += 3)
            for (int j = 0; j < n; j++) {
                double v = 0;
                for (int k = 0; k < n; k++) v += A[(size_t)i * n + k] * Ai[(size_t)k * n + j];
                worst = fmax(worst, fabs(v - (i == j ? 1.0 : 0.0)));
            }
        printf("nb %2d (n = %4d): %.3f ms, status %.0f, max |A Ainv - I| = %.3e\n", nb, n, best, st, worst);
    }
    return 0;
}
