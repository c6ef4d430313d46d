// Host-side check of the in-register FFT building blocks (fft_regs.hpp) against a naive DFT.
// Built and run by tests/test_host_logic.py::test_fft_regs_host (no GPU needed: the templates are
// __host__ __device__ and the packed float pairs are clang vector extensions).
#include "../../fpga_real_time_fft_analyzer_amd/csrc/fft_regs.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>

using safft::cf;

template <int R, int BITS>
static double check()
{
    cf a[R];
    double xr[R], xi[R];
    for (int n = 0; n < R; ++n) {
        xr[n] = std::sin(0.37 * n + 0.1) + 0.25 * n;
        xi[n] = std::cos(1.3 * n) - 0.5;
        a[safft::brev(n, BITS)] = cf{(float)xr[n], (float)xi[n]};
    }
    safft::fft_dit<R>(a);
    double worst = 0, scale = 0;
    for (int k = 0; k < R; ++k) {
        double sr = 0, si = 0;
        for (int n = 0; n < R; ++n) {
            const double ang = -2.0 * M_PI * n * k / R;
            sr += xr[n] * std::cos(ang) - xi[n] * std::sin(ang);
            si += xr[n] * std::sin(ang) + xi[n] * std::cos(ang);
        }
        worst = std::fmax(worst, std::hypot(a[k].x - sr, a[k].y - si));
        scale = std::fmax(scale, std::hypot(sr, si));
    }
    return worst / scale;
}

int main()
{
    const double e4 = check<4, 2>(), e8 = check<8, 3>(), e16 = check<16, 4>(), e32 = check<32, 5>();
    const cf p = safft::cmul(cf{1.5f, -2.0f}, cf{0.6f, 0.8f});          // (1.5-2i)(0.6+0.8i) = 2.5 + 0i
    const double ec = std::hypot(p.x - 2.5, p.y - 0.0);
    std::printf("fft4 %.3e fft8 %.3e fft16 %.3e fft32 %.3e cmul %.3e\n", e4, e8, e16, e32, ec);
    return (e4 < 5e-7 && e8 < 5e-7 && e16 < 5e-7 && e32 < 5e-7 && ec < 1e-6) ? 0 : 1;
}
