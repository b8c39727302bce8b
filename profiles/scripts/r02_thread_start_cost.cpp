// What one parallel_ranges call costs beyond its work: std::thread start + join per call.
// g++ -O2 -std=c++17 -pthread -o /tmp/tsc profiles/scripts/r02_thread_start_cost.cpp && /tmp/tsc
#include "../../score_amd/csrc/score_host.hpp"
#include <chrono>
#include <cstdio>
int main() {
    using namespace score;
    const int64_t n = 147000;
    std::vector<double> v(n * 8, 1.0);
    for (int rep = 0; rep < 3; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        std::vector<double> sums(64, 0.0);
        std::vector<double> tms(64, 0.0);
        parallel_ranges(n, 8192, [&](int t, int64_t i0, int64_t i1) {
            auto a = std::chrono::steady_clock::now();
            double s = 0;
            for (int r = 0; r < 20; ++r)
                for (int64_t i = i0 * 8; i < i1 * 8; ++i) s += v[i] * (r + 1);
            sums[t] = s;
            tms[t] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
        });
        double tot = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        std::printf("threads %d total %.2f ms; per-thread:", host_threads(), tot);
        for (int t = 0; t < host_threads(); ++t) std::printf(" %.2f", tms[t]);
        std::printf("\n");
    }
}
