// solve_mm.cpp -- a C++ host of the C ABI with no third-party dependency: reads a Matrix Market file
// the way the reference's test helper does (loadSymmetric, /root/reference/tests/test_linear_solver.cpp:
// 25-50: skip '%' lines, "M N L", then L triplets mirrored across the diagonal), solves A x = b with
// b = 1 like the reference's gr_30_30 / crystm03 tests (:551-553, :615), and prints what get_info reports.
//
//   g++ -O2 -std=c++17 -I include examples/solve_mm.cpp -o examples/solve_mm \
//       polysolve_amd/lib/libpsolve_hip.so -Wl,-rpath,'$ORIGIN/../polysolve_amd/lib'
//   examples/solve_mm matrix.mtx [general|symmetric] [jacobi|none|amg] [block_size]
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <tuple>
#include <vector>

#include "psolve_hip.h"

static void die(psolve_hip_t h, const char *what, int rc)
{
    std::fprintf(stderr, "%s failed (%d): %s\n", what, rc, psolve_hip_last_error(h));
    std::exit(2);
}

int main(int argc, char **argv)
{
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s matrix.mtx [symmetric|general] [jacobi|none|amg] [block_size]\n", argv[0]);
        return 1;
    }
    const bool symmetric = argc < 3 || std::string(argv[2]) == "symmetric";
    const std::string precond = argc > 3 ? argv[3] : "jacobi";
    const int block_size = argc > 4 ? std::atoi(argv[4]) : 1;

    std::ifstream fin(argv[1]);
    if (!fin) {
        std::fprintf(stderr, "cannot open %s\n", argv[1]);
        return 1;
    }
    while (fin.peek() == '%') fin.ignore(1 << 20, '\n');
    long M = 0, N = 0, L = 0;
    fin >> M >> N >> L;
    std::vector<std::tuple<int, int, double>> trip;
    trip.reserve((size_t)L * 2);
    for (long i = 0; i < L; ++i) {
        int m, n;
        double v;
        fin >> m >> n >> v;
        trip.emplace_back(m - 1, n - 1, v);
        if (symmetric && m != n) trip.emplace_back(n - 1, m - 1, v);
    }
    // triplets -> compressed arrays with summed duplicates (Eigen's setFromTriplets)
    std::sort(trip.begin(), trip.end());
    std::vector<int32_t> outer((size_t)M + 1, 0), inner;
    std::vector<double> values;
    for (size_t k = 0; k < trip.size(); ++k) {
        const auto [r, c, v] = trip[k];
        if (k > 0 && std::get<0>(trip[k - 1]) == r && std::get<1>(trip[k - 1]) == c) {
            values.back() += v;
            continue;
        }
        inner.push_back(c);
        values.push_back(v);
        ++outer[(size_t)r + 1];
    }
    for (long r = 0; r < M; ++r) outer[(size_t)r + 1] += outer[(size_t)r];
    const int64_t n = M, nnz = (int64_t)inner.size();

    psolve_hip_t h = nullptr;
    int rc = psolve_hip_create(&h, 0);
    if (rc) die(nullptr, "psolve_hip_create", rc);
    psolve_hip_set_param(h, "tolerance", 1e-10); // AMGCL.cpp:58 default of the reference's runs
    psolve_hip_set_param(h, "precond", precond == "none" ? 0 : precond == "amg" ? 2 : 1);
    psolve_hip_set_param(h, "block_size", block_size);
    if ((rc = psolve_hip_analyze_pattern(h, n, nnz, outer.data(), inner.data(), (int)n))) die(h, "analyze_pattern", rc);
    if ((rc = psolve_hip_factorize(h, n, nnz, outer.data(), inner.data(), values.data()))) die(h, "factorize", rc);
    std::vector<double> b((size_t)n, 1.0), x((size_t)n, 0.0);
    if ((rc = psolve_hip_solve(h, b.data(), x.data()))) die(h, "solve", rc);
    psolve_hip_info info;
    psolve_hip_get_info(h, &info);
    // ||A x - b|| / ||b|| on the host, as the reference test does (:600-601)
    double rr = 0, bb = 0;
    for (int64_t r = 0; r < n; ++r) {
        double s = -b[(size_t)r];
        for (int32_t j = outer[(size_t)r]; j < outer[(size_t)r + 1]; ++j) s += values[(size_t)j] * x[(size_t)inner[(size_t)j]];
        rr += s * s;
        bb += b[(size_t)r] * b[(size_t)r];
    }
    double renumbered = 0.0, spread = 0.0; // "reorder" (auto): was the file's numbering scattered enough to be renumbered?
    psolve_hip_get_param(h, "reorder.active", &renumbered);
    psolve_hip_get_param(h, "reorder.spread_before", &spread);
    std::printf("n=%lld nnz=%lld precond=%s block_size=%d num_iterations=%lld final_res_norm=%.3e host_residual=%.3e "
                "factorize_s=%.4f solve_s=%.4f status=%d renumbered=%d gather_spread=%.2f\n",
                (long long)n, (long long)nnz, precond.c_str(), block_size, (long long)info.num_iterations,
                info.final_res_norm, std::sqrt(rr / bb), info.time_factorize, info.time_solve, info.solver_status,
                (int)renumbered, spread);
    psolve_hip_destroy(h);
    return std::sqrt(rr / bb) < 1e-7 ? 0 : 3;
}
