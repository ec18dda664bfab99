// host_hash.hpp -- 64-bit hash of a sparsity pattern held in HOST arrays, computed by a few host threads at memory speed.
//
// Why: Newton factorizes a new Hessian of the SAME pattern every iteration (Newton.cpp:189-193) and the reference
// backends keep what the pattern determined across those calls (MAS its partition, MASSolver.cu:304-321).  The host
// contract hands over pattern + values every time; recognising the pattern on the host lets factorize(host arrays)
// move only the values (8 of the 12 bytes per stored entry) -- the hash runs while the values already travel.
#pragma once
#include <algorithm>
#include <cstdint>
#include <thread>
#include <vector>

namespace psolve {

inline uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// position-dependent hash of a[0..n): eight independent polynomial accumulators (the multiplies pipeline: ~1 entry per
// cycle), folded per 64 Ki-entry tile together with the tile's index
inline uint64_t hash_i32_range(const int32_t *a, int64_t begin, int64_t end)
{
    constexpr int64_t kTile = 1 << 16;
    constexpr uint64_t P = 0x9E3779B97F4A7C15ull;
    uint64_t total = 0;
    for (int64_t t0 = begin; t0 < end; t0 += kTile) {
        const int64_t t1 = std::min(end, t0 + kTile);
        uint64_t h[8] = {1, 2, 3, 4, 5, 6, 7, 8};
        int64_t i = t0;
        for (; i + 8 <= t1; i += 8)
            for (int k = 0; k < 8; ++k) h[k] = h[k] * P + (uint32_t)a[i + k];
        for (int k = 0; i < t1; ++i, ++k) h[k] = h[k] * P + (uint32_t)a[i];
        uint64_t f = (uint64_t)t0;
        for (int k = 0; k < 8; ++k) f = mix64(f ^ h[k]);
        total += f;
    }
    return total;
}

struct HostPatternHash {
    uint64_t outer = 0, inner = 0;
    bool operator==(const HostPatternHash &o) const { return outer == o.outer && inner == o.inner; }
};

// ranges are cut at tile multiples, so the result does not depend on the number of threads
inline HostPatternHash hash_host_pattern(int64_t n, int64_t nnz, const int32_t *outer, const int32_t *inner, int threads = 0)
{
    if (threads <= 0) threads = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency() / 2));
    constexpr int64_t kTile = 1 << 16;
    const int64_t tiles = (nnz + kTile - 1) / kTile;
    threads = (int)std::max<int64_t>(1, std::min<int64_t>(threads, tiles));
    std::vector<uint64_t> part((size_t)threads, 0);
    std::vector<std::thread> th;
    for (int r = 1; r < threads; ++r)
        th.emplace_back([&, r] {
            const int64_t b = std::min(nnz, tiles * r / threads * kTile), e = std::min(nnz, tiles * (r + 1) / threads * kTile);
            part[(size_t)r] = hash_i32_range(inner, b, e);
        });
    part[0] = hash_i32_range(inner, 0, std::min(nnz, tiles / threads * kTile));
    HostPatternHash h;
    h.outer = hash_i32_range(outer, 0, n + 1);
    for (auto &t : th) t.join();
    for (uint64_t v : part) h.inner += v;
    return h;
}

} // namespace psolve
