// cmf_rng.h -- the product's portable counter-based RNG (host side).
//
// Julia's `rand` (MersenneTwister at the reference's Julia versions) is not reproducible across
// Julia releases, so init_rand (src/model.jl:113-125) and gen_synthetic
// (datasets/synthetic.jl:29-61) draw from this splitmix64-style counter generator instead:
//     f(z): z ^= z>>30; z *= 0xBF58476D1CE4E5B9; z ^= z>>27; z *= 0x94D049BB133111EB; z ^= z>>31
//     base(seed, stream) = f(seed + 0x632BE59BD9B4E019 * (stream + 1))
//     bits(base, i)      = f(base + (i + 1) * 0x9E3779B97F4A7C15)
//     u01   = (bits >> 11) * 2^-53  in [0,1);   u01o = ((bits >> 11) + 1) * 2^-53  in (0,1]
//     normal(i) = sqrt(-2 ln u01o_a(i)) * cos(2 pi u01_b(i))   (streams a, b)
// Every value is a pure function of (seed, stream, index), so generation parallelises freely.
#pragma once
#include <cmath>
#include <cstdint>

namespace cmfrng {

inline uint64_t mix(uint64_t z)
{
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27; z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return z;
}
inline uint64_t base(uint64_t seed, uint64_t stream) { return mix(seed + 0x632BE59BD9B4E019ULL * (stream + 1)); }
inline uint64_t bits(uint64_t b, uint64_t i) { return mix(b + (i + 1) * 0x9E3779B97F4A7C15ULL); }
inline double u01(uint64_t b, uint64_t i) { return (double)(bits(b, i) >> 11) * 0x1.0p-53; }
inline double u01o(uint64_t b, uint64_t i) { return (double)((bits(b, i) >> 11) + 1) * 0x1.0p-53; }
inline double normal(uint64_t ba, uint64_t bb, uint64_t i)
{
    return std::sqrt(-2.0 * std::log(u01o(ba, i))) * std::cos(6.283185307179586476925286766559 * u01(bb, i));
}
// Gamma(a, 1): Marsaglia-Tsang, with the a < 1 boost.  Sample j, attempt t draws from counters
// 4*(64*j + t) + {0, 1}; the boost uniform is counter 4*64*j + 3.
inline double gamma(uint64_t bna, uint64_t bnb, uint64_t bu, uint64_t j, double a)
{
    double boost = 1.0;
    if (a < 1.0) {
        boost = std::pow(u01o(bu, 4 * (64 * j) + 3), 1.0 / a);
        a += 1.0;
    }
    const double d = a - 1.0 / 3.0, c = 1.0 / std::sqrt(9.0 * d);
    for (uint64_t t = 0; t < 64; ++t) {
        const uint64_t ctr = 4 * (64 * j + t);
        const double x = normal(bna, bnb, ctr);
        double v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        const double u = u01o(bu, ctr + 1);
        if (std::log(u) < 0.5 * x * x + d - d * v + d * std::log(v)) return boost * d * v;
    }
    return boost * d;
}

} // namespace cmfrng
