// Counter-based random numbers on the device: Philox4x32-10 (Salmon et al., SC'11), key = the caller's 64-bit seed, counter = (item index, draw index):
// a draw depends on (seed, item, draw) only - not on the launch geometry, not on which rank or lane produces it.  Used by the SPD sampler
// (spd_sample.hip) and by the sweep's selection kernel (spd_sweep.hip; a different constant in the fourth counter word keeps the two streams apart).
#pragma once
#include "gabo_device.hpp"

namespace gabo {

struct Philox {
    uint32_t k0, k1;
    uint64_t idx;
    uint32_t draw;
    uint32_t tag = 0x6761626fu;      // fourth counter word: which stream of the library
    __device__ __forceinline__ void next(uint32_t (&o)[4]) {
        uint32_t c0 = (uint32_t)idx, c1 = (uint32_t)(idx >> 32), c2 = draw++, c3 = tag;
        uint32_t a = k0, b = k1;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
            const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ a, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ b;
            c1 = (uint32_t)p1;
            c3 = (uint32_t)p0;
            c0 = n0;
            c2 = n2;
            a += 0x9E3779B9u;
            b += 0xBB67AE85u;
        }
        o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
    }
    // two uniforms: u1 in (0, 1], u2 in [0, 1), 53 bits each
    __device__ __forceinline__ void uniform2(double& u1, double& u2) {
        uint32_t o[4];
        next(o);
        const uint64_t a = (((uint64_t)o[0] << 32) | o[1]) >> 11, b = (((uint64_t)o[2] << 32) | o[3]) >> 11;
        u1 = ((double)a + 1.0) * 0x1.0p-53;
        u2 = (double)b * 0x1.0p-53;
    }
    __device__ __forceinline__ void normal2(double& z0, double& z1) {
        double u1, u2;
        uniform2(u1, u2);
        const double r = __builtin_sqrt(-2.0 * log(u1));
        double s, c;
        sincospi(2.0 * u2, &s, &c);
        z0 = r * c;
        z1 = r * s;
    }
};

}  // namespace gabo
