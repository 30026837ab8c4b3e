// philox.h -- counter-based random bits for the attention dropout of the fused GAT kernels (gat_drop.hip).
//
// Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11): a keyed bijection
// of 128-bit counters, ten rounds of two 32x32 -> 64-bit multiplies.  Being a pure function of (key, counter) the keep
// mask of an edge can be REGENERATED wherever that edge is visited again -- the forward pass walks the CSR, the backward
// passes walk the CSR and the CSC -- so no [E,H] mask is ever stored (torch's dropout keeps one byte per element).
//
// Mask of (edge e, head h) under seed s and drop probability p:
//   block b = h / 8:  (w0,w1,w2,w3) = philox4x32_10(counter = (e, 0, b, 0), key = (s_lo, s_hi))
//   j = h % 8:        u16 = (w[j / 2] >> (16 * (j % 2))) & 0xffff
//   keep  <=>  u16 >= thresh,  thresh = round(p * 65536)          (P[keep] = 1 - thresh / 65536 exactly)
//   d[e,h] = keep ? 65536 / (65536 - thresh) : 0                   (unbiased: E[d] = 1; p = 0.5 -> 2.0 as in torch)
// e is the edge's position in the CSR order of the forward graph (the index of its row in the [E,H] attention tensor of
// the unfused path, cogdl/layers/gat_layer.py:73-75).
#pragma once
#include <stdint.h>

#include <hip/hip_runtime.h>

namespace cogdl {

struct Philox4 {
    uint32_t x, y, z, w;
};

__host__ __device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                          uint32_t k0, uint32_t k1) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += W0;
        k1 += W1;
    }
    return {c0, c1, c2, c3};
}

// Bit h of the result = "attention of (edge, head h) is kept".  Bits at and beyond `heads` (<= 64) are unspecified.
__host__ __device__ __forceinline__ uint64_t drop_keep_bits(uint32_t seed_lo, uint32_t seed_hi, uint32_t edge,
                                                            int heads, uint32_t thresh) {
    uint64_t bits = 0;
    for (int b = 0; b * 8 < heads; ++b) {
        const Philox4 r = philox4x32_10(edge, 0u, (uint32_t)b, 0u, seed_lo, seed_hi);
        const uint32_t w[4] = {r.x, r.y, r.z, r.w};
        uint32_t byte = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) byte |= (((w[j >> 1] >> (16 * (j & 1))) & 0xffffu) >= thresh ? 1u : 0u) << j;
        bits |= (uint64_t)byte << (8 * b);
    }
    return bits;
}

// (drop probability) -> (threshold, scale of the kept values)
inline void drop_params(float p, uint32_t &thresh, float &scale) {
    float t = p * 65536.f;
    t = t < 0.f ? 0.f : (t > 65536.f ? 65536.f : t);
    thresh = (uint32_t)(t + 0.5f);
    scale = thresh >= 65536u ? 0.f : 65536.f / (float)(65536u - thresh);
}

}  // namespace cogdl
