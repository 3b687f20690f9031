// philox.h -- counter-based Philox4x32-10 draws for the device RNG mode.
//
// The reference draws from R's global stream (norm_rand()/unif_rand(), metrop.c:232,378;
// temper.c:537-996).  A sequential stream cannot feed thousands of lockstep chains, so the
// device mode addresses every draw by (key = seed ^ sampler tag; counter = iteration,
// global chain id, draw block): results do not depend on how chains are sharded over GPUs,
// and a restart with iter_offset continues the same sequence (tests/saveseed.R contract).
//   metrop: block b < 2^31 -> normals z[2b], z[2b+1] (Box-Muller); block 0xFFFFFFFF -> accept uniform
//   temper: block 0 -> (u_which, u_choose1), block 1 -> (u_choose2, u_hastings), block 2+b -> normals
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PHILOX_TAG_METROP 0u
#define PHILOX_TAG_TEMPER 0x54454d50u
#define PHILOX_BLOCK_UNIF 0xFFFFFFFFu

struct Philox4 { uint32_t v[4]; };

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                 uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint32_t lo0 = 0xD2511F53u * c0, hi0 = __umulhi(0xD2511F53u, c0);
        uint32_t lo1 = 0xCD9E8D57u * c2, hi1 = __umulhi(0xCD9E8D57u, c2);
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    Philox4 o;
    o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}

// 53-bit uniform strictly inside (0,1)
__device__ __forceinline__ double philox_u53(uint32_t a, uint32_t b)
{
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6) + 0.5) * (1.0 / 9007199254740992.0);
}

__device__ __forceinline__ void philox_u2(uint64_t seed, uint32_t tag, uint64_t chain, uint64_t iter,
                                          uint32_t block, double &ua, double &ub)
{
    Philox4 p = philox4x32_10((uint32_t)iter, (uint32_t)(iter >> 32), (uint32_t)chain, block,
                              (uint32_t)seed, (uint32_t)(seed >> 32) ^ tag);
    ua = philox_u53(p.v[0], p.v[1]);
    ub = philox_u53(p.v[2], p.v[3]);
}

__device__ __forceinline__ void philox_n2(uint64_t seed, uint32_t tag, uint64_t chain, uint64_t iter,
                                          uint32_t block, double &za, double &zb)
{
    double u1, u2;
    philox_u2(seed, tag, chain, iter, block, u1, u2);
    double r = sqrt(-2.0 * log(u1));
    double t = 6.283185307179586 * u2;
    double s, c;
    sincos(t, &s, &c);
    za = r * c;
    zb = r * s;
}
