// hobbit_blake3.hpp -- BLAKE3 of exactly 64 bytes -> 32 bytes, as one register-resident
// compression (the only BLAKE3 shape on the reference's hot path: src/Blake3_hash.cpp:5-10 =
// blake3_hasher_init/update(64)/finalize(32) = compress(IV, block, counter 0, block_len 64,
// flags CHUNK_START|CHUNK_END|ROOT), Blake/blake3.c:146-151,598-617).  Written from the BLAKE3
// specification; the message schedule is resolved at compile time so no word ever moves.
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HB3_HD __host__ __device__ __forceinline__
#else
#define HB3_HD inline
#endif

namespace hobbit {

#define HB3_ROTR(x, c) (((x) >> (c)) | ((x) << (32 - (c))))
// rotr(d ^ a, 16).  On the device: two half-word-select XORs (SDWA, 2-source VALU rate) instead of an XOR plus a v_alignbit_b32,
// which issues at half rate on this chip (profiles/r01_microbench.txt) -- the leaf chain runs at the VALU issue limit.
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ __forceinline__ uint32_t hb3_xor_rot16(uint32_t d, uint32_t a) {
    uint32_t t;
    asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n\t"
        "v_xor_b32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0" : "=&v"(t) : "v"(d), "v"(a));
    return t;
}
#define HB3_XROT16(d, a) hb3_xor_rot16(d, a)
#else
#define HB3_XROT16(d, a) HB3_ROTR((d) ^ (a), 16)
#endif
#define HB3_G(a, b, c, d, mx, my)                                              \
    do {                                                                       \
        a = a + b + (mx); d = HB3_XROT16(d, a); c = c + d; b = HB3_ROTR(b ^ c, 12); \
        a = a + b + (my); d = HB3_ROTR(d ^ a, 8);  c = c + d; b = HB3_ROTR(b ^ c, 7);  \
    } while (0)
// one round with message words given in schedule order
#define HB3_ROUND(m0, m1, m2, m3, m4, m5, m6, m7, m8, m9, m10, m11, m12, m13, m14, m15) \
    do {                                                                       \
        HB3_G(v0, v4, v8, v12, m0, m1);  HB3_G(v1, v5, v9, v13, m2, m3);       \
        HB3_G(v2, v6, v10, v14, m4, m5); HB3_G(v3, v7, v11, v15, m6, m7);      \
        HB3_G(v0, v5, v10, v15, m8, m9); HB3_G(v1, v6, v11, v12, m10, m11);    \
        HB3_G(v2, v7, v8, v13, m12, m13); HB3_G(v3, v4, v9, v14, m14, m15);    \
    } while (0)

// m[16] little-endian message words -> out[8]
HB3_HD void blake3_compress64(const uint32_t m[16], uint32_t out[8]) {
    uint32_t v0 = 0x6A09E667u, v1 = 0xBB67AE85u, v2 = 0x3C6EF372u, v3 = 0xA54FF53Au;
    uint32_t v4 = 0x510E527Fu, v5 = 0x9B05688Cu, v6 = 0x1F83D9ABu, v7 = 0x5BE0CD19u;
    uint32_t v8 = 0x6A09E667u, v9 = 0xBB67AE85u, v10 = 0x3C6EF372u, v11 = 0xA54FF53Au;
    uint32_t v12 = 0, v13 = 0, v14 = 64, v15 = 1u | 2u | 8u;
    // schedule: round r uses m[perm^r(i)], perm = {2,6,3,10,7,0,4,13,1,11,12,5,9,14,15,8}
    HB3_ROUND(m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], m[8], m[9], m[10], m[11], m[12], m[13], m[14], m[15]);
    HB3_ROUND(m[2], m[6], m[3], m[10], m[7], m[0], m[4], m[13], m[1], m[11], m[12], m[5], m[9], m[14], m[15], m[8]);
    HB3_ROUND(m[3], m[4], m[10], m[12], m[13], m[2], m[7], m[14], m[6], m[5], m[9], m[0], m[11], m[15], m[8], m[1]);
    HB3_ROUND(m[10], m[7], m[12], m[9], m[14], m[3], m[13], m[15], m[4], m[0], m[11], m[2], m[5], m[8], m[1], m[6]);
    HB3_ROUND(m[12], m[13], m[9], m[11], m[15], m[10], m[14], m[8], m[7], m[2], m[5], m[3], m[0], m[1], m[6], m[4]);
    HB3_ROUND(m[9], m[14], m[11], m[5], m[8], m[12], m[15], m[1], m[13], m[3], m[0], m[10], m[2], m[6], m[4], m[7]);
    HB3_ROUND(m[11], m[15], m[5], m[0], m[1], m[9], m[8], m[6], m[14], m[10], m[2], m[12], m[3], m[4], m[7], m[13]);
    out[0] = v0 ^ v8;  out[1] = v1 ^ v9;  out[2] = v2 ^ v10; out[3] = v3 ^ v11;
    out[4] = v4 ^ v12; out[5] = v5 ^ v13; out[6] = v6 ^ v14; out[7] = v7 ^ v15;
}

}  // namespace hobbit
