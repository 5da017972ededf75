// hobbit_field.hpp -- F_{p^2} arithmetic, p = 2^61-1, i^2 = -1, for gfx950 device code and the
// host-side transcript/orchestration code of libhobbit_hip.so.
//
// Semantics follow the reference's virgo::fieldElement (src/fieldElement.cpp:34-96, 336-360):
// 16-byte elements {real, img}, canonical outputs 0 <= x < p for canonical inputs.  Because + and *
// are exact ring operations with canonical results, any evaluation order is bit-identical to the
// reference's sequential loops -- that is what lets the kernels use tree /
// wavefront reductions and lazy 128-bit accumulation.
//
// gfx950 notes: a 64x64->128 product lowers to 4 v_mad_u64_u32; one F_{p^2} product is 3 such
// products (Karatsuba) folded with two Mersenne reductions = 12 v_mad_u64_u32 + ~60 VALU.  A
// product with a 32-bit real scalar (the expander weights, src/expanders.h:37) is 4 v_mad_u64_u32.
#pragma once
#include <vector>
#include <cstddef>
#include <stdint.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HB_HD __host__ __device__ __forceinline__
#else
#define HB_HD inline
#endif

namespace hobbit {

typedef unsigned __int128 u128;
static constexpr uint64_t P61 = 2305843009213693951ULL;

struct __attribute__((aligned(16))) F {
    uint64_t re, im;
};
// Host view of the same element: the C ABI's hobbit_F (and the reference's virgo::fieldElement) is 8-byte aligned, so every access
// through a HOST pointer that came in over the boundary goes through this type.  It is a DISTINCT type, not an under-aligned typedef of F:
// with a typedef an HF lvalue still binds directly to `const F &` (fadd(cF(p)[i], ..), vector<F>::push_back(cF(p)[i]), v[t] = cF(p)[t]) and
// the callee is then free to load it with a 16-byte ALIGNED move -- round 2's crash, and round 3's (tests/test_gpu_parity.py::
// test_abi_host_pointers_8_mod_16 found vector<F>::push_back of such an lvalue in recursive_prover_RS).  Here every use goes through the
// conversion below: two 8-byte loads / stores, whatever the address.
struct HF {
    uint64_t re, im;
    HB_HD operator F() const { F r; r.re = re; r.im = im; return r; }
    HB_HD HF &operator=(const F &v) { re = v.re; im = v.im; return *this; }
};
static_assert(sizeof(HF) == 16 && alignof(HF) == 8, "HF must mirror hobbit_F");
// Pointer to host field elements (or to device ones, when it is only handed on): constructed from F * (library-owned, 16-byte aligned
// arrays) and from HF * (caller-owned, 8-byte aligned) alike; indexing and dereferencing always go through HF.  The parameter type of every
// internal function that reads or writes host field elements.
struct MHP;
struct CHP {
    const HF *p;
    CHP() : p(nullptr) {}
    CHP(std::nullptr_t) : p(nullptr) {}
    CHP(const HF *q) : p(q) {}
    CHP(const F *q) : p(reinterpret_cast<const HF *>(q)) {}
    inline CHP(const MHP &m);
    operator const F *() const { return reinterpret_cast<const F *>(p); }
    explicit operator bool() const { return p != nullptr; }
    const HF &operator[](size_t i) const { return p[i]; }
    const HF &operator*() const { return *p; }
    CHP operator+(ptrdiff_t k) const { return CHP(p + k); }
    const void *raw() const { return p; }
};
struct MHP {
    HF *p;
    MHP() : p(nullptr) {}
    MHP(std::nullptr_t) : p(nullptr) {}
    MHP(HF *q) : p(q) {}
    MHP(F *q) : p(reinterpret_cast<HF *>(q)) {}
    operator F *() const { return reinterpret_cast<F *>(p); }
    operator const F *() const { return reinterpret_cast<const F *>(p); }
    explicit operator bool() const { return p != nullptr; }
    HF &operator[](size_t i) const { return p[i]; }
    HF &operator*() const { return *p; }
    MHP operator+(ptrdiff_t k) const { return MHP(p + k); }
    void *raw() const { return p; }
};
inline CHP::CHP(const MHP &m) : p(m.p) {}

HB_HD F fmake(uint64_t re, uint64_t im = 0) { F r; r.re = re; r.im = im; return r; }
HB_HD bool fis0(const F &a) { return (a.re | a.im) == 0; }
HB_HD bool feq(const F &a, const F &b) { return a.re == b.re && a.im == b.im; }

HB_HD uint64_t addp(uint64_t a, uint64_t b) { uint64_t s = a + b; return s >= P61 ? s - P61 : s; }
HB_HD uint64_t subp(uint64_t a, uint64_t b) { return a >= b ? a - b : a + P61 - b; }
// x < 2^124 -> canonical
HB_HD uint64_t red124(u128 x) {
    uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
    uint64_t s = (lo & P61) + ((hi << 3) | (lo >> 61));   // x>>61 < 2^63: no overflow
    uint64_t t = (s & P61) + (s >> 61);
    return t >= P61 ? t - P61 : t;
}
// canonical a,b -> canonical a*b
HB_HD uint64_t mulp(uint64_t a, uint64_t b) {
    u128 x = (u128)a * b;
    uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
    uint64_t s = (lo & P61) + ((hi << 3) | (lo >> 61));
    return s >= P61 ? s - P61 : s;
}
// canonical a (< 2^61) times a 32-bit scalar
HB_HD uint64_t mulp32(uint64_t a, uint32_t w) {
    u128 x = (u128)a * (uint64_t)w;                        // < 2^93
    uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
    uint64_t s = (lo & P61) + ((hi << 3) | (lo >> 61));   // < 2^61 + 2^32
    return s >= P61 ? s - P61 : s;
}

HB_HD F fadd(const F &a, const F &b) { return fmake(addp(a.re, b.re), addp(a.im, b.im)); }
HB_HD F fsub(const F &a, const F &b) { return fmake(subp(a.re, b.re), subp(a.im, b.im)); }
HB_HD F fneg(const F &a) { return fmake(a.re ? P61 - a.re : 0, a.im ? P61 - a.im : 0); }
// (a+bi)(c+di) = (ac - bd) + ((a+b)(c+d) - ac - bd) i, all three products kept 128-bit wide and
// folded once per component.  C = p*2^61 >= (p-1)^2 keeps ac + C - bd non-negative.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HOBBIT_FMUL_U128)
// Device form: no 128-bit value anywhere.  Split x = x0 + 2^31 x1 (x0 < 2^31, x1 <= 2^31 - 1).  With 2^61 = 1 and 2^62 = 2 (mod p)
//     x y = x0 y0 + 2 x1 y1 + 2^31 (x0 y1 + x1 y0),
// and a complex product is a sum of two such products per component (the real part takes -a.im as 2p - a.im), so each component is
//     L = sum of four 32 x 32 products (< 2^64: v_mad_u64_u32 chains that never carry),  M = sum of four (< 2^64 likewise),
//     L + 2^31 M = L + ((M mod 2^30) << 31) + (M >> 30)      (2^61 = 1 again),      one fold61 -> [0, p + 7].
// 16 + 2 v_mad_u64_u32 and ~30 32-bit operations instead of 12 multiply-adds and ~70 operations of 128-bit glue (two 128-bit
// subtractions with an offset, two canonical reductions of 124-bit values).  Inputs: every component <= p + 7 (canonical, or the lazy range
// of the FFT octets).  Host-checked against 128-bit arithmetic on 10^7 products with the bounds at their edges.
HB_HD uint64_t limb_mad(uint32_t a, uint32_t b, uint64_t c) { return (uint64_t)a * b + c; }
HB_HD uint64_t limb_finish(uint64_t L, uint64_t M) {
    const uint64_t t = limb_mad((uint32_t)M & 0x3FFFFFFFu, 0x80000000u, L) + (M >> 30);
    return (t & P61) + (t >> 61);
}
HB_HD F fmul_lazy(const F &a, const F &b) {                  // components <= p + 7 in and out
    const uint64_t nai = 2 * P61 - a.im;
    const uint32_t ar0 = (uint32_t)a.re & 0x7FFFFFFFu, ar1 = (uint32_t)(a.re >> 31), ai0 = (uint32_t)a.im & 0x7FFFFFFFu, ai1 = (uint32_t)(a.im >> 31);
    const uint32_t na0 = (uint32_t)nai & 0x7FFFFFFFu, na1 = (uint32_t)(nai >> 31);
    const uint32_t br0 = (uint32_t)b.re & 0x7FFFFFFFu, br1 = (uint32_t)(b.re >> 31), bi0 = (uint32_t)b.im & 0x7FFFFFFFu, bi1 = (uint32_t)(b.im >> 31);
    const uint32_t br1d = br1 << 1, bi1d = bi1 << 1;
    const uint64_t Lr = limb_mad(na1, bi1d, limb_mad(na0, bi0, limb_mad(ar1, br1d, limb_mad(ar0, br0, 0))));
    const uint64_t Mr = limb_mad(na1, bi0, limb_mad(na0, bi1, limb_mad(ar1, br0, limb_mad(ar0, br1, 0))));
    const uint64_t Li = limb_mad(ai1, br1d, limb_mad(ai0, br0, limb_mad(ar1, bi1d, limb_mad(ar0, bi0, 0))));
    const uint64_t Mi = limb_mad(ai1, br0, limb_mad(ai0, br1, limb_mad(ar1, bi0, limb_mad(ar0, bi1, 0))));
    return fmake(limb_finish(Lr, Mr), limb_finish(Li, Mi));
}
HB_HD F fmul(const F &a, const F &b) {
    const F r = fmul_lazy(a, b);
    return fmake(r.re >= P61 ? r.re - P61 : r.re, r.im >= P61 ? r.im - P61 : r.im);
}
#else
HB_HD F fmul(const F &a, const F &b) {
    u128 ac = (u128)a.re * b.re, bd = (u128)a.im * b.im;
    u128 all = (u128)(a.re + a.im) * (b.re + b.im);        // operands < 2^62
    const u128 C = ((u128)P61) << 61;
    return fmake(red124(ac + C - bd), red124(all - ac - bd));
}
#endif
HB_HD F fsqr(const F &a) { return fmul(a, a); }
HB_HD F fmul32(const F &a, uint32_t w) { return fmake(mulp32(a.re, w), mulp32(a.im, w)); }
// multiply by i: (a+bi) i = -b + a i
HB_HD F fmul_i(const F &a) { return fmake(a.im ? P61 - a.im : 0, a.re); }

inline F fpow(F x, u128 e) { F r = fmake(1); while (e) { if (e & 1) r = fmul(r, x); x = fmul(x, x); e >>= 1; } return r; }
inline F finv(const F &x) { return fpow(x, (u128)P61 * P61 - 2); }              // src/fieldElement.cpp:206-209
// order-2^logn root of unity (src/utils.cpp:452-463 / src/fieldElement.cpp:237-249)
inline F root_of_unity(int logn) { F r = fmake(2147483648ULL, 1033321771269002680ULL); for (int i = 0; i < 62 - logn; i++) r = fmul(r, r); return r; }

// MiMC transcript hash (src/mimc.cpp:95-107; constants Common[i] = F(i), src/mimc.cpp:11-19):
// 161 rounds t <- (h + k + c_{i-1})^3 (round 0: t = x + k), result h + k.  Strictly sequential.
HB_HD F mimc_hash_plain(const F &x, const F &k) {
    F h = fmake(0), t;
    for (int i = 0; i < 161; i++) {
        t = i == 0 ? fadd(x, k) : fadd(fadd(h, k), fmake((uint64_t)(i - 1)));
        h = fmul(fmul(t, t), t);
    }
    return fadd(h, k);
}
// Transcript recorder (test support, include/hobbit_hip.h hobbit_transcript_*): while it is on for the calling thread every transcript hash
// the library computes on that thread appends (x, k, result), 6 words -- the complete Fiat-Shamir transcript of whatever prover runs, in order.
struct TranscriptRec { bool on = false; std::vector<uint64_t> w; };
TranscriptRec &transcript_rec();                                                                    // hobbit_capi.hip (thread_local)
#if !defined(__HIP_DEVICE_COMPILE__)
// The host runs ~520 of these per open, one after the other, each on the critical path of a sumcheck round (the GPU waits for the
// challenge).  Same function with lazy reductions: values ride as any residue below 2^62 + 8, the square uses
// (a + b)(a - b) + 2ab i (two products instead of three), folds replace the conditional subtractions, and the result is made
// canonical once at the end.  Exact integer arithmetic mod p throughout, so the output equals mimc_hash_plain's bit for bit
// (tests/test_abi.py checks 20 000 inputs incl. the edge set against the oracle).
static inline uint64_t lz_fold(uint64_t s) { return (s & P61) + (s >> 61); }                       // any u64 -> < 2^61 + 8, same residue
static inline uint64_t lz_red(u128 x) {                                                             // x < 2^128 -> < 2^62 + 2^6, same residue
    const uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
    return (lo & P61) + (((hi << 3) | (lo >> 61)) & P61) + (hi >> 58);
}
inline F mimc_hash_lazy(const F &x, const F &k) {
    uint64_t hr = 0, hi_ = 0;                                                                      // h, lazy
    const uint64_t kr = k.re, ki = k.im;
    for (int i = 0; i < 161; i++) {
        // t = h + k + c, folded below 2^61 + 8
        const uint64_t a = lz_fold(i == 0 ? x.re + kr : hr + kr + (uint64_t)(i - 1)), b = lz_fold(i == 0 ? x.im + ki : hi_ + ki);
        // t^2 = (a + b)(a - b) + 2ab i ; a - b taken as a + 2p - b > 0 (b < 2p)
        const uint64_t sr = lz_red((u128)(a + b) * (a + 2 * P61 - b)), si = lz_red(((u128)a * b) << 1);     // < 2^62 + 2^6
        // t^3 = t^2 t, Karatsuba; C = 4p 2^62 >= bd keeps the real part non-negative (ac + C < 2^127)
        const u128 ac = (u128)sr * a, bd = (u128)si * b, all = (u128)(sr + si) * (a + b);
        hr = lz_red(ac + ((((u128)P61) << 64)) - bd);
        hi_ = lz_red(all - ac - bd);
    }
    const uint64_t r = lz_fold(lz_fold(hr + kr)), m = lz_fold(lz_fold(hi_ + ki));                  // <= p + small
    return fmake(r >= P61 ? r - P61 : r, m >= P61 ? m - P61 : m);
}
inline F mimc_hash(const F &x, const F &k) {
    const F o = mimc_hash_lazy(x, k);
    TranscriptRec &t = transcript_rec();
    if (t.on) { const uint64_t rec[6] = {x.re, x.im, k.re, k.im, o.re, o.im}; t.w.insert(t.w.end(), rec, rec + 6); }
    return o;
}
#else
HB_HD F mimc_hash(const F &x, const F &k) { return mimc_hash_plain(x, k); }
#endif

}  // namespace hobbit
