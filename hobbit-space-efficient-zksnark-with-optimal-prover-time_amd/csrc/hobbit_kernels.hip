// hobbit_kernels.hip -- gfx950 kernels of the HOBBIT prover hot path and their launchers.
//
// All kernels are integer (F_{p^2}, p = 2^61-1) or hash work: no MFMA.  The streaming kernels
// (fold, aggregate, leaf chain, tree levels) are HBM-bound and use 16-byte-per-lane coalesced
// accesses; the FFT and the expander encode keep a whole row / codeword in LDS (64 KB / <=110 KB
// of the CU's 160 KB) and are bound by the v_mad_u64_u32 rate.
#include "hobbit_kernels.hpp"
#include <atomic>
#include "hobbit_blake3.hpp"
#include <vector>

namespace hobbit {
// The dynamic-LDS limit is an attribute of the function ON ONE DEVICE: latched per device (bit = device ordinal), not per process -- a host that
// creates contexts on several devices from one process must set it on each (torchrun's one device per process never noticed).
static inline void set_lds_limit_once(hobbit_ctx *ctx, const void *kernel, int bytes, std::atomic<uint64_t> &done) {
    const uint64_t bit = 1ull << (ctx->device & 63);
    if (!(done.load(std::memory_order_relaxed) & bit)) { hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); done.fetch_or(bit); }
}

// ============================================================================================
// small utilities
// ============================================================================================
__device__ __forceinline__ F ldF(const F *p) {
    const uint4 v = *reinterpret_cast<const uint4 *>(p);     // one global_load_dwordx4 / ds_read_b128
    F r; r.re = (uint64_t)v.x | ((uint64_t)v.y << 32); r.im = (uint64_t)v.z | ((uint64_t)v.w << 32); return r;
}
__device__ __forceinline__ void stF(F *p, const F &a) {
    uint4 v; v.x = (uint32_t)a.re; v.y = (uint32_t)(a.re >> 32); v.z = (uint32_t)a.im; v.w = (uint32_t)(a.im >> 32);
    *reinterpret_cast<uint4 *>(p) = v;
}
__device__ __forceinline__ F shfl_down_F(const F &a, int d) {
    F r;
    r.re = __shfl_down((unsigned long long)a.re, d, 64);
    r.im = __shfl_down((unsigned long long)a.im, d, 64);
    return r;
}
// sum over the 64 lanes of a wavefront; result valid in lane 0
__device__ __forceinline__ F wave_sum(F a) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) a = fadd(a, shfl_down_F(a, d));
    return a;
}

__global__ void k_f_binop(int op, const F *a, const F *b, F *o, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        F x = ldF(a + i), y = ldF(b + i);
        stF(o + i, op == 0 ? fadd(x, y) : op == 1 ? fsub(x, y) : fmul(x, y));
    }
}
__device__ __forceinline__ uint64_t splitmix(uint64_t seed, uint64_t idx) {
    uint64_t z = seed * 0x632BE59BD9B4E019ULL + idx * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__global__ void k_fill_splitmix(F *o, size_t n, uint64_t seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        stF(o + i, fmake(splitmix(seed, 2 * i + 1) % P61, splitmix(seed, 2 * i + 2) % P61));
}

static inline int grid_for(size_t n, int block, int cap = 4096) {
    size_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > (size_t)cap) g = cap;
    return (int)g;
}

int launch_f_binop(hobbit_ctx *ctx, int op, const F *a, const F *b, F *o, size_t n) {
    HB_LAUNCH(ctx, "k_f_binop", k_f_binop, dim3(grid_for(n, 256)), dim3(256), 0, op, a, b, o, n);
    return 0;
}
// Multi-GPU aggregate exchange (parallel.sharded_open): the per-rank partial aggregates are summed by a plain 64-bit integer all-reduce -- field
// elements are < 2^61, so up to eight of them add up without leaving 64 bits -- and reduced mod p afterwards.  fold == 0: w += bias (the
// partials are shifted by -2^60 first, so that a SIGNED 64-bit sum cannot overflow either); fold != 0: w = (w + bias) mod p, canonical.
__global__ void k_u64_bias_fold(uint64_t *w, size_t n, uint64_t bias, int fold) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t x = w[i] + bias;
        if (fold) { x = (x & P61) + (x >> 61); x = x >= P61 ? x - P61 : x; }
        w[i] = x;
    }
}
int launch_u64_bias_fold(hobbit_ctx *ctx, uint64_t *w, size_t n, uint64_t bias, int fold) {
    HB_LAUNCH(ctx, "k_u64_bias_fold", k_u64_bias_fold, dim3(grid_for(n, 256, 8192)), dim3(256), 0, w, n, bias, fold);
    return 0;
}
int launch_fill_splitmix(hobbit_ctx *ctx, F *o, size_t n, uint64_t seed) {
    HB_LAUNCH(ctx, "k_fill_splitmix", k_fill_splitmix, dim3(grid_for(n, 256)), dim3(256), 0, o, n, seed);
    return 0;
}

// ============================================================================================
// FFT: one workgroup per row, the whole row (<= 4096 x 16 B = 64 KB) in LDS.
// Radix-2 DIT on bit-reversed input == the reference's _fft (src/utils.cpp:605-673): same DFT,
// natural order in and out.  Two radix-2 stages are fused per barrier (radix-4 in registers).
// The source row may be shorter than the transform (zero padding, src/PC_utils.cpp:75-84) and the
// destination may be strided (dst_es = element stride), which is how the row FFT of the tensor
// code writes straight into the codeword-major tensor.
// ============================================================================================
// Short transforms (len < 1024) are packed: a workgroup owns tpw = 1024 / len consecutive rows, so that every radix-4 stage still
// has one butterfly per thread (a single 256-point row would keep one wave of four busy).
// The stages themselves, on `span / len` bit-reversed rows held in LDS.  Element i of a row sits at slot i + i/8 (one pad per eight):
// the stride-4 and stride-16 butterflies of the first two radix-4 stages then spread over all banks (a plain layout is 4-way
// conflicted there); the row stride ldr must be >= fft_row_slots(len).
__device__ __host__ __forceinline__ uint32_t fft_slot(uint32_t i) { return i + (i >> 3); }
__device__ __host__ __forceinline__ uint32_t fft_row_slots(uint32_t len) { return len + (len >> 3); }
__device__ __forceinline__ void fft_lds_stages(F *s, int logn, const F *__restrict__ tw, uint32_t span, uint32_t ldr) {
    const uint32_t len = 1u << logn;
    uint32_t h = 1;
    int st = 0;
    if (logn & 1) {   // single radix-2 stage first when the stage count is odd
        for (uint32_t g = threadIdx.x; g < span / 2; g += blockDim.x) {
            const uint32_t base = (g >> (logn - 1)) * ldr, i = 2 * (g & (len / 2 - 1));
            F u = ldF(&s[base + fft_slot(i)]), v = ldF(&s[base + fft_slot(i + 1)]);   // twiddle w^0 = 1
            stF(&s[base + fft_slot(i)], fadd(u, v)); stF(&s[base + fft_slot(i + 1)], fsub(u, v));
        }
        __syncthreads();
        h = 2; st = 1;
    }
    for (; st < logn; st += 2, h <<= 2) {
        const uint32_t sA = len / (2 * h), sB = len / (4 * h);   // twiddle strides of the two stages
        for (uint32_t g = threadIdx.x; g < span / 4; g += blockDim.x) {
            const uint32_t t = g >> (logn - 2), b = g & (len / 4 - 1);
            const uint32_t k = b & (h - 1), j = b / h;
            const uint32_t base = t * ldr, i0 = j * 4 * h + k;
            const uint32_t p0 = base + fft_slot(i0), p1 = base + fft_slot(i0 + h), p2 = base + fft_slot(i0 + 2 * h), p3 = base + fft_slot(i0 + 3 * h);
            F a0 = ldF(&s[p0]), a1 = ldF(&s[p1]), a2 = ldF(&s[p2]), a3 = ldF(&s[p3]);
            const F wA = ldF(tw + (size_t)k * sA);
            F t1 = fmul(a1, wA), t3 = fmul(a3, wA);
            F b0 = fadd(a0, t1), b1 = fsub(a0, t1), b2 = fadd(a2, t3), b3 = fsub(a2, t3);
            const F wB0 = ldF(tw + (size_t)k * sB), wB1 = ldF(tw + (size_t)(k + h) * sB);
            F u2 = fmul(b2, wB0), u3 = fmul(b3, wB1);
            stF(&s[p0], fadd(b0, u2)); stF(&s[p2], fsub(b0, u2));
            stF(&s[p1], fadd(b1, u3)); stF(&s[p3], fsub(b1, u3));
        }
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256)
k_fft_rows(const F *__restrict__ src, size_t src_ld, uint32_t src_len, F *__restrict__ dst, size_t dst_ld, size_t dst_es,
           int logn, const F *__restrict__ tw, F scale, int do_scale, uint32_t rows_per_group, size_t src_gs, size_t dst_gs, uint32_t total_rows,
           uint32_t tpw) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    F *s = reinterpret_cast<F *>(lds_raw);
    const uint32_t len = 1u << logn;
    const uint32_t row0 = blockIdx.x * tpw, nrows = min(tpw, total_rows - row0), span = nrows * len, ldr = fft_row_slots(len);
    // row = (group, r): lets one launch cover K chunks x trs rows with per-chunk base strides
    for (uint32_t e = threadIdx.x; e < span; e += blockDim.x) {
        const uint32_t t = e >> logn, i = e & (len - 1), row = row0 + t;
        const F *in = src + (size_t)(row / rows_per_group) * src_gs + (size_t)(row % rows_per_group) * src_ld;
        F v = i < src_len ? ldF(in + i) : fmake(0);
        stF(&s[t * ldr + fft_slot(__brev(i) >> (32 - logn))], v);
    }
    __syncthreads();
    fft_lds_stages(s, logn, tw, span, ldr);
    for (uint32_t e = threadIdx.x; e < span; e += blockDim.x) {
        const uint32_t t = e >> logn, i = e & (len - 1), row = row0 + t;
        F *out = dst + (size_t)(row / rows_per_group) * dst_gs + (size_t)(row % rows_per_group) * dst_ld;
        F v = ldF(&s[t * ldr + fft_slot(i)]);
        if (do_scale) v = fmul(v, scale);
        stF(out + (size_t)i * dst_es, v);
    }
}
// The second half of a long transform len = 4096 R in ONE pass (was: twiddle + transpose, 4096 x FFT-R, transpose).  Input
// y[n1][k2] (R rows of 4096: the R sub-transforms), output X[k1 * 4096 + k2] = sum_n1 W_R^(n1 k1) W_len^(n1 k2) y[n1][k2].
// A workgroup owns C = 16 adjacent k2 for all n1: loads and stores are 256-byte segments, the twiddles come from the 2-D table
// tw2[n1][k2] with the same pattern, each of the C columns is one length-R transform in LDS (row stride odd: the column-major
// fill and drain are bank-conflict-free).
__global__ void __launch_bounds__(512)
k_fft_cols(const F *__restrict__ y, size_t gs, int logr, F *__restrict__ out, const F *__restrict__ tw2, const F *__restrict__ twr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    F *s = reinterpret_cast<F *>(lds_raw);
    constexpr uint32_t C = 16;
    const uint32_t R = 1u << logr, ldr = fft_row_slots(R) + 1, k20 = blockIdx.x * C;
    const F *src = y + (size_t)blockIdx.y * gs;
    F *dst = out + (size_t)blockIdx.y * gs;
    for (uint32_t e = threadIdx.x; e < C * R; e += blockDim.x) {
        const uint32_t n1 = e / C, t = e % C;
        const size_t at = (size_t)n1 * 4096 + k20 + t;
        F v = ldF(src + at);
        if (n1) v = fmul(v, ldF(tw2 + at));                               // W_len^(n1 k2); row 0 of the table is all ones
        stF(&s[t * ldr + fft_slot(__brev(n1) >> (32 - logr))], v);
    }
    __syncthreads();
    fft_lds_stages(s, logr, twr, C * R, ldr);
    for (uint32_t e = threadIdx.x; e < C * R; e += blockDim.x) {
        const uint32_t k1 = e / C, t = e % C;
        stF(dst + (size_t)k1 * 4096 + k20 + t, ldF(&s[t * ldr + fft_slot(k1)]));
    }
}
int launch_fft_cols(hobbit_ctx *ctx, const F *y, size_t gs, int logr, F *out, const F *tw2, const F *twr, uint32_t batch) {
    if (logr < 2 || logr > 8) return ctx->fail(HOBBIT_EINVAL, "fft_cols: R must be in [4, 256]");
    const size_t lds = (size_t)16 * (fft_row_slots(1u << logr) + 1) * 16;
    static std::atomic<uint64_t> attr_set{0};
    set_lds_limit_once(ctx, (const void *)k_fft_cols, 16 * 289 * 16, attr_set);
    HB_LAUNCH(ctx, "k_fft_cols", k_fft_cols, dim3(4096 / 16, batch), dim3(512), lds, y, gs, logr, out, tw2, twr);
    return 0;
}

int launch_fft_rows(hobbit_ctx *ctx, const F *src, size_t src_ld, uint32_t src_len, F *dst, size_t dst_ld, size_t dst_es,
                    int logn, const F *tw, F scale, int do_scale, uint32_t groups, uint32_t rows_per_group, size_t src_gs, size_t dst_gs) {
    if (logn < 1 || logn > 12) return ctx->fail(HOBBIT_EINVAL, "fft: logn must be in [1,12] for the LDS-resident kernel");
    static std::atomic<uint64_t> attr_set{0};
    set_lds_limit_once(ctx, (const void *)k_fft_rows, 81920, attr_set);
    const size_t total = (size_t)groups * rows_per_group;
    if (total == 0) return 0;
    if (total >> 32) return ctx->fail(HOBBIT_EINVAL, "fft: too many rows");
    const uint32_t tpw = logn < 10 ? (1024u >> logn) : 1u;
    const size_t lds = (size_t)16 * fft_row_slots(1u << logn) * tpw;
    const size_t blocks = (total + tpw - 1) / tpw;
    HB_LAUNCH(ctx, "k_fft_rows", k_fft_rows, dim3((unsigned)blocks), dim3(256), lds, src, src_ld, src_len, dst, dst_ld, dst_es, logn, tw,
              scale, do_scale, rows_per_group, src_gs, dst_gs, (uint32_t)total, tpw);
    return 0;
}

// --------------------------------------------------------------------------------------------
// FFT-4096, the hot size (row RS-encode of every tensor code): 512 threads, radix-8 DIT in
// registers, four passes, LDS padded by one slot per eight (phys(i) = i + i/8) so that the
// stride-8 / stride-64 / stride-512 butterflies and the bit-reversed first-pass gather are all
// bank-conflict-free; 73.7 KB of LDS -> two workgroups per CU.
//   load     global -> LDS in natural order (coalesced, contiguous LDS writes)
//   pass 0   butterfly b reads x[rev9(b) + 512 u]  (= DIT positions 8b+t), writes positions 8b+t.
//            Its twiddles are the 8th roots of unity: 1, w4 = +-i (free), w8, w8^3; with the
//            zero-padded message rows (upper half 0) the first stage has no arithmetic at all.
//   pass 1,2 in place, twiddle tables stored per pass as [7][h] (contiguous in k)
//   pass 3   results go straight to global memory (row-major or codeword-major/transposed)
// Same DFT as the reference's _fft (src/utils.cpp:605-673): bit-exact by exactness of F_{p^2}.
// --------------------------------------------------------------------------------------------
struct Fft4kConst { F w8, w8_3; int w4_plus_i; };

// Inside the transform the sums are lazy.  p = 2^61 - 1 leaves three spare bits in a 64-bit word (8p + 7 = 2^64 - 1), exactly the
// growth of the three add/sub levels of an 8-point DFT: a + b is a plain 64-bit add, a - b is a + (B p - b) with B p the static bound of
// b, and every output is folded once (mask / shift / add) into [0, p + 7] -- 16 folds per octet instead of 48.  Bounds: the octet's
// first element comes from LDS in [0, p + 7], the other seven are products (canonical, < p); the first element is always the left
// operand of its butterflies, so the +7 only rides along: level 1 <= 2p + 7, level 2 <= 4p + 7, level 3 <= 8p + 7.
// (Same-box A/B at 2^28: 5.56 vs 6.04 ms for the zero-padded rows, 6.26 vs 6.74 ms for full rows.)
__device__ __forceinline__ uint64_t fold61(uint64_t s) { return (s & P61) + (s >> 61); }            // any u64 -> [0, p + 7], same residue
__device__ __forceinline__ F ffold(const F &a) { return fmake(fold61(a.re), fold61(a.im)); }
__device__ __forceinline__ F fcanon(const F &a) {                                                    // any u64 pair -> canonical
    const uint64_t r = fold61(a.re), i = fold61(a.im);
    return fmake(r >= P61 ? r - P61 : r, i >= P61 ? i - P61 : i);
}
__device__ __forceinline__ F faddl(const F &a, const F &b) { return fmake(a.re + b.re, a.im + b.im); }
template <int B> __device__ __forceinline__ F fsubl(const F &a, const F &b) {                       // b's components <= B p
    return fmake(a.re + ((uint64_t)B * P61 - b.re), a.im + ((uint64_t)B * P61 - b.im));
}
// a * b for a lazy a (components < 2^62 - 2^10) and a canonical twiddle b.  Same Karatsuba form as fmul, with the offset that keeps
// ac - bd non-negative doubled: C = p 2^62 >= bd, ac + C < 2^124, ad + bc < 2^124.
#if !defined(HOBBIT_FMUL_U128)
// (the carry-free limb product of hobbit_field.hpp takes components up to p + 7 -- what ffold leaves -- and its canonical form is what the octet's
// bounds need for the seven right-hand operands)
__device__ __forceinline__ F fmul_lz(const F &a, const F &b) { return fmul(a, b); }
#else
__device__ __forceinline__ F fmul_lz(const F &a, const F &b) {
    const u128 ac = (u128)a.re * b.re, bd = (u128)a.im * b.im;
    const u128 all = (u128)(a.re + a.im) * (b.re + b.im);
    const u128 C = ((u128)P61) << 62;
    return fmake(red124(ac + C - bd), red124(all - ac - bd));
}
#endif
template <int B> __device__ __forceinline__ F fmul_w4(const F &a, int plus_i) {   // a * (+i) or a * (-i); components <= B p in and out
    return plus_i ? fmake((uint64_t)B * P61 - a.im, a.re) : fmake(a.im, (uint64_t)B * P61 - a.re);
}
// a * w8, w8 = the primitive 8th root of unity of the transform direction.  sqrt(2) = 2^31 in F_p (2^62 = 2), so
// w8 = 2^30 (1 - i) forward (w4 = -i) and 2^30 (1 + i) inverse: a rotation by 30 bits of (a.re +- a.im), no product at all.
__device__ __forceinline__ uint64_t rot30(uint64_t x) {      // x * 2^30 mod p for any u64 x; result < 2^61 + 2^31 <= 2p
    x = fold61(x);                                           // < 2^62: x = lo31 + hi 2^31, x 2^30 = lo31 2^30 + hi (2^61 = 1)
    return ((x << 30) & P61) + (x >> 31);
}
template <int B> __device__ __forceinline__ F fmul_w8(const F &a, int plus_i) {   // components <= B p (B <= 4) in, <= 2p out
    const uint64_t s = a.re + a.im;
    return plus_i ? fmake(rot30(a.re + ((uint64_t)B * P61 - a.im)), rot30(s)) : fmake(rot30(s), rot30(a.im + ((uint64_t)B * P61 - a.re)));
}
__device__ __forceinline__ uint32_t fft_phys(uint32_t i) { return i + (i >> 3); }
#define HB_BFLY(x, y, w) do { F v__ = fmul(y, w); y = fsub(x, v__); x = fadd(x, v__); } while (0)
#define HB_BFLYL(B, x, y) do { F v__ = y; y = fsubl<B>(x, v__); x = faddl(x, v__); } while (0)      /* y <= B p */
// first stage of the 8-point DIT DFT: right operands canonical
#define HB_DFT8_HEAD(a) do { HB_BFLYL(1, a[0], a[1]); HB_BFLYL(1, a[2], a[3]); HB_BFLYL(1, a[4], a[5]); HB_BFLYL(1, a[6], a[7]); } while (0)
// the last two stages of an 8-point DIT DFT on a[0..7] (inputs in bit-reversed order, first stage done): twiddles 1, w4, w8, w8^3.
// In: a[0], a[1] <= 2p + 7, the rest <= 2p.  Out: <= 8p + 7 = 2^64 - 1, unfolded.
__device__ __forceinline__ void dft8_tail(F (&a)[8], int plus_i) {
    F t;
    HB_BFLYL(2, a[0], a[2]);
    t = fmul_w4<2>(a[3], plus_i); a[3] = fsubl<2>(a[1], t); a[1] = faddl(a[1], t);
    HB_BFLYL(2, a[4], a[6]);
    t = fmul_w4<2>(a[7], plus_i); a[7] = fsubl<2>(a[5], t); a[5] = faddl(a[5], t);
    // left operands a[0..3] <= 4p + 7, right operands a[4..7] <= 4p
    HB_BFLYL(4, a[0], a[4]);
    t = fmul_w8<4>(a[5], plus_i); a[5] = fsubl<2>(a[1], t); a[1] = faddl(a[1], t);
    t = fmul_w4<4>(a[6], plus_i); a[6] = fsubl<4>(a[2], t); a[2] = faddl(a[2], t);
    t = fmul_w4<2>(fmul_w8<4>(a[7], plus_i), plus_i); a[7] = fsubl<2>(a[3], t); a[3] = faddl(a[3], t);
}

template <bool PADDED>
__global__ void __launch_bounds__(512)
k_fft4096(const F *__restrict__ src, size_t src_ld, size_t src_es, F *__restrict__ dst, size_t dst_ld, size_t dst_es, const F *__restrict__ tw1,
          const F *__restrict__ tw2, const F *__restrict__ tw3, Fft4kConst cst, F scale, int do_scale, uint32_t rows_per_group, size_t src_gs,
          size_t dst_gs, int line_remap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    F *s = reinterpret_cast<F *>(lds_raw);
    // Transposed stores (dst_ld == 1): row r's element lands 16 B into the 128-byte line it shares with rows 8g .. 8g+7.  Workgroups b, b+8,
    // b+16, ... share an XCD and its L2, so with line_remap the eight rows of a line group go to eight consecutive slots of ONE XCD, whose
    // L2 can merge the pieces into whole lines (within every run of 64 blocks: block (xcd, j) -> row 8 xcd + j).  Placement only.
    uint32_t bid = blockIdx.x;
    if (line_remap) bid = (bid & ~63u) + 8 * (bid & 7u) + ((bid >> 3) & 7u);
    const uint32_t grp = bid / rows_per_group, r = bid % rows_per_group;
    const F *in = src + (size_t)grp * src_gs + (size_t)r * src_ld;
    F *out = dst + (size_t)grp * dst_gs + (size_t)r * dst_ld;
    const uint32_t tid = threadIdx.x;
    const int plus_i = cst.w4_plus_i;
    constexpr uint32_t NLOAD = PADDED ? 2048 : 4096;
    F w[7];                                                         // the next pass's seven twiddles, requested one pass early (-2 %, same-box A/B)
#pragma unroll
    for (int t8 = 0; t8 < 7; t8++) w[t8] = ldF(tw1 + t8 * 8 + (tid & 7));          // pass 1's, in flight over the load and pass 0
#pragma unroll
    for (uint32_t i = 0; i < NLOAD; i += 512) stF(&s[fft_phys(i + tid)], ldF(in + (size_t)(i + tid) * src_es));
    __syncthreads();
    F a[8];
    {   // ---- pass 0 (h = 1): a plain 8-point DFT
        const uint32_t m = __brev(tid) >> 23;                       // rev9(b), b = tid
        a[0] = ldF(&s[fft_phys(m)]); a[2] = ldF(&s[fft_phys(m + 1024)]); a[4] = ldF(&s[fft_phys(m + 512)]); a[6] = ldF(&s[fft_phys(m + 1536)]);
        if (PADDED) { a[1] = a[0]; a[3] = a[2]; a[5] = a[4]; a[7] = a[6]; }       // (u, 0) -> (u, u)
        else {
            a[1] = ldF(&s[fft_phys(m + 2048)]); a[3] = ldF(&s[fft_phys(m + 3072)]); a[5] = ldF(&s[fft_phys(m + 2560)]); a[7] = ldF(&s[fft_phys(m + 3584)]);
            HB_DFT8_HEAD(a);
        }
        dft8_tail(a, plus_i);
        __syncthreads();                                            // every input has been read
        const uint32_t o = fft_phys(8 * tid);                       // 9*tid: positions 8b..8b+7 are contiguous
#pragma unroll
        for (int t8 = 0; t8 < 8; t8++) stF(&s[o + t8], ffold(a[t8]));
        __syncthreads();
    }
    // ---- passes 1..3: true radix-8 DIT.  Block t of the stride-h octet holds the sub-transform of the samples = rev3(t) mod 8, so
    // X[k + m h] = sum_t W_8^{rev3(t) m} (W_{8h}^{rev3(t) k} a[t]): seven general products (tables [7][h], t = 1..7), then the same
    // product-free 8-point DFT as pass 0 -- 7 instead of 12 products per octet.
#pragma unroll
    for (int pass = 1; pass <= 3; pass++) {
        const uint32_t h = pass == 1 ? 8u : pass == 2 ? 64u : 512u;
        const uint32_t k = tid & (h - 1), j = tid / h, i0 = j * 8 * h + k;
#pragma unroll
        for (int t8 = 0; t8 < 8; t8++) a[t8] = ldF(&s[fft_phys(i0 + t8 * h)]);
#pragma unroll
        for (int t8 = 1; t8 < 8; t8++) a[t8] = fmul_lz(a[t8], w[t8 - 1]);
        if (pass < 3) {                                    // next pass's twiddles: in flight over this pass's sums, stores and barrier
            const uint32_t hn = pass == 1 ? 64u : 512u;
            const F *twn = pass == 1 ? tw2 : tw3;
#pragma unroll
            for (int t8 = 0; t8 < 7; t8++) w[t8] = ldF(twn + t8 * hn + (tid & (hn - 1)));
        }
        HB_DFT8_HEAD(a);
        dft8_tail(a, plus_i);
        if (pass < 3) {
#pragma unroll
            for (int t8 = 0; t8 < 8; t8++) stF(&s[fft_phys(i0 + t8 * h)], ffold(a[t8]));
            __syncthreads();
        } else {
#pragma unroll
            for (int t8 = 0; t8 < 8; t8++) {
                F v = do_scale ? fmul_lz(ffold(a[t8]), scale) : fcanon(a[t8]);    // canonical out (a product already is)
                stF(out + (size_t)(i0 + t8 * h) * dst_es, v);
            }
        }
    }
}

int launch_fft4096(hobbit_ctx *ctx, const F *src, size_t src_ld, size_t src_es, uint32_t src_len, F *dst, size_t dst_ld, size_t dst_es, const F *tw1,
                   const F *tw2, const F *tw3, F w8, F w8_3, int w4_plus_i, F scale, int do_scale, uint32_t groups, uint32_t rows_per_group,
                   size_t src_gs, size_t dst_gs) {
    size_t blocks = (size_t)groups * rows_per_group;
    if (blocks == 0) return 0;
    const size_t lds = (size_t)(4096 + 512) * 16;
    Fft4kConst cst; cst.w8 = w8; cst.w8_3 = w8_3; cst.w4_plus_i = w4_plus_i;
    if (src_len == 2048) {
        hipFuncSetAttribute((const void *)k_fft4096<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        static const int remap_mode = [] { const char *e = getenv("HOBBIT_FFT_LINE_REMAP"); return e ? atoi(e) : 1; }();
        const int remap = (remap_mode && dst_ld == 1 && dst_es > 1 && rows_per_group % 64 == 0) ? 1 : 0;
        HB_LAUNCH(ctx, "k_fft4096", k_fft4096<true>, dim3((unsigned)blocks), dim3(512), lds, src, src_ld, src_es, dst, dst_ld, dst_es, tw1, tw2, tw3, cst,
                  scale, do_scale, rows_per_group, src_gs, dst_gs, remap);
    } else {
        hipFuncSetAttribute((const void *)k_fft4096<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        HB_LAUNCH(ctx, "k_fft4096_full", k_fft4096<false>, dim3((unsigned)blocks), dim3(512), lds, src, src_ld, src_es, dst, dst_ld, dst_es, tw1, tw2, tw3,
                  cst, scale, do_scale, rows_per_group, src_gs, dst_gs, 0);
    }
    return 0;
}

// Long rows (len = R * 4096, R = 2, 4, 8; Elastic_PC opt 2 uses 32768-point row codes,
// src/Elastic_PC.cpp:737-771) by one Cooley-Tukey split:
//   X[4096 k1 + k2] = sum_{n1 < R} W_R^{n1 k1} * W_len^{n1 k2} * Y_{n1}[k2],   Y_{n1} = FFT_4096(x[R n2 + n1])
// --------------------------------------------------------------------------------------------
// The same radix-8 machinery for the shorter transforms (64 ... 2048 points): the column codes of the RS x RS tensors
// (Elastic_PC option 1: 1024 points; Our_PC option 1 and the MLP witness: 256) ran in the generic radix-2/4 kernel at log2(N)/2 products
// per element.  N = 8^P * R (R = 1, 2, 4): the R sub-sequences x[R i' + q] are transformed by P radix-8 passes (7/8 product per element
// and pass, none in the first), then ONE radix-R stage combines them (1/2 resp. 3/4 product per element): 1024 points = 2.25 N products
// instead of 5 N.  512 threads, 4096 / N rows per workgroup, the LDS layout and the lazy sums of k_fft4096.
// tabs: [pass 1 | pass 2 | ...] each [7][h] = w_N2^(rev3(t) k N2 / (8h)), then the tail [R-1][N2] = w_N^(q k).  Forward transform only.
// --------------------------------------------------------------------------------------------
template <int LOGN, bool PADDED>
__global__ void __launch_bounds__(512)
k_fft_r8(const F *__restrict__ src, size_t src_ld, F *__restrict__ dst, size_t dst_ld, size_t dst_es, const F *__restrict__ tabs, int plus_i,
         uint32_t rows_per_group, size_t src_gs, size_t dst_gs, uint32_t total_rows) {
    constexpr uint32_t N = 1u << LOGN, P = LOGN / 3, R = 1u << (LOGN % 3), N2 = N / R, TPR = N / 8, ROWS = 512 / TPR, LDR = N + N / 8, OCT = N2 / 8;
    constexpr uint32_t SRCLEN = PADDED ? N / 2 : N;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    F *s = reinterpret_cast<F *>(lds_raw);
    const uint32_t tid = threadIdx.x, row0 = blockIdx.x * ROWS;
    // load: natural order, coalesced
    for (uint32_t e = tid; e < ROWS * SRCLEN; e += 512) {
        const uint32_t t = e / SRCLEN, i = e % SRCLEN, row = row0 + t;
        if (row < total_rows) {
            const F *in = src + (size_t)(row / rows_per_group) * src_gs + (size_t)(row % rows_per_group) * src_ld;
            stF(&s[t * LDR + fft_phys(i)], ldF(in + i));
        }
    }
    __syncthreads();
    const uint32_t t = tid / TPR, u = tid % TPR, q = u / OCT, b = u % OCT;
    F *sr = s + t * LDR;
    const bool live = row0 + t < total_rows;
    F a[8];
    {   // ---- pass 0 (h = 1): plain 8-point DFTs; octet b of sub-sequence q gathers x[R (rev(b) + OCT rev3(j)) + q]
        const uint32_t m = P > 1 ? (__brev(b) >> (32 - 3 * (P - 1))) : 0u;
        if (live) {
            a[0] = ldF(&sr[fft_phys(R * m + q)]); a[2] = ldF(&sr[fft_phys(R * (m + 2 * OCT) + q)]);
            a[4] = ldF(&sr[fft_phys(R * (m + OCT) + q)]); a[6] = ldF(&sr[fft_phys(R * (m + 3 * OCT) + q)]);
            if (PADDED) { a[1] = a[0]; a[3] = a[2]; a[5] = a[4]; a[7] = a[6]; }      // the upper half of the input is zero: (u, 0) -> (u, u)
            else {
                a[1] = ldF(&sr[fft_phys(R * (m + 4 * OCT) + q)]); a[3] = ldF(&sr[fft_phys(R * (m + 6 * OCT) + q)]);
                a[5] = ldF(&sr[fft_phys(R * (m + 5 * OCT) + q)]); a[7] = ldF(&sr[fft_phys(R * (m + 7 * OCT) + q)]);
                HB_DFT8_HEAD(a);
            }
            dft8_tail(a, plus_i);
        }
        __syncthreads();                                            // every input has been read
        if (live) {
            const uint32_t o = fft_phys(q * N2 + 8 * b);
#pragma unroll
            for (int t8 = 0; t8 < 8; t8++) stF(&sr[o + t8], ffold(a[t8]));
        }
        __syncthreads();
    }
    const F *tw = tabs;
#pragma unroll
    for (uint32_t pass = 1; pass < P; pass++) {
        const uint32_t h = pass == 1 ? 8u : pass == 2 ? 64u : 512u;
        const uint32_t k = b & (h - 1), j = b / h, i0 = q * N2 + j * 8 * h + k;
        if (live) {
#pragma unroll
            for (int t8 = 0; t8 < 8; t8++) a[t8] = ldF(&sr[fft_phys(i0 + t8 * h)]);
#pragma unroll
            for (int t8 = 1; t8 < 8; t8++) a[t8] = fmul_lz(a[t8], ldF(tw + (t8 - 1) * h + k));
            HB_DFT8_HEAD(a);
            dft8_tail(a, plus_i);
        }
        // (an octet reads and writes the same eight slots: no barrier between its loads and stores)
        if (live) {
            if (R == 1 && pass == P - 1) {
                F *out = dst + (size_t)((row0 + t) / rows_per_group) * dst_gs + (size_t)((row0 + t) % rows_per_group) * dst_ld;
#pragma unroll
                for (int t8 = 0; t8 < 8; t8++) stF(out + (size_t)(i0 + t8 * h) * dst_es, fcanon(a[t8]));
            } else {
#pragma unroll
                for (int t8 = 0; t8 < 8; t8++) stF(&sr[fft_phys(i0 + t8 * h)], ffold(a[t8]));
            }
        }
        tw += 7 * h;
        __syncthreads();
    }
    if (R == 1 || !live) return;
    F *out = dst + (size_t)((row0 + t) / rows_per_group) * dst_gs + (size_t)((row0 + t) % rows_per_group) * dst_ld;
    if (R == 2) {               // X[k] = E[k] + w^k O[k], X[k + N/2] = E[k] - w^k O[k]
#pragma unroll
        for (uint32_t c = 0; c < N2 / TPR; c++) {
            const uint32_t k = u + TPR * c;
            const F x = ldF(&sr[fft_phys(k)]), y = fmul_lz(ldF(&sr[fft_phys(N2 + k)]), ldF(tw + k));
            stF(out + (size_t)k * dst_es, fcanon(faddl(x, y)));
            stF(out + (size_t)(k + N2) * dst_es, fcanon(fsubl<1>(x, y)));
        }
    } else {                    // X[k + m N/4] = sum_q W_4^(q m) w^(q k) S_q[k]
#pragma unroll
        for (uint32_t c = 0; c < N2 / TPR; c++) {
            const uint32_t k = u + TPR * c;
            const F y0 = ldF(&sr[fft_phys(k)]);
            const F y1 = fmul_lz(ldF(&sr[fft_phys(N2 + k)]), ldF(tw + k));
            const F y2 = fmul_lz(ldF(&sr[fft_phys(2 * N2 + k)]), ldF(tw + N2 + k));
            const F y3 = fmul_lz(ldF(&sr[fft_phys(3 * N2 + k)]), ldF(tw + 2 * N2 + k));
            const F t0 = faddl(y0, y2), t1 = fsubl<1>(y0, y2), t2 = faddl(y1, y3), t3 = fmul_w4<2>(fsubl<1>(y1, y3), plus_i);     // <= 2p + 7, 2p + 7, 2p, 2p
            stF(out + (size_t)k * dst_es, fcanon(faddl(t0, t2)));
            stF(out + (size_t)(k + N2) * dst_es, fcanon(faddl(t1, t3)));
            stF(out + (size_t)(k + 2 * N2) * dst_es, fcanon(fsubl<2>(t0, t2)));
            stF(out + (size_t)(k + 3 * N2) * dst_es, fcanon(fsubl<2>(t1, t3)));
        }
    }
}
template <int LOGN>
static int launch_fft_r8_n(hobbit_ctx *ctx, const F *src, size_t src_ld, uint32_t src_len, F *dst, size_t dst_ld, size_t dst_es, const F *tabs, int plus_i,
                           uint32_t groups, uint32_t rows_per_group, size_t src_gs, size_t dst_gs) {
    constexpr uint32_t N = 1u << LOGN, ROWS = 512 / (N / 8);
    const size_t total = (size_t)groups * rows_per_group, lds = (size_t)(4096 + 512) * 16;
    const unsigned blocks = (unsigned)((total + ROWS - 1) / ROWS);
    if (src_len == N / 2) {
        hipFuncSetAttribute((const void *)k_fft_r8<LOGN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        HB_LAUNCH(ctx, "k_fft_r8", (k_fft_r8<LOGN, true>), dim3(blocks), dim3(512), lds, src, src_ld, dst, dst_ld, dst_es, tabs, plus_i, rows_per_group, src_gs, dst_gs, (uint32_t)total);
    } else {
        hipFuncSetAttribute((const void *)k_fft_r8<LOGN, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        HB_LAUNCH(ctx, "k_fft_r8", (k_fft_r8<LOGN, false>), dim3(blocks), dim3(512), lds, src, src_ld, dst, dst_ld, dst_es, tabs, plus_i, rows_per_group, src_gs, dst_gs, (uint32_t)total);
    }
    return 0;
}
// src_len must be N or N/2 (zero-padded upper half); 6 <= logn <= 11
int launch_fft_r8(hobbit_ctx *ctx, const F *src, size_t src_ld, uint32_t src_len, F *dst, size_t dst_ld, size_t dst_es, int logn, const F *tabs, int plus_i,
                  uint32_t groups, uint32_t rows_per_group, size_t src_gs, size_t dst_gs) {
    if ((size_t)groups * rows_per_group == 0) return 0;
    switch (logn) {
        case 6: return launch_fft_r8_n<6>(ctx, src, src_ld, src_len, dst, dst_ld, dst_es, tabs, plus_i, groups, rows_per_group, src_gs, dst_gs);
        case 7: return launch_fft_r8_n<7>(ctx, src, src_ld, src_len, dst, dst_ld, dst_es, tabs, plus_i, groups, rows_per_group, src_gs, dst_gs);
        case 8: return launch_fft_r8_n<8>(ctx, src, src_ld, src_len, dst, dst_ld, dst_es, tabs, plus_i, groups, rows_per_group, src_gs, dst_gs);
        case 9: return launch_fft_r8_n<9>(ctx, src, src_ld, src_len, dst, dst_ld, dst_es, tabs, plus_i, groups, rows_per_group, src_gs, dst_gs);
        case 10: return launch_fft_r8_n<10>(ctx, src, src_ld, src_len, dst, dst_ld, dst_es, tabs, plus_i, groups, rows_per_group, src_gs, dst_gs);
        case 11: return launch_fft_r8_n<11>(ctx, src, src_ld, src_len, dst, dst_ld, dst_es, tabs, plus_i, groups, rows_per_group, src_gs, dst_gs);
    }
    return ctx->fail(HOBBIT_EINVAL, "fft_r8: logn must be in [6, 11]");
}

// The same passes for the SECOND half of a long transform (k_fft_cols's job: N-point transforms down the columns of y[n1][k2], N = len / 4096
// = 32 ... 256, inter-stage twiddle on load, result X[k1 * 4096 + k2]).  A workgroup owns C = 4096 / N adjacent columns; the column index is
// the fastest thread index, so global loads and stores are C * 16-byte runs and -- with an odd LDS row stride -- the butterflies of the 64 lanes
// of a wave fall on different banks.  Products per element: 1 (twiddle) + 7/8 per radix-8 pass after the first + 1/2 or 3/4 for the tail,
// against 1 + log2(N)/2 in the radix-2/4 stages of k_fft_cols.
template <int LOGN>
__global__ void __launch_bounds__(512)
k_fft_cols_r8(const F *__restrict__ y, size_t gs, F *__restrict__ out, const F *__restrict__ tw2, const F *__restrict__ tabs, int plus_i) {
    constexpr uint32_t N = 1u << LOGN, P = LOGN / 3, T = 1u << (LOGN % 3), N2 = N / T, TPR = N / 8, C = 4096 / N, LDR = N + N / 8 + 1, OCT = N2 / 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    F *s = reinterpret_cast<F *>(lds_raw);
    const uint32_t tid = threadIdx.x, k20 = blockIdx.x * C;
    const F *src = y + (size_t)blockIdx.y * gs;
    F *dst = out + (size_t)blockIdx.y * gs;
    for (uint32_t e = tid; e < C * N; e += 512) {
        const uint32_t n1 = e / C, t = e % C;
        const size_t at = (size_t)n1 * 4096 + k20 + t;
        F v = ldF(src + at);
        if (n1) v = fmul(v, ldF(tw2 + at));                               // W_len^(n1 k2); row 0 of the table is all ones
        stF(&s[t * LDR + fft_phys(n1)], v);
    }
    __syncthreads();
    const uint32_t t = tid % C, u = tid / C, q = u / OCT, b = u % OCT;
    F *sr = s + t * LDR;
    F a[8];
    {   // pass 0: plain 8-point DFTs on the T interleaved sub-sequences
        const uint32_t m = P > 1 ? (__brev(b) >> (32 - 3 * (P - 1))) : 0u;
        a[0] = ldF(&sr[fft_phys(T * m + q)]); a[2] = ldF(&sr[fft_phys(T * (m + 2 * OCT) + q)]);
        a[4] = ldF(&sr[fft_phys(T * (m + OCT) + q)]); a[6] = ldF(&sr[fft_phys(T * (m + 3 * OCT) + q)]);
        a[1] = ldF(&sr[fft_phys(T * (m + 4 * OCT) + q)]); a[3] = ldF(&sr[fft_phys(T * (m + 6 * OCT) + q)]);
        a[5] = ldF(&sr[fft_phys(T * (m + 5 * OCT) + q)]); a[7] = ldF(&sr[fft_phys(T * (m + 7 * OCT) + q)]);
        HB_DFT8_HEAD(a);
        dft8_tail(a, plus_i);
        __syncthreads();
        if (P == 1 && T == 1) {
#pragma unroll
            for (int t8 = 0; t8 < 8; t8++) stF(dst + (size_t)t8 * 4096 + k20 + t, fcanon(a[t8]));
            return;
        }
        const uint32_t o = fft_phys(q * N2 + 8 * b);
#pragma unroll
        for (int t8 = 0; t8 < 8; t8++) stF(&sr[o + t8], ffold(a[t8]));
        __syncthreads();
    }
    const F *tw = tabs;
#pragma unroll
    for (uint32_t pass = 1; pass < P; pass++) {
        const uint32_t h = pass == 1 ? 8u : 64u;
        const uint32_t k = b & (h - 1), j = b / h, i0 = q * N2 + j * 8 * h + k;
#pragma unroll
        for (int t8 = 0; t8 < 8; t8++) a[t8] = ldF(&sr[fft_phys(i0 + t8 * h)]);
#pragma unroll
        for (int t8 = 1; t8 < 8; t8++) a[t8] = fmul_lz(a[t8], ldF(tw + (t8 - 1) * h + k));
        HB_DFT8_HEAD(a);
        dft8_tail(a, plus_i);
        if (T == 1 && pass == P - 1) {
#pragma unroll
            for (int t8 = 0; t8 < 8; t8++) stF(dst + (size_t)(i0 + t8 * h) * 4096 + k20 + t, fcanon(a[t8]));
        } else {
#pragma unroll
            for (int t8 = 0; t8 < 8; t8++) stF(&sr[fft_phys(i0 + t8 * h)], ffold(a[t8]));
        }
        tw += 7 * h;
        __syncthreads();
    }
    if (T == 1) return;
    if (T == 2) {
#pragma unroll
        for (uint32_t c = 0; c < N2 / TPR; c++) {
            const uint32_t k = u + TPR * c;
            const F x = ldF(&sr[fft_phys(k)]), yv = fmul_lz(ldF(&sr[fft_phys(N2 + k)]), ldF(tw + k));
            stF(dst + (size_t)k * 4096 + k20 + t, fcanon(faddl(x, yv)));
            stF(dst + (size_t)(k + N2) * 4096 + k20 + t, fcanon(fsubl<1>(x, yv)));
        }
    } else {
#pragma unroll
        for (uint32_t c = 0; c < N2 / TPR; c++) {
            const uint32_t k = u + TPR * c;
            const F y0 = ldF(&sr[fft_phys(k)]);
            const F y1 = fmul_lz(ldF(&sr[fft_phys(N2 + k)]), ldF(tw + k));
            const F y2 = fmul_lz(ldF(&sr[fft_phys(2 * N2 + k)]), ldF(tw + N2 + k));
            const F y3 = fmul_lz(ldF(&sr[fft_phys(3 * N2 + k)]), ldF(tw + 2 * N2 + k));
            const F t0 = faddl(y0, y2), t1 = fsubl<1>(y0, y2), t2 = faddl(y1, y3), t3 = fmul_w4<2>(fsubl<1>(y1, y3), plus_i);
            stF(dst + (size_t)k * 4096 + k20 + t, fcanon(faddl(t0, t2)));
            stF(dst + (size_t)(k + N2) * 4096 + k20 + t, fcanon(faddl(t1, t3)));
            stF(dst + (size_t)(k + 2 * N2) * 4096 + k20 + t, fcanon(fsubl<2>(t0, t2)));
            stF(dst + (size_t)(k + 3 * N2) * 4096 + k20 + t, fcanon(fsubl<2>(t1, t3)));
        }
    }
}
template <int LOGN>
static int launch_fft_cols_r8_n(hobbit_ctx *ctx, const F *y, size_t gs, F *out, const F *tw2, const F *tabs, int plus_i, uint32_t batch) {
    constexpr uint32_t N = 1u << LOGN, C = 4096 / N, LDR = N + N / 8 + 1;
    const size_t lds = (size_t)C * LDR * 16;
    hipFuncSetAttribute((const void *)k_fft_cols_r8<LOGN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    HB_LAUNCH(ctx, "k_fft_cols", (k_fft_cols_r8<LOGN>), dim3(4096 / C, batch), dim3(512), lds, y, gs, out, tw2, tabs, plus_i);
    return 0;
}
int launch_fft_cols_r8(hobbit_ctx *ctx, const F *y, size_t gs, int logr, F *out, const F *tw2, const F *tabs, int plus_i, uint32_t batch) {
    switch (logr) {
        case 5: return launch_fft_cols_r8_n<5>(ctx, y, gs, out, tw2, tabs, plus_i, batch);
        case 6: return launch_fft_cols_r8_n<6>(ctx, y, gs, out, tw2, tabs, plus_i, batch);
        case 7: return launch_fft_cols_r8_n<7>(ctx, y, gs, out, tw2, tabs, plus_i, batch);
        case 8: return launch_fft_cols_r8_n<8>(ctx, y, gs, out, tw2, tabs, plus_i, batch);
    }
    return ctx->fail(HOBBIT_EINVAL, "fft_cols_r8: R must be in [32, 256]");
}

// The R sub-transforms run in k_fft4096 (strided source); this kernel applies the twiddles and
// the R-point DFT across n1, one lane per k2 (coalesced on both sides).  tw[m] = W_len^m, m < len/2.
template <int LOGR>
__global__ void __launch_bounds__(256)
k_fft_combine(const F *__restrict__ Y, F *__restrict__ dst, size_t dst_ld, const F *__restrict__ tw, uint32_t rows) {
    constexpr uint32_t R = 1u << LOGR, len = R * 4096, half = len / 2;
    const size_t total = (size_t)rows * 4096;
    for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
        const uint32_t row = (uint32_t)(g >> 12), k2 = (uint32_t)(g & 4095);
        const F *y = Y + (size_t)row * len + k2;
        F a[R];
#pragma unroll
        for (uint32_t n1 = 0; n1 < R; n1++) {
            F v = ldF(y + (size_t)n1 * 4096);
            if (n1) {
                const uint32_t m = n1 * k2;                 // < len
                F w = ldF(tw + (m & (half - 1)));
                v = fmul(v, w);
                if (m >= half) v = fneg(v);                  // W^(m) = -W^(m - len/2)
            }
            a[__brev(n1) >> (32 - LOGR)] = v;               // bit-reversed order for the in-register DIT
        }
#pragma unroll
        for (uint32_t h = 1; h < R; h <<= 1)
#pragma unroll
            for (uint32_t b = 0; b < R / 2; b++) {
                const uint32_t k = b & (h - 1), i0 = (b / h) * 2 * h + k;
                F v = a[i0 + h];
                if (k) v = fmul(v, ldF(tw + (size_t)k * (len / (2 * h))));   // W_{2h}^k = W_len^(k len/2h)
                a[i0 + h] = fsub(a[i0], v); a[i0] = fadd(a[i0], v);
            }
        F *o = dst + (size_t)row * dst_ld + k2;
#pragma unroll
        for (uint32_t k1 = 0; k1 < R; k1++) stF(o + (size_t)k1 * 4096, a[k1]);
    }
}
int launch_fft_combine(hobbit_ctx *ctx, int logr, const F *Y, F *dst, size_t dst_ld, const F *tw, uint32_t rows) {
    size_t total = (size_t)rows * 4096;
    if (!total) return 0;
    dim3 g(grid_for(total, 256, 1 << 16)), b(256);
    if (logr == 1) HB_LAUNCH(ctx, "k_fft_combine", k_fft_combine<1>, g, b, 0, Y, dst, dst_ld, tw, rows);
    else if (logr == 2) HB_LAUNCH(ctx, "k_fft_combine", k_fft_combine<2>, g, b, 0, Y, dst, dst_ld, tw, rows);
    else if (logr == 3) HB_LAUNCH(ctx, "k_fft_combine", k_fft_combine<3>, g, b, 0, Y, dst, dst_ld, tw, rows);
    else return ctx->fail(HOBBIT_EINVAL, "fft_combine: row length must be 8192, 16384 or 32768");
    return 0;
}

// Row-major (rows x cols) -> codeword-major (cols x ld_out) tiled transpose through LDS: both the
// global reads and the global writes are 512-byte contiguous runs.  (Writing the FFT output
// transposed directly costs a 16-byte scattered store per element: measured 22 ms vs 8 ms for the
// 2^28 commit's row pass, hence this separate, HBM-bound pass.)
__global__ void __launch_bounds__(256)
k_transpose(const F *__restrict__ in, size_t in_gs, size_t in_ld, uint32_t rows, uint32_t cols, F *__restrict__ out, size_t out_gs, size_t ld_out) {
    __shared__ F tile[32][33];
    const F *src = in + (size_t)blockIdx.z * in_gs;
    F *dst = out + (size_t)blockIdx.z * out_gs;
    const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
    const uint32_t c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t r = r0 + ty + 8 * i;
        if (r < rows && c0 + tx < cols) stF(&tile[ty + 8 * i][tx], ldF(src + (size_t)r * in_ld + c0 + tx));
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t c = c0 + ty + 8 * i;
        if (c < cols && r0 + tx < rows) stF(dst + (size_t)c * ld_out + r0 + tx, ldF(&tile[tx][ty + 8 * i]));
    }
}
// Large exact multiples: 64 x 64 tiles, 1 KB contiguous on both sides instead of 512 B (measured at 2^28 on one box: 3.08 vs 3.42 ms;
// 128 x 32 and 128 x 16 tiles, longer on the write side only, gave 3.42 and 3.47).
template <int TR, int TC>
__global__ void __launch_bounds__(256)
k_transpose_big(const F *__restrict__ in, size_t in_gs, size_t in_ld, F *__restrict__ out, size_t out_gs, size_t ld_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    F *tile = reinterpret_cast<F *>(lds_raw);                              // [TR][TC + 1]
    const F *src = in + (size_t)blockIdx.z * in_gs;
    F *dst = out + (size_t)blockIdx.z * out_gs;
    const uint32_t c0 = blockIdx.x * TC, r0 = blockIdx.y * TR;
    {
        const uint32_t tx = threadIdx.x % TC, ty = threadIdx.x / TC;
#pragma unroll
        for (int i = 0; i < TR; i += 256 / TC) stF(&tile[(ty + i) * (TC + 1) + tx], ldF(src + (size_t)(r0 + ty + i) * in_ld + c0 + tx));
    }
    __syncthreads();
    {
        const uint32_t tx = threadIdx.x % TR, ty = threadIdx.x / TR;
#pragma unroll
        for (int i = 0; i < TC; i += 256 / TR) stF(dst + (size_t)(c0 + ty + i) * ld_out + r0 + tx, ldF(&tile[tx * (TC + 1) + ty + i]));
    }
}
int launch_transpose_ld(hobbit_ctx *ctx, const F *in, size_t in_gs, size_t in_ld, uint32_t rows, uint32_t cols, F *out, size_t out_gs, size_t ld_out,
                        uint32_t groups) {
    if (!rows || !cols || !groups) return 0;
    if (rows % 64 == 0 && cols % 64 == 0 && (size_t)rows * cols * groups >= ((size_t)1 << 24)) {
        const size_t lds = (size_t)64 * 65 * 16;
        hipFuncSetAttribute((const void *)k_transpose_big<64, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        HB_LAUNCH(ctx, "k_transpose", (k_transpose_big<64, 64>), dim3(cols / 64, rows / 64, groups), dim3(256), lds, in, in_gs, in_ld, out, out_gs, ld_out);
        return 0;
    }
    HB_LAUNCH(ctx, "k_transpose", k_transpose, dim3((cols + 31) / 32, (rows + 31) / 32, groups), dim3(256), 0, in, in_gs, in_ld, rows, cols, out, out_gs,
              ld_out);
    return 0;
}
int launch_transpose(hobbit_ctx *ctx, const F *in, size_t in_gs, uint32_t rows, uint32_t cols, F *out, size_t out_gs, size_t ld_out, uint32_t groups) {
    return launch_transpose_ld(ctx, in, in_gs, cols, rows, cols, out, out_gs, ld_out, groups);
}

// Long transforms (len = 4096 * R, R up to 4096) by one Cooley-Tukey split, every pass coalesced:
//   transpose (n2,n1)->(n1,n2) | R x FFT-4096 | twiddle W_len^(n1 k2) fused into the transpose (n1,k2)->(k2,n1)
//   | 4096 x FFT-R | transpose (k2,k1)->(k1,k2).  This kernel is the middle one.  tw[m] = W_len^m, m < len/2.
// tw2 != NULL: the twiddles come from a 2-D table tw2[n1 * 4096 + k2] = W_len^(n1 k2), read with the same coalesced pattern as the data
// (the 1-D lookup tw[n1 k2 mod half] is a 16-byte load per lane from 64 different cache lines: 0.71 -> see DESIGN.md 4).
__global__ void __launch_bounds__(256)
k_transpose_tw(const F *__restrict__ in, size_t gs, uint32_t R, F *__restrict__ out, const F *__restrict__ tw, uint32_t half, const F *__restrict__ tw2) {
    __shared__ F tile[32][33];
    const F *src = in + (size_t)blockIdx.z * gs;
    F *dst = out + (size_t)blockIdx.z * gs;
    const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const uint32_t k20 = blockIdx.x * 32, n10 = blockIdx.y * 32;          // input: R rows (n1) x 4096 cols (k2)
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t n1 = n10 + ty + 8 * i, k2 = k20 + tx;
        if (n1 < R) {
            F v = ldF(src + (size_t)n1 * 4096 + k2);
            const uint32_t m = n1 * k2;                                   // < len = 2*half
            if (tw2) { if (m) v = fmul(v, ldF(tw2 + (size_t)n1 * 4096 + k2)); }
            else if (m) { v = fmul(v, ldF(tw + (m & (half - 1)))); if (m >= half) v = fneg(v); }
            stF(&tile[ty + 8 * i][tx], v);
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t k2 = k20 + ty + 8 * i, n1 = n10 + tx;
        if (n1 < R) stF(dst + (size_t)k2 * R + n1, ldF(&tile[tx][ty + 8 * i]));
    }
}
__global__ void __launch_bounds__(256)
k_build_tw2d(const F *__restrict__ tw, uint32_t half, uint32_t R, F *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)R * 4096) return;
    const uint32_t n1 = (uint32_t)(i >> 12), k2 = (uint32_t)(i & 4095), m = n1 * k2;
    F v = ldF(tw + (m & (half - 1)));
    if (m >= half) v = fneg(v);
    stF(out + i, v);
}
int launch_build_tw2d(hobbit_ctx *ctx, const F *tw, uint32_t half, uint32_t R, F *out) {
    HB_LAUNCH(ctx, "k_build_tw2d", k_build_tw2d, dim3((unsigned)(((size_t)R * 4096 + 255) / 256)), dim3(256), 0, tw, half, R, out);
    return 0;
}
int launch_transpose_tw(hobbit_ctx *ctx, const F *in, size_t gs, uint32_t R, F *out, const F *tw, uint32_t half, const F *tw2, uint32_t groups) {
    HB_LAUNCH(ctx, "k_transpose_tw", k_transpose_tw, dim3(4096 / 32, (R + 31) / 32, groups), dim3(256), 0, in, gs, R, out, tw, half, tw2);
    return 0;
}

// ============================================================================================
// Expander encode (src/linear_code_encode.h:62-119), one workgroup per message, the whole
// codeword in LDS.  The recursion unrolls into a straight sequence of SpMV steps over one buffer
//   [x_0 | x_1 = C_0 x_0 | ... | x_D | z_{D-1} = D_{D-1} cw_D | ... | z_0 = D_0 cw_1]
// (the reference places the child codeword at dst+n and the D output behind it), executed in
// gather form (reverse adjacency: no atomics, deterministic, one writer per output).
// ============================================================================================
// Sliced-ELL geometry of one SpMV step: a slice is ENC_SW consecutive outputs; a wavefront owns a
// slice and splits each output's in-edges over ENC_SPLIT = 64/ENC_SW lane groups (lane l works on
// output l % ENC_SW, edges k = l / ENC_SW (mod ENC_SPLIT)), so one wave-instruction reads 64
// consecutive edge records (fully coalesced) and the per-lane dependent chain is ENC_SPLIT x
// shorter than one-lane-per-output.  The host pads every slice to a multiple of
// ENC_SPLIT*ENC_UNROLL edges per output with zero-weight records.
//
// SMALLW (all weights real and < 2^32, the only kind the reference ever draws, src/expanders.h:37):
// products are accumulated UNREDUCED in four 96-bit sums per output,
//     S_lo = sum w * lo32(a),  S_hi = sum w * hi32(a)        (for a = re and a = im)
// i.e. 4 v_mad_u64_u32 + 4 carry adds per edge and ONE Mersenne fold per output -- exact integer
// arithmetic, so the result is the same canonical element the reference's per-edge mod-p loop gives.
__device__ __forceinline__ void store8w(void *p, const uint32_t h[8]);
struct Acc96 { uint64_t lo; uint32_t hi; };
// a += w * x (32x32 -> 64 product into a 96-bit sum): one v_mad_u64_u32 whose carry-out feeds one
// v_addc (hipcc does not use the instruction's own carry output, hence the two-line asm).
__device__ __forceinline__ void acc96_mad(Acc96 &a, uint32_t w, uint32_t x) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc" : "+v"(a.lo), "+v"(a.hi) : "v"(w), "v"(x) : "vcc");
}
// the four sums of one edge (re.lo, re.hi, im.lo, im.hi) in ONE asm statement: between separate statements hipcc pads every vcc hand-over with an s_nop
__device__ __forceinline__ void acc96_mad4(Acc96 &a, Acc96 &b, Acc96 &c, Acc96 &d, uint32_t w, const uint4 &x) {
    asm("v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
        "v_mad_u64_u32 %2, vcc, %8, %10, %2\n\tv_addc_co_u32_e32 %3, vcc, 0, %3, vcc\n\t"
        "v_mad_u64_u32 %4, vcc, %8, %11, %4\n\tv_addc_co_u32_e32 %5, vcc, 0, %5, vcc\n\t"
        "v_mad_u64_u32 %6, vcc, %8, %12, %6\n\tv_addc_co_u32_e32 %7, vcc, 0, %7, vcc"
        : "+v"(a.lo), "+v"(a.hi), "+v"(b.lo), "+v"(b.hi), "+v"(c.lo), "+v"(c.hi), "+v"(d.lo), "+v"(d.hi)
        : "v"(w), "v"(x.x), "v"(x.y), "v"(x.z), "v"(x.w) : "vcc");
}
__device__ __forceinline__ uint64_t acc_fold(const Acc96 &lo, const Acc96 &hi) {
    // value = lo + hi 2^32 with lo, hi 96-bit sums (any in-degree < 2^29).  No 128-bit arithmetic: 2^64 = 8 and 2^61 = 1 (mod p), so a 96-bit
    // sum {l, h} folds to fold61(l) + 8h < 2^61 + 2^36, and for the high sum b = b0 + 2^29 b1: b 2^32 = (b0 << 32) + b1.
    const uint64_t a = ((lo.lo & P61) + (lo.lo >> 61)) + ((uint64_t)lo.hi << 3);
    const uint64_t b = ((hi.lo & P61) + (hi.lo >> 61)) + ((uint64_t)hi.hi << 3);
    const uint64_t t = a + ((b & 0x1FFFFFFFull) << 32) + (b >> 29);                     // < 2^63
    const uint64_t r = (t & P61) + (t >> 61);                                            // <= p + 3
    return r >= P61 ? r - P61 : r;
}
__device__ __forceinline__ F shfl_xor_F(const F &a, int m) {
    F r;
    r.re = __shfl_xor((unsigned long long)a.re, m, 64);
    r.im = __shfl_xor((unsigned long long)a.im, m, 64);
    return r;
}

// A launch executes steps [s_lo, s_hi) of the encode on codeword indices [base, base + LDS extent):
// it loads [ld_lo, ld_hi) from the column in global memory, runs the steps (an output that falls
// outside the LDS window goes straight to global memory), and stores [st_lo, st_hi) (zeros past
// the codeword).  Small codes run as ONE pass; for n = 4096 (110 KB codeword, one workgroup per
// CU) the launcher splits the work into pass A = C_0 alone (64 KB message window, 2 workgroups per
// CU) and pass B = everything else (46 KB window, 3 per CU) so that one workgroup's global
// load/store overlaps another's gather loop.
struct EncPass { uint32_t base, ld_lo, ld_hi, s_lo, s_hi, st_lo, st_hi, direct_out; };
template <bool SMALLW>
__global__ void __launch_bounds__(1024)
k_encode(const F *__restrict__ src, size_t ld_src, F *__restrict__ dst, size_t ld_dst, uint32_t len, EncPass ps,
         const EncStep *__restrict__ steps, const uint32_t *__restrict__ slice_ptr,
         const uint32_t *__restrict__ slice_width, const uint32_t *__restrict__ slice_out, const uint2 *__restrict__ e32,
         const uint32_t *__restrict__ eidx, const F *__restrict__ ew) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    F *cw = reinterpret_cast<F *>(lds_raw) - ps.base;          // cw[i] addresses codeword index i
    const uint32_t b = blockIdx.x;
    const F *in = src + (size_t)b * ld_src;
    F *out = dst + (size_t)b * ld_dst;
    // window load: eight global loads per thread in flight before the first LDS store (one at a time cost a full memory round
    // trip per element: 7 400 of pass A's 27 000 cycles per column, measured with s_memtime stamps)
    for (uint32_t b0 = ps.ld_lo; b0 < ps.ld_hi; b0 += 8 * blockDim.x) {
        F v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const uint32_t i = b0 + u * blockDim.x + threadIdx.x; if (i < ps.ld_hi) v[u] = ldF(in + i); }
#pragma unroll
        for (int u = 0; u < 8; u++) { const uint32_t i = b0 + u * blockDim.x + threadIdx.x; if (i < ps.ld_hi) stF(&cw[i], v[u]); }
    }
    // (the barrier that publishes the window sits after the first step's first edge-record request: those loads do not touch LDS)
    // wave index as a scalar: the slice descriptors below then come through the scalar cache (s_load) instead of a per-lane
    // global load, and the slice loop is scalar control flow
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = blockDim.x >> 6;
    for (uint32_t s = ps.s_lo; s < ps.s_hi; s++) {
        const EncStep sp = steps[s];
        const F *cin = cw + sp.in_off;
        const uint32_t sw = sp.sw, split = 64 / sw, tl = lane & (sw - 1);      // sw = 64: one output per lane; ENC_SW: lane groups share one
        if (SMALLW) {
            // slices of this wave: sl = wave, wave + nwaves, ...  The first group of edge records of the NEXT slice is requested
            // before the current slice's remainder / fold / shuffle / store, so its L2 latency is off the critical path.
            uint32_t sl = wave;
            bool have = sl < sp.n_slices;
            uint32_t nbase = 0, niters = 0;
            uint2 en[ENC_UNROLL];
            if (have) {
                nbase = slice_ptr[sp.slice_base + sl] + lane; niters = slice_width[sp.slice_base + sl] / split;
                if (niters >= ENC_UNROLL) {
#pragma unroll
                    for (int u = 0; u < ENC_UNROLL; u++) en[u] = e32[nbase + u * 64];
                }
            }
            if (s == ps.s_lo) __syncthreads();
            while (have) {
                const uint32_t base = nbase, iters = niters, cur = sl;
                Acc96 rl = {0, 0}, rh = {0, 0}, il = {0, 0}, ih = {0, 0};
                uint32_t j = 0;
                if (iters >= ENC_UNROLL) {
                    for (; j + ENC_UNROLL <= iters; j += ENC_UNROLL) {
                        uint2 e[ENC_UNROLL]; uint4 x[ENC_UNROLL];
#pragma unroll
                        for (int u = 0; u < ENC_UNROLL; u++) e[u] = en[u];
                        if (j + 2 * ENC_UNROLL <= iters) {     // prefetch the next full group of edge records
#pragma unroll
                            for (int u = 0; u < ENC_UNROLL; u++) en[u] = e32[base + (j + ENC_UNROLL + u) * 64];
                        }
#pragma unroll
                        for (int u = 0; u < ENC_UNROLL; u++) x[u] = *reinterpret_cast<const uint4 *>(cin + e[u].x);
#pragma unroll
                        for (int u = 0; u < ENC_UNROLL; u++) {
                            acc96_mad(rl, e[u].y, x[u].x); acc96_mad(rh, e[u].y, x[u].y);
                            acc96_mad(il, e[u].y, x[u].z); acc96_mad(ih, e[u].y, x[u].w);
                        }
                    }
                }
                sl += nwaves; have = sl < sp.n_slices;
                if (have) {                                   // next slice: descriptor + first record group in flight
                    nbase = slice_ptr[sp.slice_base + sl] + lane; niters = slice_width[sp.slice_base + sl] / split;
                    if (niters >= ENC_UNROLL) {
#pragma unroll
                        for (int u = 0; u < ENC_UNROLL; u++) en[u] = e32[nbase + u * 64];
                    }
                }
                for (; j < iters; j++) {                      // remainder (narrow steps only; rows are sorted by in-degree: at most ENC_UNROLL-1 records)
                    const uint2 e = e32[base + j * 64];
                    const uint4 x = *reinterpret_cast<const uint4 *>(cin + e.x);
                    acc96_mad(rl, e.y, x.x); acc96_mad(rh, e.y, x.y); acc96_mad(il, e.y, x.z); acc96_mad(ih, e.y, x.w);
                }
                F acc = fmake(acc_fold(rl, rh), acc_fold(il, ih));
                if (sw <= 16) acc = fadd(acc, shfl_xor_F(acc, 16));              // combine the lane groups (uniform branches)
                if (sw <= 32) acc = fadd(acc, shfl_xor_F(acc, 32));
                const uint32_t t = slice_out[sp.out_base + cur * sw + tl];       // outputs are sliced in order of in-degree
                if (lane < sw && t != 0xFFFFFFFFu) {
                    if (ps.direct_out) stF(out + sp.out_off + t, acc); else stF(&cw[sp.out_off + t], acc);
                }
            }
        } else {
            if (s == ps.s_lo) __syncthreads();
            for (uint32_t sl = wave; sl < sp.n_slices; sl += nwaves) {
                const uint32_t base = slice_ptr[sp.slice_base + sl] + lane;
                const uint32_t iters = slice_width[sp.slice_base + sl] / split;    // edge records per lane
                F acc = fmake(0);
                for (uint32_t j = 0; j < iters; j++) {
                    const uint32_t id = eidx[base + j * 64];
                    acc = fadd(acc, fmul(ldF(cin + id), ldF(ew + base + j * 64)));
                }
                if (sw <= 16) acc = fadd(acc, shfl_xor_F(acc, 16));
                if (sw <= 32) acc = fadd(acc, shfl_xor_F(acc, 32));
                const uint32_t t = slice_out[sp.out_base + sl * sw + tl];
                if (lane < sw && t != 0xFFFFFFFFu) {
                    if (ps.direct_out) stF(out + sp.out_off + t, acc); else stF(&cw[sp.out_off + t], acc);
                }
            }
        }
        __syncthreads();
    }
    for (uint32_t i = ps.st_lo + threadIdx.x; i < ps.st_hi; i += blockDim.x) stF(out + i, i < len ? ldF(&cw[i]) : fmake(0));
}

template <bool SMALLW>
static int launch_encode_pass(hobbit_ctx *ctx, const char *name, const F *src, size_t ld_src, F *dst, size_t ld_dst, size_t batch, EncPass ps,
                              uint32_t lds_elems, uint32_t block) {
    DeviceCode &c = ctx->code;
    hipFuncSetAttribute((const void *)k_encode<SMALLW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    HB_LAUNCH(ctx, name, (k_encode<SMALLW>), dim3((unsigned)batch), dim3(block), (size_t)lds_elems * 16, src, ld_src, dst, ld_dst, (uint32_t)c.len, ps,
              c.d_steps, c.d_slice_ptr, c.d_slice_width, c.d_slice_out, c.d_edges32, c.d_eidx, c.d_ew);
    return 0;
}
static uint32_t block_for(const DeviceCode &c, uint32_t s_lo, uint32_t s_hi, uint32_t max_waves) {
    uint32_t widest = 1;
    for (uint32_t s = s_lo; s < s_hi; s++) widest = c.steps[s].n_slices > widest ? c.steps[s].n_slices : widest;
    return (widest >= max_waves ? max_waves : widest) * 64;
}

// ============================================================================================
// Wide SpMV steps of a deep code (n = 4096), persistent and register-resident (hobbit_ctx.hpp FatStep).
// One workgroup of NW fat waves per CU for the whole launch.  A lane owns NOUT outputs of the step for good and keeps their edge records --
// 32-bit weight, 16-bit LDS byte offset -- in registers: the inner loop is one address add (SDWA word select), one ds_read_b128 and the
// 4 v_mad_u64_u32 + 4 v_addc of the unreduced 96-bit sums per edge; after the prologue no edge record, slice descriptor or index comes from
// memory.  The step's input window of the NEXT column streams into the second of two LDS buffers by LDS-DMA (global_load_lds_dwordx4, 1 KiB
// per wave-instruction, no registers) while the waves work on the current one; every wave issues its share of the pieces right after the
// one barrier per column that hands the buffers over.  The outputs of column c are stored one iteration late, behind that barrier, so that
// the `vmcnt(0)` which retires a wave's DMA pieces never waits for a store it has just issued.
// (The one-workgroup-per-column k_encode re-reads 8 bytes of record per edge and column through L2 -- 40 GB per commit at 2^28 -- and its
// window load, barriers and slice loop do not overlap: DESIGN.md section 4.)
// ============================================================================================
template <int NOUT, int CAP0, int CAP1, int CAP2, int NW, int NLOAD, int CP>
__global__ void __launch_bounds__((NW + NLOAD) * 64)
k_enc_fat(F *__restrict__ tensor, size_t ld, uint32_t ncols, uint32_t in_off, uint32_t in_len, uint32_t out_off, uint32_t z_lo, uint32_t z_hi,
          const uint32_t *__restrict__ wt, const uint32_t *__restrict__ ot, const uint32_t *__restrict__ oidx, const uint32_t *__restrict__ wid) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int LANES = NW * 64, TOT = CAP0 + CAP1 + CAP2;
    constexpr int CAP[3] = {CAP0, CAP1, CAP2}, BASE[3] = {0, CAP0, CAP0 + CAP1};
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint32_t pieces = (in_len + 63) >> 6, win_bytes = pieces << 10;             // whole 1-KiB DMA pieces
    // NLOAD > 0: the last NLOAD waves only load (a wave's DMA pieces cost it ~100+ cycles each under load: with dedicated loaders the consumers
    // never stall on them -- 2.5 vs 3.3 ms for C_0 at 2^28); NLOAD = 0: every wave issues its share right after the barrier
    // CP > 0 (with NLOAD > 0): every consumer wave also issues CP pieces per column right after the barrier, the loaders the rest -- one wave can keep
    // at most 63 vector-memory instructions in flight and its pieces cost it ~100+ cycles each, so a single loader caps the window stream
    constexpr int NISS = NLOAD > 0 ? NLOAD : NW;
    const bool loader = NLOAD > 0 && wave >= NW;
    const uint32_t first_piece = NLOAD > 0 ? (loader ? NW * CP + (wave - NW) : wave * CP) : wave;
    const uint32_t end_piece = (NLOAD > 0 && !loader) ? wave * CP + CP : 0xFFFFFFFFu, step_piece = (NLOAD > 0 && !loader) ? 1u : (uint32_t)NISS;
    auto issue = [&](uint32_t c, uint32_t buf) {                                         // this wave's pieces of column c's window -> buffer buf
        const F *colp = tensor + (size_t)c * ld + in_off;
        for (uint32_t p = first_piece; p < pieces && p < end_piece; p += step_piece) {
            uint32_t i = (p << 6) + lane; i = i < in_len ? i : in_len - 1;             // (the pad lanes of the last piece re-read the last element)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(colp + i),
                                             (__attribute__((address_space(3))) void *)(lds_raw + buf * win_bytes + (p << 10)), 16, 0, 0);
        }
    };
    if (NLOAD > 0) {
        if (loader) {
            uint32_t c = blockIdx.x, it = 0;
            if (c < ncols) issue(c, 0);
            for (; c < ncols; c += gridDim.x, it++) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         // this column's window has landed
                __builtin_amdgcn_s_barrier();                                              // ... and the consumers are done with the other buffer
                const uint32_t cn = c + gridDim.x;
                if (cn < ncols) issue(cn, (it + 1) & 1);
            }
            return;
        }
    }
    if ((NLOAD == 0 || CP > 0) && blockIdx.x < ncols) issue(blockIdx.x, 0);
    // edge records into registers, once
    uint32_t w[TOT], o[TOT / 2], oi[NOUT], W[NOUT];
#pragma unroll
    for (int k = 0; k < TOT; k++) w[k] = wt[(size_t)k * LANES + threadIdx.x];
#pragma unroll
    for (int k = 0; k < TOT / 2; k++) o[k] = ot[(size_t)k * LANES + threadIdx.x];
#pragma unroll
    for (int j = 0; j < NOUT; j++) { oi[j] = oidx[(size_t)j * LANES + threadIdx.x]; W[j] = __builtin_amdgcn_readfirstlane(wid[wave * NOUT + j]); }
    F held[NOUT];                                                                        // the previous column's outputs, stored behind the next barrier
    F *hcol = nullptr;
    uint32_t it = 0;
    for (uint32_t c = blockIdx.x; c < ncols; c += gridDim.x, it++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                 // my pieces of this column's window have landed
        __builtin_amdgcn_s_barrier();                                                      // ... everyone's have, and everyone is done with the other buffer
        if (hcol) {
#pragma unroll
            for (int j = 0; j < NOUT; j++) if (oi[j] != 0xFFFFFFFFu) stF(hcol + out_off + oi[j], held[j]);
            for (uint32_t i = z_lo + threadIdx.x; i < z_hi; i += LANES) stF(hcol + i, fmake(0));
        }
        if (NLOAD == 0 || CP > 0) { const uint32_t cn = c + gridDim.x; if (cn < ncols) issue(cn, (it + 1) & 1); }
        const unsigned char *win = lds_raw + (it & 1) * win_bytes;
#pragma unroll
        for (int j = 0; j < NOUT; j++) {
            Acc96 rl = {0, 0}, rh = {0, 0}, il = {0, 0}, ih = {0, 0};
#pragma unroll
            for (int g = 0; g < CAP[j]; g += 4) {
                if ((uint32_t)g < W[j]) {                                                  // wave-uniform
                    uint4 x[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int k = BASE[j] + g + u;
                        const uint32_t off = (k & 1) ? (o[k >> 1] >> 16) : (o[k >> 1] & 0xFFFFu);
                        x[u] = *reinterpret_cast<const uint4 *>(win + off);
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) acc96_mad4(rl, rh, il, ih, w[BASE[j] + g + u], x[u]);
                }
            }
            held[j] = fmake(acc_fold(rl, rh), acc_fold(il, ih));
        }
        hcol = tensor + (size_t)c * ld;
    }
    if (hcol) {
#pragma unroll
        for (int j = 0; j < NOUT; j++) if (oi[j] != 0xFFFFFFFFu) stF(hcol + out_off + oi[j], held[j]);
        for (uint32_t i = z_lo + threadIdx.x; i < z_hi; i += LANES) stF(hcol + i, fmake(0));
    }
}
template <int NOUT, int CAP0, int CAP1, int CAP2, int NW, int NLOAD, int CP = 0>
static int launch_enc_fat(hobbit_ctx *ctx, const char *name, const FatStep &f, F *tensor, size_t ld, size_t ncols, uint32_t z_lo, uint32_t z_hi, uint32_t wgs) {
    const size_t lds = (size_t)2 * ((f.in_len + 63) / 64) * 1024;
    auto kern = k_enc_fat<NOUT, CAP0, CAP1, CAP2, NW, NLOAD, CP>;
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const size_t grid = std::min(ncols, (size_t)wgs);
    HB_LAUNCH(ctx, name, kern, dim3((unsigned)grid), dim3((NW + NLOAD) * 64), lds, tensor, ld, (uint32_t)ncols, f.in_off, f.in_len, f.out_off, z_lo, z_hi,
              f.d_wt, f.d_ot, f.d_oidx, f.d_w);
    return 0;
}

// The narrow dependent steps between the first and the last one (hobbit_ctx.hpp MidCode), persistent: one workgroup of MID_WAVES waves per CU
// walks the columns.  Each step hands an output to 2^lg adjacent lanes that split its in-edges, hold their share of the records in registers,
// accumulate unreduced, fold once and combine by shuffles; one barrier per step.  x_1 of the next column arrives by LDS-DMA into the other
// window buffer meanwhile, and the finished [x_2 .. z_1] of this column leaves one iteration late (held in a register per lane), so that the
// vmcnt(0) which retires the DMA never waits for a store just issued.  (The one-workgroup-per-column k_encode spends 2.6 ms on these 21 % of
// the edges at 2^28: descriptor -> record -> gather -> fold -> barrier chains of ~4 K cycles per step and column.)
struct MidArgs { uint32_t nsteps, win_off, win_len, in_len, st_lo, lg[MID_MAX_STEPS], R[MID_MAX_STEPS], out_rel[MID_MAX_STEPS]; };
__global__ void __launch_bounds__(MID_WAVES * 64)
k_enc_mid(F *__restrict__ tensor, size_t ld, uint32_t ncols, MidArgs a, const uint32_t *__restrict__ wt, const uint32_t *__restrict__ ot,
          const uint32_t *__restrict__ oidx, const uint32_t *__restrict__ wid) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int NW = MID_WAVES, LANES = NW * 64, NS = MID_MAX_STEPS;
    constexpr int CAP[NS] = {(int)MID_CAP[0], (int)MID_CAP[1], (int)MID_CAP[2], (int)MID_CAP[3], (int)MID_CAP[4], (int)MID_CAP[5]};
    constexpr int BASE[NS] = {0, CAP[0], CAP[0] + CAP[1], CAP[0] + CAP[1] + CAP[2], CAP[0] + CAP[1] + CAP[2] + CAP[3], CAP[0] + CAP[1] + CAP[2] + CAP[3] + CAP[4]};
    constexpr int TOT = BASE[5] + CAP[5];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint32_t pieces = (a.in_len + 63) >> 6, win_bytes = ((a.win_len + 63) >> 6) << 10;
    auto issue = [&](uint32_t c, uint32_t buf) {
        const F *colp = tensor + (size_t)c * ld + a.win_off;
        for (uint32_t p = wave; p < pieces; p += NW) {
            uint32_t i = (p << 6) + lane; i = i < a.in_len ? i : a.in_len - 1;          // (pad lanes land on slots the first step overwrites later)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(colp + i),
                                             (__attribute__((address_space(3))) void *)(lds_raw + buf * win_bytes + (p << 10)), 16, 0, 0);
        }
    };
    if (blockIdx.x < ncols) issue(blockIdx.x, 0);
    uint32_t w[TOT], o[TOT / 2], oi[NS], W[NS];
#pragma unroll
    for (int k = 0; k < TOT; k++) w[k] = wt[(size_t)k * LANES + threadIdx.x];
#pragma unroll
    for (int k = 0; k < TOT / 2; k++) o[k] = ot[(size_t)k * LANES + threadIdx.x];
#pragma unroll
    for (int s = 0; s < NS; s++) {
        oi[s] = (uint32_t)s < a.nsteps ? oidx[(size_t)s * LANES + threadIdx.x] : 0xFFFFFFFFu;
        W[s] = (uint32_t)s < a.nsteps ? __builtin_amdgcn_readfirstlane(wid[wave * a.nsteps + s]) : 0u;
    }
    F held = fmake(0); F *hcol = nullptr;
    const uint32_t my_st = a.st_lo + threadIdx.x;                                        // the window element this lane carries back to memory
    uint32_t it = 0;
    for (uint32_t c = blockIdx.x; c < ncols; c += gridDim.x, it++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (hcol && my_st < a.win_len) stF(hcol + a.win_off + my_st, held);
        const uint32_t cn = c + gridDim.x;
        if (cn < ncols) issue(cn, (it + 1) & 1);
        unsigned char *win = lds_raw + (it & 1) * win_bytes;
#pragma unroll
        for (int s = 0; s < NS; s++) {
            if ((uint32_t)s < a.nsteps) {
                Acc96 rl = {0, 0}, rh = {0, 0}, il = {0, 0}, ih = {0, 0};
#pragma unroll
                for (int g = 0; g < CAP[s]; g += 2) {
                    if ((uint32_t)g < W[s]) {
                        uint4 x[2];
#pragma unroll
                        for (int u = 0; u < 2; u++) {
                            const int k = BASE[s] + g + u;
                            const uint32_t off = (k & 1) ? (o[k >> 1] >> 16) : (o[k >> 1] & 0xFFFFu);
                            x[u] = *reinterpret_cast<const uint4 *>(win + off);
                        }
#pragma unroll
                        for (int u = 0; u < 2; u++) acc96_mad4(rl, rh, il, ih, w[BASE[s] + g + u], x[u]);
                    }
                }
                F v = fmake(acc_fold(rl, rh), acc_fold(il, ih));
                const uint32_t lg = a.lg[s];
                if (lg > 0) v = fadd(v, shfl_xor_F(v, 1));
                if (lg > 1) v = fadd(v, shfl_xor_F(v, 2));
                if (lg > 2) v = fadd(v, shfl_xor_F(v, 4));
                if (lg > 3) v = fadd(v, shfl_xor_F(v, 8));
                if (lg > 4) v = fadd(v, shfl_xor_F(v, 16));
                if (oi[s] != 0xFFFFFFFFu) stF(reinterpret_cast<F *>(win) + a.out_rel[s] + oi[s], v);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
        if (my_st < a.win_len) held = ldF(reinterpret_cast<const F *>(win) + my_st);
        hcol = tensor + (size_t)c * ld;
    }
    if (hcol && my_st < a.win_len) stF(hcol + a.win_off + my_st, held);
}
static int launch_enc_mid(hobbit_ctx *ctx, const MidCode &m, F *tensor, size_t ld, size_t ncols, uint32_t wgs) {
    if (m.win_len - m.st_lo > MID_WAVES * 64) return ctx->fail(HOBBIT_EINVAL, "encode: middle window longer than the workgroup");
    MidArgs a{}; a.nsteps = m.nsteps; a.win_off = m.win_off; a.win_len = m.win_len; a.in_len = m.in_len; a.st_lo = m.st_lo;
    for (uint32_t s = 0; s < MID_MAX_STEPS; s++) { a.lg[s] = m.lg[s]; a.R[s] = m.R[s]; a.out_rel[s] = m.out_rel[s]; }
    const size_t lds = (size_t)2 * ((m.win_len + 63) / 64) * 1024;
    hipFuncSetAttribute((const void *)k_enc_mid, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const size_t grid = std::min(ncols, (size_t)wgs);
    HB_LAUNCH(ctx, "k_enc_mid", k_enc_mid, dim3((unsigned)grid), dim3(MID_WAVES * 64), lds, tensor, ld, (uint32_t)ncols, a, m.d_wt, m.d_ot, m.d_oidx, m.d_w);
    return 0;
}

int launch_encode(hobbit_ctx *ctx, const F *src, size_t ld_src, F *dst, size_t ld_dst, long long n, size_t batch, int write_msg) {
    DeviceCode &c = ctx->code;
    if (c.n != n) return ctx->fail(HOBBIT_ESTATE, "encode: graphs for this n are not finalized (hobbit_graph_finalize)");
    if (batch == 0) return 0;
    if ((size_t)c.len * 16 > 160 * 1024) return ctx->fail(HOBBIT_EINVAL, "encode: codeword does not fit in 160 KB of LDS (n <= 4096 supported)");
    const uint32_t nn = (uint32_t)n, nsteps = (uint32_t)c.steps.size();
    const bool split = nsteps >= 2 && (size_t)c.len * 16 > 80 * 1024;     // cannot co-schedule two workgroups per CU otherwise
    if (!split) {
        EncPass ps = {0, 0, nn, 0, nsteps, write_msg ? 0u : nn, 2 * nn, 0};
        uint32_t block = block_for(c, 0, nsteps, 16);
        return c.small_weights ? launch_encode_pass<true>(ctx, "k_encode", src, ld_src, dst, ld_dst, batch, ps, (uint32_t)c.len, block)
                               : launch_encode_pass<false>(ctx, "k_encode_fullw", src, ld_src, dst, ld_dst, batch, ps, (uint32_t)c.len, block);
    }
    // pass A: x_1 = C_0 x_0, outputs straight to the column in global memory
    const uint32_t r0 = c.steps[0].out_len;
    EncPass pa = {0, 0, nn, 0, 1, write_msg ? 0u : nn, write_msg ? nn : nn, 1};
    // pass B: the remaining steps on the window [n, len)
    EncPass pb = {nn, nn, nn + r0, 1, nsteps, nn + r0, 2 * nn, 0};
    // persistent register-resident first step (in place only: the message is where the codeword goes)
    const char *fat_env = getenv("HOBBIT_ENC_FAT"); const int fat = fat_env ? atoi(fat_env) : 3;       // bit 0: C_0 fat; bit 1: D_0 fat (+ the middle steps as their own pass); bit 2: middle steps by k_enc_mid (measured slower: DESIGN.md 4)
    if (c.small_weights && (fat & 1) && c.fatA.ok && src == dst && ld_src == ld_dst && !write_msg) {
        const char *wg_env = getenv("HOBBIT_ENC_FAT_WGS");
        const uint32_t wgs = wg_env ? (uint32_t)atoi(wg_env) : 256u;
        // (CP = 2 / 4 / 6 pieces per consumer wave beside the loader were measured: 3.06 / 3.15 / 3.27 ms against 2.65 with the loader alone, same call)
        HB_TRY((launch_enc_fat<FAT_A_NOUT, FAT_A_CAP0, FAT_A_CAP1, 0, FAT_A_CONS, 1>(ctx, "k_enc_fat_A", c.fatA, dst, ld_dst, batch, 0, 0, wgs)));
        if ((fat & 2) && c.fatD.ok && nsteps >= 4) {
            // the narrow dependent steps C_1 .. D_1 on the 24 KB window [x_1 .. z_1], then D_0 in fat form (it also writes the zero tail)
            const EncStep &last = c.steps[nsteps - 1];
            EncPass pm = {nn, nn, nn + r0, 1, nsteps - 1, nn + r0, last.out_off, 0};
            const char *m2_env = getenv("HOBBIT_ENC_M2"); const int m2 = m2_env ? atoi(m2_env) : 2;
            if (m2 > 0 && nsteps >= 5) {
                // the middle as two launches (HOBBIT_ENC_M2=0: one, k_encode_M): C_1 alone -- 21 % of the edges on a 14 KB window: fat form, three workgroups per CU --
                // then the five short steps behind it with a two-wave workgroup per column on the 10 KB window [x_2 .. z_1] (sixteen of them per CU
                // hide the dependent step chains).  2^28: 2.44 ms as one launch, 0.93 + 1.05 split, 0.63 + 1.05 with C_1 fat
                const uint32_t r1 = c.steps[1].out_len, x2 = nn + r0;
                EncPass p1 = {nn, nn, nn + r0, 1, 2, x2, x2, 1};
                const char *c1_env = getenv("HOBBIT_ENC_FAT_C1");
                if (c.fatC1.ok && !(c1_env && c1_env[0] == '0')) {
                    const char *wc_env = getenv("HOBBIT_ENC_FAT_WGS_C1");
                    HB_TRY((launch_enc_fat<FAT_C1_NOUT, FAT_C1_CAP0, 0, 0, FAT_C1_CONS, 1>(ctx, "k_enc_fat_C1", c.fatC1, dst, ld_dst, batch, 0, 0, wc_env ? (uint32_t)atoi(wc_env) : 768u)));
                } else
                    HB_TRY(launch_encode_pass<true>(ctx, "k_encode_C1", dst, ld_dst, dst, ld_dst, batch, p1, r0, block_for(c, 1, 2, 8)));
                EncPass p2 = {x2, x2, x2 + r1, 2, nsteps - 1, x2 + r1, last.out_off, 0};
                HB_TRY(launch_encode_pass<true>(ctx, "k_encode_M2", dst, ld_dst, dst, ld_dst, batch, p2, last.out_off - x2, block_for(c, 2, nsteps - 1, (uint32_t)m2)));
            } else if ((fat & 4) && c.mid.ok) {
                const char *wm_env = getenv("HOBBIT_ENC_MID_WGS");
                HB_TRY(launch_enc_mid(ctx, c.mid, dst, ld_dst, batch, wm_env ? (uint32_t)atoi(wm_env) : 512u));
            } else
                HB_TRY(launch_encode_pass<true>(ctx, "k_encode_M", dst, ld_dst, dst, ld_dst, batch, pm, last.out_off - nn, block_for(c, 1, nsteps - 1, 8)));
            const char *wd_env = getenv("HOBBIT_ENC_FAT_WGS_D");
            // the zero tail [len, 2n): a caller that answers those rows as zeros itself (commit_impl: leaf chain, gathers and row reads know the
            // codeword length) asks for the rows up to the next multiple of four only -- the leaf group that straddles the codeword's end is read whole
            uint32_t z_hi = 2 * nn;
            if (ctx->enc_skip_tail) { z_hi = std::min<uint32_t>(2 * nn, ((uint32_t)c.len + 3) & ~3u); ctx->enc_tail_skipped = true; }
            return launch_enc_fat<FAT_D_NOUT, FAT_D_CAP0, FAT_D_CAP1, FAT_D_CAP2, FAT_D_CONS, 1>(ctx, "k_enc_fat_D", c.fatD, dst, ld_dst, batch, (uint32_t)c.len, z_hi,
                                                                                                     wd_env ? (uint32_t)atoi(wd_env) : 256u);
        }
        return launch_encode_pass<true>(ctx, "k_encode_B", dst, ld_dst, dst, ld_dst, batch, pb, (uint32_t)c.len - nn, block_for(c, 1, nsteps, 8));
    }
    const char *s3_env = getenv("HOBBIT_ENC_SPLIT3"); const int split3 = s3_env ? atoi(s3_env) : 0;      // (read per call: scripts/ab_commit.py alternates it in one process)
    if (c.small_weights && split3 && nsteps >= 4) {
        // EXPERIMENT: pass B as two launches -- the narrow dependent steps C_1 .. D_1 on the 24 KB window [x_1 .. z_1] (six workgroups per CU), then
        // D_0 alone: it reads that window back and streams its outputs and the zero tail to global memory
        const EncStep &last = c.steps[nsteps - 1];
        const uint32_t w_end = last.in_off + (uint32_t)(last.out_off - last.in_off);      // = out_off of D_0 = end of cw_1
        EncPass pm = {nn, nn, nn + r0, 1, nsteps - 1, nn + r0, w_end, 0};
        EncPass pd = {nn, nn, w_end, nsteps - 1, nsteps, (uint32_t)c.len, 2 * nn, 1};
        HB_TRY(launch_encode_pass<true>(ctx, "k_encode_A", src, ld_src, dst, ld_dst, batch, pa, nn, block_for(c, 0, 1, 16)));
        HB_TRY(launch_encode_pass<true>(ctx, "k_encode_M", dst, ld_dst, dst, ld_dst, batch, pm, w_end - nn, block_for(c, 1, nsteps - 1, 8)));
        return launch_encode_pass<true>(ctx, "k_encode_D", dst, ld_dst, dst, ld_dst, batch, pd, w_end - nn, block_for(c, nsteps - 1, nsteps, 8));
    }
    if (c.small_weights) {
        HB_TRY(launch_encode_pass<true>(ctx, "k_encode_A", src, ld_src, dst, ld_dst, batch, pa, nn, block_for(c, 0, 1, 16)));
        return launch_encode_pass<true>(ctx, "k_encode_B", dst, ld_dst, dst, ld_dst, batch, pb, (uint32_t)c.len - nn, block_for(c, 1, nsteps, 8));
    }
    HB_TRY(launch_encode_pass<false>(ctx, "k_encode_fullw_A", src, ld_src, dst, ld_dst, batch, pa, nn, block_for(c, 0, 1, 8)));
    return launch_encode_pass<false>(ctx, "k_encode_fullw_B", dst, ld_dst, dst, ld_dst, batch, pb, (uint32_t)c.len - nn, block_for(c, 1, nsteps, 8));
}

// ============================================================================================
// BLAKE3 / Merkle
// ============================================================================================
__device__ __forceinline__ void load16w(const void *p, uint32_t m[16]) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
#pragma unroll
    for (int i = 0; i < 4; i++) { uint4 v = q[i]; m[4 * i] = v.x; m[4 * i + 1] = v.y; m[4 * i + 2] = v.z; m[4 * i + 3] = v.w; }
}
__device__ __forceinline__ void load8w(const void *p, uint32_t m[8]) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
#pragma unroll
    for (int i = 0; i < 2; i++) { uint4 v = q[i]; m[4 * i] = v.x; m[4 * i + 1] = v.y; m[4 * i + 2] = v.z; m[4 * i + 3] = v.w; }
}
__device__ __forceinline__ void store8w(void *p, const uint32_t h[8]) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(h[0], h[1], h[2], h[3]); q[1] = make_uint4(h[4], h[5], h[6], h[7]);
}

__global__ void k_blake3_64(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t m[16], h[8];
        load16w(in + 64 * i, m); blake3_compress64(m, h); store8w(out + 32 * i, h);
    }
}
// out[i] = H( H(xyzw[i]) | prev[i] )   (src/merkle_tree.cpp:62-87)
__global__ void k_hash_md(const F *__restrict__ xyzw, const uint8_t *__restrict__ prev, uint8_t *__restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t m[16], h[8];
        load16w(xyzw + 4 * i, m); blake3_compress64(m, h);
#pragma unroll
        for (int j = 0; j < 8; j++) m[j] = h[j];
        load8w(prev + 32 * i, m + 8);
        blake3_compress64(m, h); store8w(out + 32 * i, h);
    }
}
// level kernel: cur[i] = H(prev[2i] | prev[2i + (quirk ? 0 : 1)])   (src/merkle_tree.cpp:255-287)
__global__ void k_merkle_level(const uint8_t *__restrict__ prev, uint8_t *__restrict__ cur, size_t n_cur, int quirk) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_cur; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t m[16], h[8];
        load8w(prev + 64 * i, m);
        if (quirk) {
#pragma unroll
            for (int j = 0; j < 8; j++) m[8 + j] = m[j];
        } else load8w(prev + 64 * i + 32, m + 8);
        blake3_compress64(m, h); store8w(cur + 32 * i, h);
    }
}
// the last levels (n_cur <= 1024 nodes and everything above them) in ONE workgroup: a level per launch is pure launch latency
// up there (ten launches of a few microseconds of work each, and the opening builds about twenty small trees)
__global__ void __launch_bounds__(1024) k_merkle_top(const uint8_t *__restrict__ prev, uint8_t *__restrict__ cur, uint32_t n_cur, int quirk) {
    __shared__ uint32_t buf[2][1024 * 8];
    const uint32_t tid = threadIdx.x;
    uint32_t m[16], h[8];
    if (tid < n_cur) {
        load8w(prev + 64 * (size_t)tid, m);
        if (quirk) {
#pragma unroll
            for (int j = 0; j < 8; j++) m[8 + j] = m[j];
        } else load8w(prev + 64 * (size_t)tid + 32, m + 8);
        blake3_compress64(m, h); store8w(cur + 32 * (size_t)tid, h);
#pragma unroll
        for (int j = 0; j < 8; j++) buf[0][tid * 8 + j] = h[j];
    }
    uint32_t off = n_cur; int pb = 0;
    for (uint32_t sz = n_cur / 2; sz >= 1; sz /= 2) {
        __syncthreads();
        if (tid < sz) {
#pragma unroll
            for (int j = 0; j < 8; j++) m[j] = buf[pb][(2 * tid) * 8 + j];
#pragma unroll
            for (int j = 0; j < 8; j++) m[8 + j] = quirk ? m[j] : buf[pb][(2 * tid + 1) * 8 + j];
            blake3_compress64(m, h); store8w(cur + 32 * (size_t)(off + tid), h);
#pragma unroll
            for (int j = 0; j < 8; j++) buf[pb ^ 1][tid * 8 + j] = h[j];
        }
        off += sz; pb ^= 1;
    }
}
// `g` consecutive levels in one launch: a workgroup hashes 1024 nodes of the level that has `sz` nodes (from 2048 nodes of `prev`) and then
// the 512, 256, ... nodes above them that depend on nothing else, out of LDS -- the opening builds about eight trees of 2^19 .. 2^20
// leaves, and one launch per level was nine launches of a few microseconds each per tree.  sz must be a multiple of 1024, 1 <= g <= 11;
// `cur` points at the level of sz nodes, the levels above it follow in the flat layout.
__global__ void __launch_bounds__(1024) k_merkle_sub(const uint8_t *__restrict__ prev, uint8_t *__restrict__ cur, size_t sz, int g, int quirk) {
    __shared__ uint32_t buf[2][1024 * 8];
    const uint32_t tid = threadIdx.x;
    const size_t node = (size_t)blockIdx.x * 1024 + tid;
    uint32_t m[16], h[8];
    load8w(prev + 64 * node, m);
    if (quirk) {
#pragma unroll
        for (int j = 0; j < 8; j++) m[8 + j] = m[j];
    } else load8w(prev + 64 * node + 32, m + 8);
    blake3_compress64(m, h); store8w(cur + 32 * node, h);
#pragma unroll
    for (int j = 0; j < 8; j++) buf[0][tid * 8 + j] = h[j];
    size_t off = sz, lsz = sz; uint32_t width = 1024; int pb = 0;
    for (int lv = 1; lv < g; lv++) {
        width >>= 1; lsz >>= 1;
        __syncthreads();
        if (tid < width) {
#pragma unroll
            for (int j = 0; j < 8; j++) m[j] = buf[pb][(2 * tid) * 8 + j];
#pragma unroll
            for (int j = 0; j < 8; j++) m[8 + j] = quirk ? m[j] : buf[pb][(2 * tid + 1) * 8 + j];
            blake3_compress64(m, h); store8w(cur + 32 * (off + (size_t)blockIdx.x * width + tid), h);
#pragma unroll
            for (int j = 0; j < 8; j++) buf[pb ^ 1][tid * 8 + j] = h[j];
        }
        off += lsz; pb ^= 1;
    }
}
// Our_PC leaf chain over all K chunks (src/Our_PC.cpp:162-166).  The tensor is codeword-major
// ([chunk][col][2 trs]), so the 4 field elements of leaf (j, col) are 64 contiguous bytes.  One
// thread owns one leaf and keeps its Merkle-Damgard state in registers across the chunk loop:
// the tensor is read exactly once and the leaf array is written exactly once.
// Rows at or beyond the codeword length are zero in every chunk (the expander code fills 1.72 n of the 2 n rows), so for the groups
// j >= zero_from the inner digest is the constant H(0^64) (passed in as zdig): one compression per chunk instead of two for those
// leaves (14 % of them at n = 4096).  A wavefront covers 64 consecutive j of one column, so the branch is wave-uniform but for one wave
// per column.
struct ZeroDig { uint32_t w[8]; };
template <int NL>
__global__ void __launch_bounds__(256)
k_leaf_chain(const F *__restrict__ tensor, size_t chunk_stride, int K, uint32_t cols, uint32_t half_trs, uint8_t *__restrict__ leaves, uint32_t zero_from, ZeroDig zdig) {
    // NL leaves per thread (independent hash chains interleaved for instruction-level parallelism)
    const size_t total = (size_t)cols * half_trs, per = (total + NL - 1) / NL;
    // Blocks are dealt round-robin over the 8 XCDs, and with 8 blocks per column the all-zero groups of EVERY column would land on one
    // XCD (b % 8 == 7): that XCD would idle while the other seven set the kernel's time (measured: skipping 6 % of the compressions
    // bought 0.5 %).  Rotating the block index inside each run of 8 by the run's number spreads the cheap blocks evenly.
    // The grid is capped (launch_leaf_chain): a block walks several runs of 8, and the pass number joins the rotation, or it would meet the same
    // position of every column it visits.  (The rotation permutes whole runs of 8 blocks: only when the block count is a multiple of 8.)
    const bool rot = (gridDim.x & 7) == 0 && (((per + blockDim.x - 1) / blockDim.x) & 7) == 0;
    uint32_t pass = 0;
    for (size_t vb = blockIdx.x; vb * (size_t)blockDim.x < per; vb += gridDim.x, pass++) {
        const size_t bid = rot ? (vb & ~(size_t)7) | ((vb + (vb >> 3) + pass) & 7) : vb;
        const size_t g0 = bid * (size_t)blockDim.x + threadIdx.x;
        if (g0 >= per) continue;
        uint32_t st[NL][8];
        const F *p[NL];
        size_t g[NL];
        bool zero[NL];
#pragma unroll
        for (int l = 0; l < NL; l++) {
            g[l] = g0 + (size_t)l * per;
            const size_t gg = g[l] < total ? g[l] : total - 1;      // tail lanes recompute the last leaf (not stored)
#pragma unroll
            for (int q = 0; q < 8; q++) st[l][q] = 0;
            p[l] = tensor + gg * 4;                                   // (c * 2trs + 4j) with gg = c*half_trs + j
            zero[l] = (uint32_t)(gg % half_trs) >= zero_from;
        }
        for (int i = 0; i < K; i++) {
            uint32_t m[NL][16], h[NL][8];
#pragma unroll
            for (int l = 0; l < NL; l++) {
                if (zero[l]) {
#pragma unroll
                    for (int q = 0; q < 8; q++) h[l][q] = zdig.w[q];
                } else { load16w(p[l] + (size_t)i * chunk_stride, m[l]); blake3_compress64(m[l], h[l]); }
            }
#pragma unroll
            for (int l = 0; l < NL; l++) {
#pragma unroll
                for (int q = 0; q < 8; q++) { m[l][q] = h[l][q]; m[l][8 + q] = st[l][q]; }
            }
#pragma unroll
            for (int l = 0; l < NL; l++) blake3_compress64(m[l], st[l]);
        }
#pragma unroll
        for (int l = 0; l < NL; l++)
            if (g[l] < total) {
                const uint32_t c = (uint32_t)(g[l] / half_trs), j = (uint32_t)(g[l] % half_trs);
                store8w(leaves + 32 * ((size_t)j * cols + c), st[l]);
            }
    }
}
// Multi-GPU commit by chain relay: the same chain over the K_local chunks of a rank's tensor shard, for the leaf slots
// g in [g_begin, g_begin + g_count) (slot g = col * half_trs + j: the order the shard is read in), starting from the state the previous
// rank handed over (state_in, 32 B per slot in slot order; NULL = the zero state of the first chunks) and leaving either the running
// state for the next rank (state_out, slot order: both sides of the hand-over are coalesced) or -- last rank -- the final leaves in the
// reference's leaf order j * cols + col.
__global__ void __launch_bounds__(256)
k_leaf_chain_relay(const F *__restrict__ tensor, size_t chunk_stride, int K, uint32_t cols, uint32_t half_trs, size_t g_begin, size_t g_count,
                   const uint8_t *__restrict__ state_in, uint8_t *__restrict__ state_out, uint8_t *leaves, uint32_t zero_from, ZeroDig zdig, int leaves_inout) {
    // capped grid + per-pass rotation of the block order inside each run of 8, as in k_leaf_chain
    const bool rot = (gridDim.x & 7) == 0 && (((g_count + blockDim.x - 1) / blockDim.x) & 7) == 0;
    uint32_t pass = 0;
    for (size_t vb = blockIdx.x; vb * (size_t)blockDim.x < g_count; vb += gridDim.x, pass++) {
        const size_t bid = rot ? (vb & ~(size_t)7) | ((vb + (vb >> 3) + pass) & 7) : vb;
        const size_t t = bid * (size_t)blockDim.x + threadIdx.x;
        if (t >= g_count) continue;
        const size_t g = g_begin + t;
        uint32_t st[8];
        if (state_in) load8w(state_in + 32 * t, st);
        else if (leaves_inout) load8w(leaves + 32 * ((size_t)(g % half_trs) * cols + (size_t)(g / half_trs)), st);    // continue from the leaves already there (leaf order)
        else {
#pragma unroll
            for (int q = 0; q < 8; q++) st[q] = 0;
        }
        const F *p = tensor + g * 4;
        const bool zero = (uint32_t)(g % half_trs) >= zero_from;
        for (int i = 0; i < K; i++) {
            uint32_t m[16], h[8];
            if (zero) {
#pragma unroll
                for (int q = 0; q < 8; q++) h[q] = zdig.w[q];
            } else { load16w(p + (size_t)i * chunk_stride, m); blake3_compress64(m, h); }
#pragma unroll
            for (int q = 0; q < 8; q++) { m[q] = h[q]; m[8 + q] = st[q]; }
            blake3_compress64(m, st);
        }
        if (state_out) store8w(state_out + 32 * t, st);
        if (leaves) { const uint32_t c = (uint32_t)(g / half_trs), j = (uint32_t)(g % half_trs); store8w(leaves + 32 * ((size_t)j * cols + c), st); }
    }
}
int launch_leaf_chain_relay(hobbit_ctx *ctx, const F *tensor, size_t chunk_stride, int K, uint32_t cols, uint32_t half_trs, size_t g_begin, size_t g_count,
                            const uint8_t *state_in, uint8_t *state_out, uint8_t *leaves, uint32_t zero_rows_from, int leaves_inout) {
    if (!g_count) return 0;
    ZeroDig zd; { uint32_t z[16] = {0}; blake3_compress64(z, zd.w); }
    const char *ge = getenv("HOBBIT_LEAF_GRID");
    HB_LAUNCH(ctx, "k_leaf_chain_relay", k_leaf_chain_relay, dim3(grid_for(g_count, 256, ge ? atoi(ge) : 4096)), dim3(256), 0, tensor, chunk_stride, K, cols, half_trs, g_begin,
              g_count, state_in, state_out, leaves, (zero_rows_from + 3) / 4, zd, leaves_inout);
    return 0;
}
// Multi-GPU commit, step 1: inner digests H(t[4j..4j+3][c]) of the chunks a rank
// owns, written in the reference's LEAF ORDER (leaf = j*cols + c) so that a contiguous leaf range
// is a contiguous byte range to send: out[(i*M + j*cols + c)*32].
__global__ void k_inner_digests(const F *__restrict__ tensor, size_t chunk_stride, int nchunks, uint32_t cols, uint32_t half_trs,
                                uint8_t *__restrict__ out) {
    const size_t M = (size_t)cols * half_trs, total = M * nchunks;
    for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
        const size_t i = g / M, rem = g % M;
        const uint32_t c = (uint32_t)(rem / half_trs), j = (uint32_t)(rem % half_trs);
        uint32_t m[16], h[8];
        load16w(tensor + i * chunk_stride + ((size_t)c * half_trs + j) * 4, m);
        blake3_compress64(m, h);
        store8w(out + 32 * (i * M + (size_t)j * cols + c), h);
    }
}
// step 2: Merkle-Damgard chain over the K chunks for a leaf range held in leaf order:
// leaf[p] = H(dig[K-1][p] | ... H(dig[0][p] | leaf[p]) ...), dig[i] at digests + i*stride
__global__ void k_chain_digests(const uint8_t *__restrict__ digests, size_t stride_bytes, int K, size_t m, uint8_t *__restrict__ leaves) {
    for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < m; p += (size_t)gridDim.x * blockDim.x) {
        uint32_t st[8], mm[16];
        load8w(leaves + 32 * p, st);
        for (int i = 0; i < K; i++) {
            load8w(digests + (size_t)i * stride_bytes + 32 * p, mm);
#pragma unroll
            for (int q = 0; q < 8; q++) mm[8 + q] = st[q];
            blake3_compress64(mm, st);
        }
        store8w(leaves + 32 * p, st);
    }
}
int launch_inner_digests(hobbit_ctx *ctx, const F *tensor, size_t chunk_stride, int nchunks, uint32_t cols, uint32_t half_trs, uint8_t *out) {
    size_t total = (size_t)cols * half_trs * nchunks;
    if (!total) return 0;
    HB_LAUNCH(ctx, "k_inner_digests", k_inner_digests, dim3(grid_for(total, 256, 1 << 20)), dim3(256), 0, tensor, chunk_stride, nchunks, cols, half_trs, out);
    return 0;
}
int launch_chain_digests(hobbit_ctx *ctx, const uint8_t *digests, size_t stride_bytes, int K, size_t m, uint8_t *leaves) {
    if (!m) return 0;
    HB_LAUNCH(ctx, "k_chain_digests", k_chain_digests, dim3(grid_for(m, 256, 1 << 20)), dim3(256), 0, digests, stride_bytes, K, m, leaves);
    return 0;
}
// Elastic_PC streaming commit (src/Elastic_PC.cpp:228-243): every 4th chunk the four stored tensors
// are hashed position-wise into the running leaves, leaf[p] = H( H(a|b|c|d) | leaf[p] ), p = row*cols+col.
// The call passes (ci0[counter], ci1[counter], ci2[counter++], tensor[j][k]) as arguments of ONE call;
// as built by GCC the first two are read at counter+1 -- `shift` = 1 reproduces that
// (a, b taken at p+1, zero past the end), 0 reads all four at p.  Tensors and the running leaf state
// are codeword-major (index g = col*rows2 + row) so every access is coalesced; k_elastic_finish
// permutes the state into the reference's leaf order once at the end.
__global__ void __launch_bounds__(256)
k_elastic_leaf(const F *__restrict__ t0, const F *__restrict__ t1, const F *__restrict__ t2, const F *__restrict__ t3, uint32_t rows2, uint32_t cols,
               int shift, uint8_t *__restrict__ state) {
    const size_t T = (size_t)rows2 * cols;
    for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < T; g += (size_t)gridDim.x * blockDim.x) {
        const uint32_t k = (uint32_t)(g / rows2), j = (uint32_t)(g % rows2);
        size_t gs = g; bool zero = false;
        if (shift) {
            if (k + 1 < cols) gs = g + rows2;                       // (j, k+1)
            else if (j + 1 < rows2) gs = (size_t)j + 1;             // wraps to (j+1, 0)
            else zero = true;                                       // p+1 == 4B: past the end
        }
        const F a = zero ? fmake(0) : ldF(t0 + gs), b = zero ? fmake(0) : ldF(t1 + gs), c = ldF(t2 + g), d = ldF(t3 + g);
        uint32_t m[16], h[8];
        m[0] = (uint32_t)a.re; m[1] = (uint32_t)(a.re >> 32); m[2] = (uint32_t)a.im; m[3] = (uint32_t)(a.im >> 32);
        m[4] = (uint32_t)b.re; m[5] = (uint32_t)(b.re >> 32); m[6] = (uint32_t)b.im; m[7] = (uint32_t)(b.im >> 32);
        m[8] = (uint32_t)c.re; m[9] = (uint32_t)(c.re >> 32); m[10] = (uint32_t)c.im; m[11] = (uint32_t)(c.im >> 32);
        m[12] = (uint32_t)d.re; m[13] = (uint32_t)(d.re >> 32); m[14] = (uint32_t)d.im; m[15] = (uint32_t)(d.im >> 32);
        blake3_compress64(m, h);
#pragma unroll
        for (int q = 0; q < 8; q++) m[q] = h[q];
        load8w(state + 32 * g, m + 8);
        blake3_compress64(m, h);
        store8w(state + 32 * g, h);
    }
}
// Multi-GPU streaming commit ("Elastic: groups of 4 consecutive chunks"): only the INNER digest H(c0, c1, c2, t3) of every
// position of one 4-chunk group, written in the reference's leaf order (p = j*cols + k) so that a rank's leaf range is one byte range
__global__ void __launch_bounds__(256)
k_elastic_inner(const F *__restrict__ t0, const F *__restrict__ t1, const F *__restrict__ t2, const F *__restrict__ t3, uint32_t rows2, uint32_t cols,
                int shift, uint8_t *__restrict__ out) {
    const size_t T = (size_t)rows2 * cols;
    for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < T; g += (size_t)gridDim.x * blockDim.x) {
        const uint32_t k = (uint32_t)(g / rows2), j = (uint32_t)(g % rows2);
        size_t gs = g; bool zero = false;
        if (shift) {
            if (k + 1 < cols) gs = g + rows2;
            else if (j + 1 < rows2) gs = (size_t)j + 1;
            else zero = true;
        }
        const F a = zero ? fmake(0) : ldF(t0 + gs), b = zero ? fmake(0) : ldF(t1 + gs), c = ldF(t2 + g), d = ldF(t3 + g);
        uint32_t m[16], h[8];
        m[0] = (uint32_t)a.re; m[1] = (uint32_t)(a.re >> 32); m[2] = (uint32_t)a.im; m[3] = (uint32_t)(a.im >> 32);
        m[4] = (uint32_t)b.re; m[5] = (uint32_t)(b.re >> 32); m[6] = (uint32_t)b.im; m[7] = (uint32_t)(b.im >> 32);
        m[8] = (uint32_t)c.re; m[9] = (uint32_t)(c.re >> 32); m[10] = (uint32_t)c.im; m[11] = (uint32_t)(c.im >> 32);
        m[12] = (uint32_t)d.re; m[13] = (uint32_t)(d.re >> 32); m[14] = (uint32_t)d.im; m[15] = (uint32_t)(d.im >> 32);
        blake3_compress64(m, h);
        store8w(out + 32 * ((size_t)j * cols + k), h);
    }
}
int launch_elastic_inner(hobbit_ctx *ctx, const F *t0, const F *t1, const F *t2, const F *t3, uint32_t rows2, uint32_t cols, int shift, uint8_t *out) {
    HB_LAUNCH(ctx, "k_elastic_inner", k_elastic_inner, dim3(grid_for((size_t)rows2 * cols, 256, 1 << 16)), dim3(256), 0, t0, t1, t2, t3, rows2, cols, shift, out);
    return 0;
}
__global__ void k_elastic_finish(const uint8_t *__restrict__ state, uint32_t rows2, uint32_t cols, uint8_t *__restrict__ leaves) {
    const size_t T = (size_t)rows2 * cols;
    for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < T; g += (size_t)gridDim.x * blockDim.x) {
        const uint32_t k = (uint32_t)(g / rows2), j = (uint32_t)(g % rows2);
        uint32_t h[8]; load8w(state + 32 * g, h);
        store8w(leaves + 32 * ((size_t)j * cols + k), h);
    }
}
int launch_elastic_leaf(hobbit_ctx *ctx, const F *t0, const F *t1, const F *t2, const F *t3, uint32_t rows2, uint32_t cols, int shift, uint8_t *state) {
    HB_LAUNCH(ctx, "k_elastic_leaf", k_elastic_leaf, dim3(grid_for((size_t)rows2 * cols, 256, 1 << 16)), dim3(256), 0, t0, t1, t2, t3, rows2, cols, shift, state);
    return 0;
}
int launch_elastic_finish(hobbit_ctx *ctx, const uint8_t *state, uint32_t rows2, uint32_t cols, uint8_t *leaves) {
    HB_LAUNCH(ctx, "k_elastic_finish", k_elastic_finish, dim3(grid_for((size_t)rows2 * cols, 256, 1 << 16)), dim3(256), 0, state, rows2, cols, leaves);
    return 0;
}
// shockwave_commit column digests (src/Virgo.cpp:143-151): digest of column c = root of MT_commit_Blake over the
// k entries enc[0..k)[c] (k/4 leaves of 4 elements, then the tree with the configured parent rule), k <= 64
__global__ void __launch_bounds__(256)
k_col_digest(const F *__restrict__ enc, size_t W, int k, int quirk, uint8_t *__restrict__ out) {
    for (size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x; c < W; c += (size_t)gridDim.x * blockDim.x) {
        const int leaves = k / 4;
        if (quirk) {
            // create_tree_blake's parent is H(left | left) (src/merkle_tree.cpp:275-280): the root is leaf 0 re-hashed log2(leaves)
            // times -- rows 0..3 of the column are all that reaches it.  1 + log2(leaves) compressions instead of 2*leaves - 1, in registers.
            uint32_t m[16], h[8];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const F v = ldF(enc + (size_t)e * W + c);
                m[4 * e] = (uint32_t)v.re; m[4 * e + 1] = (uint32_t)(v.re >> 32); m[4 * e + 2] = (uint32_t)v.im; m[4 * e + 3] = (uint32_t)(v.im >> 32);
            }
            blake3_compress64(m, h);
            for (int n = leaves; n > 1; n >>= 1) {
#pragma unroll
                for (int q = 0; q < 8; q++) { m[q] = h[q]; m[8 + q] = h[q]; }
                blake3_compress64(m, h);
            }
            store8w(out + 32 * c, h);
            continue;
        }
        uint32_t node[16][8];
        for (int l = 0; l < leaves; l++) {
            uint32_t m[16];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const F v = ldF(enc + (size_t)(4 * l + e) * W + c);
                m[4 * e] = (uint32_t)v.re; m[4 * e + 1] = (uint32_t)(v.re >> 32); m[4 * e + 2] = (uint32_t)v.im; m[4 * e + 3] = (uint32_t)(v.im >> 32);
            }
            blake3_compress64(m, node[l]);
        }
        for (int n = leaves / 2; n >= 1; n /= 2)
            for (int i = 0; i < n; i++) {
                uint32_t m[16];
#pragma unroll
                for (int q = 0; q < 8; q++) { m[q] = node[2 * i][q]; m[8 + q] = quirk ? node[2 * i][q] : node[2 * i + 1][q]; }
                blake3_compress64(m, node[i]);
            }
        store8w(out + 32 * c, node[0]);
    }
}
int launch_col_digest(hobbit_ctx *ctx, const F *enc, size_t W, int k, int quirk, uint8_t *out) {
    HB_LAUNCH(ctx, "k_col_digest", k_col_digest, dim3(grid_for(W, 256)), dim3(256), 0, enc, W, k, quirk, out);
    return 0;
}
// one level of change_form (src/Virgo.cpp:104-118): every block of S elements becomes [even entries | odd - even]
__global__ void k_change_form_level(const F *__restrict__ in, F *__restrict__ out, size_t n, size_t S) {
    for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < n / 2; g += (size_t)gridDim.x * blockDim.x) {
        const size_t blk = g / (S / 2), i = g % (S / 2), pos = blk * S;
        const F a = ldF(in + pos + 2 * i), b = ldF(in + pos + 2 * i + 1);
        stF(out + pos + i, a); stF(out + pos + S / 2 + i, fsub(b, a));
    }
}
// all levels with block size <= T (T a power of two <= 4096) at once: below that size the recursion stays inside one T-block, so a
// workgroup takes a T-block through LDS (in place, pairs staged in registers between the two barriers of a level)
__global__ void __launch_bounds__(256) k_change_form_tail(F *__restrict__ data, uint32_t T) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    F *s = reinterpret_cast<F *>(lds_raw);
    F *blk = data + (size_t)blockIdx.x * T;
    for (uint32_t i = threadIdx.x; i < T; i += 256) stF(&s[i], ldF(blk + i));
    __syncthreads();
    for (uint32_t S = T; S >= 2; S >>= 1) {
        F a[8], b[8];                                     // T/2 <= 2048 pairs over 256 threads
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t g = threadIdx.x + 256 * u;
            if (g < T / 2) { const uint32_t pos = (g / (S / 2)) * S, i = g % (S / 2); a[u] = ldF(&s[pos + 2 * i]); b[u] = ldF(&s[pos + 2 * i + 1]); }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t g = threadIdx.x + 256 * u;
            if (g < T / 2) { const uint32_t pos = (g / (S / 2)) * S, i = g % (S / 2); stF(&s[pos + i], a[u]); stF(&s[pos + S / 2 + i], fsub(b[u], a[u])); }
        }
        __syncthreads();
    }
    for (uint32_t i = threadIdx.x; i < T; i += 256) stF(blk + i, ldF(&s[i]));
}
int launch_change_form_tail(hobbit_ctx *ctx, F *data, size_t n, uint32_t T) {
    hipFuncSetAttribute((const void *)k_change_form_tail, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    HB_LAUNCH(ctx, "k_change_form_tail", k_change_form_tail, dim3((unsigned)(n / T)), dim3(256), (size_t)T * 16, data, T);
    return 0;
}
int launch_change_form_level(hobbit_ctx *ctx, const F *in, F *out, size_t n, size_t S) {
    HB_LAUNCH(ctx, "k_change_form_level", k_change_form_level, dim3(grid_for(n / 2, 256)), dim3(256), 0, in, out, n, S);
    return 0;
}
// paths[q][l] = levels[off_l + (pos_q >> l) ^ 1]   (src/merkle_tree.cpp:308-324)
__global__ void k_merkle_paths(const uint8_t *__restrict__ levels, size_t n, const uint64_t *__restrict__ pos, size_t nq, int depth,
                               uint8_t *__restrict__ paths) {
    size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (g >= nq * depth) return;
    size_t q = g / depth; int l = (int)(g % depth);
    size_t off = 0, sz = n;
    for (int t = 0; t < l; t++) { off += sz; sz >>= 1; }
    size_t p = (pos[q] >> l) ^ 1;
    const uint4 *s = reinterpret_cast<const uint4 *>(levels + 32 * (off + p));
    uint4 *d = reinterpret_cast<uint4 *>(paths + 32 * g);
    d[0] = s[0]; d[1] = s[1];
}

int launch_blake3_64(hobbit_ctx *ctx, const uint8_t *in, uint8_t *out, size_t n) {
    if (!n) return 0;
    HB_LAUNCH(ctx, "k_blake3_64", k_blake3_64, dim3(grid_for(n, 256)), dim3(256), 0, in, out, n);
    return 0;
}
int launch_hash_md(hobbit_ctx *ctx, const F *xyzw, const uint8_t *prev, uint8_t *out, size_t n) {
    if (!n) return 0;
    HB_LAUNCH(ctx, "k_hash_md", k_hash_md, dim3(grid_for(n, 256)), dim3(256), 0, xyzw, prev, out, n);
    return 0;
}
int launch_merkle_levels(hobbit_ctx *ctx, uint8_t *levels, size_t n, int quirk) {
    static const int sub_mode = [] { const char *e = getenv("HOBBIT_MERKLE_SUB"); return e ? atoi(e) : 1; }();
    size_t off = 0, tot = n;
    for (size_t sz = n / 2; sz >= 1;) {
        if (sz <= 1024) {                                   // this level and all above it in one workgroup
            HB_LAUNCH(ctx, "k_merkle_top", k_merkle_top, dim3(1), dim3(1024), 0, levels + 32 * off, levels + 32 * tot, (uint32_t)sz, quirk);
            break;
        }
        // levels of more than 1024 nodes: up to 11 of them per launch (every workgroup's 1024 nodes carry their own ancestors), but only
        // while the level is small enough that the idle upper levels of a workgroup cost less than the launches they replace
        int wide = 0; for (size_t t = sz; t > 1024; t >>= 1) wide++;
        if (sub_mode && sz % 1024 == 0 && sz <= ((size_t)1 << 21) && (n & (n - 1)) == 0) {
            const int g = wide < 11 ? wide : 11;
            HB_LAUNCH(ctx, "k_merkle_level", k_merkle_sub, dim3((unsigned)(sz / 1024)), dim3(1024), 0, levels + 32 * off, levels + 32 * tot, sz, g, quirk);
            for (int lv = 0; lv < g; lv++) { off = tot; tot += sz; sz >>= 1; }
            continue;
        }
        HB_LAUNCH(ctx, "k_merkle_level", k_merkle_level, dim3(grid_for(sz, 256)), dim3(256), 0, levels + 32 * off, levels + 32 * tot, sz, quirk);
        off = tot; tot += sz; sz >>= 1;
    }
    return 0;
}
// zero_rows_from: first row index that is zero in EVERY chunk (the expander codeword length; 2*half_trs = none)
int launch_leaf_chain(hobbit_ctx *ctx, const F *tensor, size_t chunk_stride, int K, uint32_t cols, uint32_t half_trs, uint8_t *leaves, uint32_t zero_rows_from) {
    size_t total = (size_t)cols * half_trs;
    constexpr int NL = 1;   // 2 interleaved chains measured no faster: the kernel sits at the VALU issue limit (profiles/r01_microbench.txt)
    ZeroDig zd; { uint32_t z[16] = {0}; blake3_compress64(z, zd.w); }
    const char *e = getenv("HOBBIT_LEAF_ZERO_SKIP");
    const uint32_t zero_from = (e && e[0] == '0') ? 0xFFFFFFFFu : (zero_rows_from + 3) / 4;
    // 4096 resident-ish workgroups walking the leaves instead of one workgroup per 256 leaves (32 768 at 2^28): 10.15 -> 9.65 ms, same call
    // (2048: 9.7, 8192: 9.7; HOBBIT_LEAF_GRID for the A/B) -- fewer dispatches, and the per-pass rotation spreads the cheap all-zero groups evenly
    const char *ge = getenv("HOBBIT_LEAF_GRID");
    HB_LAUNCH(ctx, "k_leaf_chain", k_leaf_chain<NL>, dim3(grid_for((total + NL - 1) / NL, 256, ge ? atoi(ge) : 4096)), dim3(256), 0, tensor, chunk_stride, K, cols,
              half_trs, leaves, zero_from, zd);
    return 0;
}
int launch_merkle_paths(hobbit_ctx *ctx, const uint8_t *levels, size_t n, const uint64_t *d_pos, size_t nq, int depth, uint8_t *d_paths) {
    size_t total = nq * depth;
    if (!total) return 0;
    HB_LAUNCH(ctx, "k_merkle_paths", k_merkle_paths, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, levels, n, d_pos, nq, depth, d_paths);
    return 0;
}

// ============================================================================================
// eq table / aggregate / gather
// ============================================================================================
// one doubling step of precompute_beta (src/utils.cpp:251-296): new[2j] = old[j] - r*old[j]; new[2j+1] = r*old[j]
__global__ void k_eq_step(const F *__restrict__ old, F *__restrict__ nw, size_t m, F r) {
    for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < m; j += (size_t)gridDim.x * blockDim.x) {
        F o = ldF(old + j), t = fmul(r, o);
        stF(nw + 2 * j, fsub(o, t)); stF(nw + 2 * j + 1, t);
    }
}
// two doubling steps in one pass (levels with challenges ra, then rb): old[j] -> new[4j .. 4j+3]; the intermediate level is never stored
__global__ void k_eq_step2(const F *__restrict__ old, F *__restrict__ nw, size_t m, F ra, F rb) {
    for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < m; j += (size_t)gridDim.x * blockDim.x) {
        const F o = ldF(old + j), t = fmul(ra, o), x0 = fsub(o, t);
        const F u0 = fmul(rb, x0), u1 = fmul(rb, t);
        stF(nw + 4 * j, fsub(x0, u0)); stF(nw + 4 * j + 1, u0); stF(nw + 4 * j + 2, fsub(t, u1)); stF(nw + 4 * j + 3, u1);
    }
}
// the first h <= 12 doubling steps in ONE workgroup (table in LDS, pairs staged in registers between the two barriers of a level):
// up there a level per launch is a dozen launches of a few hundred nanoseconds of work each.  r.b[i] = challenge of level i.
struct EqHead { F b[12]; };
__global__ void __launch_bounds__(256) k_eq_head(EqHead r, int h, F *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    F *s = reinterpret_cast<F *>(lds_raw);
    if (threadIdx.x == 0) s[0] = fmake(1);
    __syncthreads();
    for (int i = 0; i < h; i++) {
        const uint32_t m = 1u << i;
        F o[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const uint32_t j = threadIdx.x + 256 * u; if (j < m) o[u] = ldF(&s[j]); }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t j = threadIdx.x + 256 * u;
            if (j < m) { const F t = fmul(r.b[i], o[u]); stF(&s[2 * j], fsub(o[u], t)); stF(&s[2 * j + 1], t); }
        }
        __syncthreads();
    }
    for (uint32_t j = threadIdx.x; j < (1u << h); j += 256) stF(out + j, ldF(&s[j]));
}
// last doubling step of TWO tables fused with their combination: out = eq(r1) + a * eq(r2)  (src/PC_utils.cpp:344-349), from the
// two half-size tables o1, o2 and the last challenges c1, c2 -- the full-size tables are never written and read back
__global__ void k_eq_final_axpy(const F *__restrict__ o1, const F *__restrict__ o2, size_t m, F c1, F c2, F a, F *__restrict__ out) {
    for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < m; j += (size_t)gridDim.x * blockDim.x) {
        const F x = ldF(o1 + j), y = ldF(o2 + j), t1 = fmul(c1, x), t2 = fmul(c2, y);
        stF(out + 2 * j, fadd(fsub(x, t1), fmul(a, fsub(y, t2))));
        stF(out + 2 * j + 1, fadd(t1, fmul(a, t2)));
    }
}
int launch_eq_table(hobbit_ctx *ctx, CHP h_r, int k, F *d_out) {
    // head in one workgroup, then ping-pong between d_out (final) and workspace so that the last step lands in d_out
    size_t n = (size_t)1 << k;
    F *tmp = nullptr;
    const int h = k < 12 ? k : 12, rest = k - h;
    if (rest > 0) HB_TRY(ctx->workspace(n / 2 * sizeof(F), (void **)&tmp));
    const int launches = rest / 2 + rest % 2;                       // one single step if `rest` is odd, then two levels per pass
    F *cur = (launches % 2 == 0) ? d_out : tmp;
    EqHead hd;
    for (int i = 0; i < 12; i++) hd.b[i] = i < h ? h_r[k - 1 - i] : fmake(0);
    static std::atomic<uint64_t> attr_set{0};
    set_lds_limit_once(ctx, (const void *)k_eq_head, 65536, attr_set);
    HB_LAUNCH(ctx, "k_eq_head", k_eq_head, dim3(1), dim3(256), ((size_t)16 << h), hd, h, cur);
    for (int i = h; i < k;) {
        F *nxt = cur == d_out ? tmp : d_out;
        size_t m = (size_t)1 << i;
        if ((k - i) % 2) { HB_LAUNCH(ctx, "k_eq_step", k_eq_step, dim3(grid_for(m, 256)), dim3(256), 0, cur, nxt, m, h_r[k - 1 - i]); i += 1; }
        else { HB_LAUNCH(ctx, "k_eq_step", k_eq_step2, dim3(grid_for(m, 256)), dim3(256), 0, cur, nxt, m, h_r[k - 1 - i], h_r[k - 2 - i]); i += 2; }
        cur = nxt;
    }
    return 0;
}
// d_out[0..2^k) = eq(r1) + a * eq(r2); d_half: scratch of 2^k elements (the two half-size tables)
int launch_eq_pair_axpy(hobbit_ctx *ctx, CHP h_r1, CHP h_r2, int k, F a, F *d_half, F *d_out) {
    if (k < 1) return ctx->fail(HOBBIT_EINVAL, "eq_pair_axpy: k must be >= 1");
    const size_t m = (size_t)1 << (k - 1);
    HB_TRY(launch_eq_table(ctx, h_r1 + 1, k - 1, d_half));             // levels 0..k-2 use r[k-1] .. r[1]; the last level uses r[0]
    HB_TRY(launch_eq_table(ctx, h_r2 + 1, k - 1, d_half + m));
    HB_LAUNCH(ctx, "k_eq_final_axpy", k_eq_final_axpy, dim3(grid_for(m, 256)), dim3(256), 0, d_half, d_half + m, m, h_r1[0], h_r2[0], a, d_out);
    return 0;
}
// aggr[j] = sum_i beta[i] * poly[i*M + j]   (src/Our_PC.cpp:258-272)
__global__ void k_aggregate_dev(const F *__restrict__ poly, size_t M, int K, const F *__restrict__ beta, F *__restrict__ aggr) {
    for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < M; j += (size_t)gridDim.x * blockDim.x) {
        F acc = fmake(0);
        for (int i = 0; i < K; i++) acc = fadd(acc, fmul(ldF(beta + i), ldF(poly + (size_t)i * M + j)));
        stF(aggr + j, acc);
    }
}
// the same with the K <= 64 coefficients passed by value in the kernel arguments (1 KB): no host buffer, no copy, no synchronisation
struct AggCoef { F b[64]; };
__global__ void k_aggregate_arg(const F *__restrict__ poly, size_t M, int K, AggCoef beta, F *__restrict__ aggr) {
    for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < M; j += (size_t)gridDim.x * blockDim.x) {
        F acc = fmake(0);
        for (int i = 0; i < K; i++) acc = fadd(acc, fmul(beta.b[i], ldF(poly + (size_t)i * M + j)));
        stF(aggr + j, acc);
    }
}
int launch_aggregate(hobbit_ctx *ctx, const F *poly, size_t M, int K, CHP h_beta, F *aggr) {
    if (K <= 64) {
        AggCoef cf;
        for (int i = 0; i < 64; i++) cf.b[i] = i < K ? h_beta[i] : fmake(0);
        HB_LAUNCH(ctx, "k_aggregate", k_aggregate_arg, dim3(grid_for(M, 256)), dim3(256), 0, poly, M, K, cf, aggr);
        return 0;
    }
    // more chunks than fit the argument block: the coefficients are read by the kernel straight from a small pinned
    // (device-visible) host buffer
    F *pc; HB_TRY(ctx->pinned_const((size_t)K * sizeof(F), (void **)&pc));
    for (int i = 0; i < K; i++) pc[i] = h_beta[i];
    HB_LAUNCH(ctx, "k_aggregate", k_aggregate_dev, dim3(grid_for(M, 256)), dim3(256), 0, poly, M, K, pc, aggr);
    return 0;
}
// reply[q*K + i] = tensor[i][col_q][row_q]   (src/Our_PC.cpp:291-305, codeword-major tensor)
// rows >= rows_valid of a column are zero BY CONSTRUCTION and may never have been written (a commitment's RS x expander tensor past its codeword
// length: commit_impl): they are answered as zero, not read
__global__ void k_gather(const F *__restrict__ tensor, size_t chunk_stride, uint32_t rows2, int K, const uint32_t *__restrict__ rows,
                         const uint32_t *__restrict__ cols, size_t nq, F *__restrict__ reply, uint32_t rows_valid) {
    size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (g >= nq * K) return;
    size_t q = g / K; int i = (int)(g % K);
    stF(reply + g, rows[q] < rows_valid ? ldF(tensor + (size_t)i * chunk_stride + (size_t)cols[q] * rows2 + rows[q]) : fmake(0));
}
int launch_gather(hobbit_ctx *ctx, const F *tensor, size_t chunk_stride, uint32_t rows2, int K, const uint32_t *d_rows, const uint32_t *d_cols,
                  size_t nq, F *d_reply, uint32_t rows_valid) {
    size_t total = nq * K;
    if (!total) return 0;
    HB_LAUNCH(ctx, "k_gather", k_gather, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, tensor, chunk_stride, rows2, K, d_rows, d_cols, nq, d_reply,
              rows_valid);
    return 0;
}
// the unwritten tails made real: rows [rows_valid, rows2) of every column := 0 (before a raw pointer to the tensor leaves the library)
__global__ void k_zero_rows(F *__restrict__ tensor, size_t ncols, uint32_t rows2, uint32_t rows_valid) {
    const uint32_t span = rows2 - rows_valid;
    for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < ncols * span; g += (size_t)gridDim.x * blockDim.x)
        stF(tensor + (g / span) * rows2 + rows_valid + (g % span), fmake(0));
}
int launch_zero_rows(hobbit_ctx *ctx, F *tensor, size_t ncols, uint32_t rows2, uint32_t rows_valid) {
    if (rows_valid >= rows2 || !ncols) return 0;
    HB_LAUNCH(ctx, "k_zero_rows", k_zero_rows, dim3(grid_for(ncols * (rows2 - rows_valid), 256, 8192)), dim3(256), 0, tensor, ncols, rows2, rows_valid);
    return 0;
}
// row `row` of one chunk in the reference's row-major order: out[c] = tensor[c*rows2 + row]
__global__ void k_tensor_row(const F *__restrict__ chunk, uint32_t rows2, uint32_t cols, uint32_t row, F *__restrict__ out, uint32_t rows_valid) {
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < cols) stF(out + c, row < rows_valid ? ldF(chunk + (size_t)c * rows2 + row) : fmake(0));
}
int launch_tensor_row(hobbit_ctx *ctx, const F *chunk, uint32_t rows2, uint32_t cols, uint32_t row, F *d_out, uint32_t rows_valid) {
    HB_LAUNCH(ctx, "k_tensor_row", k_tensor_row, dim3((cols + 255) / 256), dim3(256), 0, chunk, rows2, cols, row, d_out, rows_valid);
    return 0;
}
// multilinear evaluation fold step of evaluate_vector (src/utils.cpp:789-802): v'[j] = (1-r) v[2j] + r v[2j+1]
__global__ void k_eval_fold(const F *__restrict__ v, F *__restrict__ o, size_t L, F r) {
    for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < L; j += (size_t)gridDim.x * blockDim.x) {
        F a = ldF(v + 2 * j), b = ldF(v + 2 * j + 1);
        stF(o + j, fadd(a, fmul(r, fsub(b, a))));
    }
}
int launch_eval_fold(hobbit_ctx *ctx, const F *v, F *o, size_t L, F r) {
    HB_LAUNCH(ctx, "k_eval_fold", k_eval_fold, dim3(grid_for(L, 256)), dim3(256), 0, v, o, L, r);
    return 0;
}
// Two levels per launch (64 contiguous bytes per lane, the access pattern of k_sc2_double's fold): o[j] from v[4j .. 4j+3] with (r0, r1),
// level by level in the same order as two k_eval_fold launches -- the same field operations, half the launches, 2/3 of the traffic.
__global__ void k_eval_fold2(const F *__restrict__ v, F *__restrict__ o, size_t L, F r0, F r1) {
    for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < L; j += (size_t)gridDim.x * blockDim.x) {
        const F a = ldF(v + 4 * j), b = ldF(v + 4 * j + 1), c = ldF(v + 4 * j + 2), d = ldF(v + 4 * j + 3);
        const F x = fadd(a, fmul(r0, fsub(b, a))), y = fadd(c, fmul(r0, fsub(d, c)));
        stF(o + j, fadd(x, fmul(r1, fsub(y, x))));
    }
}
// The last levels (n <= 4096 elements) in one workgroup: the table lives in LDS, one barrier per level.
struct EvalTailR { F r[12]; };
__global__ void __launch_bounds__(1024) k_eval_tail(const F *__restrict__ v, F *__restrict__ o, uint32_t n, int levels, EvalTailR rr) {
    __shared__ F t[4096];
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) t[i] = ldF(v + i);
    __syncthreads();
    for (int l = 0; l < levels; l++) {
        const uint32_t L = n >> (l + 1);
        F out[2]; int cnt = 0;
        for (uint32_t j = threadIdx.x; j < L; j += blockDim.x) { const F a = t[2 * j], b = t[2 * j + 1]; out[cnt++] = fadd(a, fmul(rr.r[l], fsub(b, a))); }
        __syncthreads();
        cnt = 0;
        for (uint32_t j = threadIdx.x; j < L; j += blockDim.x) t[j] = out[cnt++];
        __syncthreads();
    }
    if (threadIdx.x == 0) stF(o, t[0]);
}
int launch_eval_fold2(hobbit_ctx *ctx, const F *v, F *o, size_t L, F r0, F r1) {
    HB_LAUNCH(ctx, "k_eval_fold", k_eval_fold2, dim3(grid_for(L, 256)), dim3(256), 0, v, o, L, r0, r1);
    return 0;
}
int launch_eval_tail(hobbit_ctx *ctx, const F *v, F *o, size_t n, int levels, const F *r) {
    if (n > 4096 || levels > 12 || ((size_t)1 << levels) != n) return ctx->fail(HOBBIT_EINVAL, "eval_tail: at most 4096 elements");
    EvalTailR rr; for (int i = 0; i < levels; i++) rr.r[i] = r[i];
    HB_LAUNCH(ctx, "k_eval_fold", k_eval_tail, dim3(1), dim3(1024), 0, v, o, (uint32_t)n, levels, rr);
    return 0;
}

// ============================================================================================
// code-membership / FFT-as-sumcheck tables
// ============================================================================================
// A = H^T beta (src/sumcheck.cpp:2888-2929) as one CSR gather: A[a] = sum_e w_e * beta[idx_e]
// (the "-= beta" identity terms are edges of weight -1)
__global__ void k_csr_gather(const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ idx, const F *__restrict__ w,
                             const F *__restrict__ x, F *__restrict__ y, size_t rows) {
    for (size_t a = blockIdx.x * (size_t)blockDim.x + threadIdx.x; a < rows; a += (size_t)gridDim.x * blockDim.x) {
        F acc = fmake(0);
        for (uint32_t e = rowptr[a]; e < rowptr[a + 1]; e++) acc = fadd(acc, fmul(ldF(x + idx[e]), ldF(w + e)));
        stF(y + a, acc);
    }
}
int launch_csr_gather(hobbit_ctx *ctx, const uint32_t *rowptr, const uint32_t *idx, const F *w, const F *x, F *y, size_t rows) {
    if (!rows) return 0;
    HB_LAUNCH(ctx, "k_csr_gather", k_csr_gather, dim3(grid_for(rows, 256)), dim3(256), 0, rowptr, idx, w, x, y, rows);
    return 0;
}
// one doubling level of phiGInit (src/utils.cpp:694-755): for b < half, l = b, r = b ^ half:
//   t2 = rx * pm[b << m];  g[r] = g[l] * ((1-rx) - t2);  g[l] = g[l] * ((1-rx) + t2)
// last_only (the non-inverse table's closing loop) updates g[l] alone.
__global__ void k_phi_step(F *__restrict__ g, size_t half, int m, F rx, const F *__restrict__ pm, int last_only) {
    const F t1 = fsub(fmake(1), rx);
    for (size_t b = blockIdx.x * (size_t)blockDim.x + threadIdx.x; b < half; b += (size_t)gridDim.x * blockDim.x) {
        const F t2 = fmul(rx, ldF(pm + (b << m))), gl = ldF(g + b);
        if (!last_only) stF(g + (b ^ half), fmul(gl, fsub(t1, t2)));
        stF(g + b, fmul(gl, fadd(t1, t2)));
    }
}
// levels 1..h (h <= 11) of the forward table in one workgroup: g[0] = scale, then h in-place doubling levels in LDS.  rx.b[i-1] = the
// challenge of level i (= h_rx[n - i]).
__global__ void __launch_bounds__(256) k_phi_head(F *__restrict__ g, int n, int h, EqHead rx, F scale, const F *__restrict__ pm) {
    __shared__ F s[2048];
    if (threadIdx.x == 0) s[0] = scale;
    __syncthreads();
    for (int i = 1; i <= h; i++) {
        const uint32_t half = 1u << (i - 1);
        const F r = rx.b[i - 1], t1 = fsub(fmake(1), r);
        for (uint32_t b = threadIdx.x; b < half; b += 256) {
            const F t2 = fmul(r, ldF(pm + ((size_t)b << (n - i)))), gl = ldF(&s[b]);
            stF(&s[b ^ half], fmul(gl, fsub(t1, t2)));
            stF(&s[b], fmul(gl, fadd(t1, t2)));
        }
        __syncthreads();
    }
    for (uint32_t j = threadIdx.x; j < (1u << h); j += 256) stF(g + j, ldF(&s[j]));
}
int launch_phi_head(hobbit_ctx *ctx, F *g, int n, int h, CHP h_rx, F scale, const F *pm) {
    if (h < 1 || h > 11 || h >= n) return ctx->fail(HOBBIT_EINVAL, "phi_head: 1 <= h <= min(11, n - 1)");
    EqHead rx;
    for (int i = 1; i <= 12; i++) rx.b[i - 1] = i <= h ? h_rx[n - i] : fmake(0);
    HB_LAUNCH(ctx, "k_phi_step", k_phi_head, dim3(1), dim3(256), 0, g, n, h, rx, scale, pm);
    return 0;
}
int launch_phi_step(hobbit_ctx *ctx, F *g, size_t half, int m, F rx, const F *pm, int last_only) {
    HB_LAUNCH(ctx, "k_phi_step", k_phi_step, dim3(grid_for(half, 256)), dim3(256), 0, g, half, m, rx, pm, last_only);
    return 0;
}
// prepare_matrix(transpose(M), r) step (src/utils.cpp:758-775): fold adjacent ROWS of a row-major matrix
__global__ void k_fold_rows(const F *__restrict__ in, F *__restrict__ out, size_t out_rows, size_t cols, F r) {
    const size_t total = out_rows * cols;
    for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
        const size_t j = g / cols, c = g % cols;
        const F a = ldF(in + (2 * j) * cols + c), b = ldF(in + (2 * j + 1) * cols + c);
        stF(out + g, fadd(a, fmul(r, fsub(b, a))));
    }
}
int launch_fold_rows(hobbit_ctx *ctx, const F *in, F *out, size_t out_rows, size_t cols, F r) {
    if (!out_rows || !cols) return 0;
    HB_LAUNCH(ctx, "k_fold_rows", k_fold_rows, dim3(grid_for(out_rows * cols, 256)), dim3(256), 0, in, out, out_rows, cols, r);
    return 0;
}

// ---- dense helpers of recursive_prover_Spielman (src/PC_utils.cpp:293-345) -------------------
// out[i] = sum_j v[j] * M[i][j]   (one workgroup per row; aggr_c = [M' | C] . s)
__global__ void __launch_bounds__(256) k_matvec_rows(const F *__restrict__ Mx, size_t cols, const F *__restrict__ v, F *__restrict__ out) {
    const F *row = Mx + (size_t)blockIdx.x * cols;
    F c[1] = {fmake(0)};
    for (size_t j = threadIdx.x; j < cols; j += blockDim.x) c[0] = fadd(c[0], fmul(ldF(v + j), ldF(row + j)));
    __shared__ F red[4];
    F sm = wave_sum(c[0]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sm;
    __syncthreads();
    if (threadIdx.x == 0) stF(out + blockIdx.x, fadd(fadd(red[0], red[1]), fadd(red[2], red[3])));
}
// out[i] = sum_j v[j] * M[j][i]   (evals = beta^T [M' | C]): lanes run along i (coalesced), the row range is
// split over blockIdx.y into partial sums that a second launch adds up
__global__ void __launch_bounds__(256) k_vecmat(const F *__restrict__ Mx, size_t rows, size_t cols, const F *__restrict__ v, F *__restrict__ part) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= cols) return;
    const size_t per = (rows + gridDim.y - 1) / gridDim.y, j0 = blockIdx.y * per, j1 = j0 + per < rows ? j0 + per : rows;
    F acc = fmake(0);
    for (size_t j = j0; j < j1; j++) acc = fadd(acc, fmul(ldF(v + j), ldF(Mx + j * cols + i)));
    stF(part + (size_t)blockIdx.y * cols + i, acc);
}
__global__ void __launch_bounds__(256) k_colsum(const F *__restrict__ part, size_t nparts, size_t cols, F *__restrict__ out) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= cols) return;
    F acc = fmake(0);
    for (size_t p = 0; p < nparts; p++) acc = fadd(acc, ldF(part + p * cols + i));
    stF(out + i, acc);
}
__global__ void k_scatter(const uint64_t *__restrict__ idx, const F *__restrict__ val, size_t n, F *__restrict__ out) {
    size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (g < n) stF(out + idx[g], ldF(val + g));
}
// y += a * x
__global__ void k_axpy(F *__restrict__ y, const F *__restrict__ x, F a, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) stF(y + i, fadd(ldF(y + i), fmul(a, ldF(x + i))));
}
int launch_matvec_rows(hobbit_ctx *ctx, const F *Mx, size_t rows, size_t cols, const F *v, F *out) {
    if (!rows) return 0;
    HB_LAUNCH(ctx, "k_matvec_rows", k_matvec_rows, dim3((unsigned)rows), dim3(256), 0, Mx, cols, v, out);
    return 0;
}
int launch_vecmat(hobbit_ctx *ctx, const F *Mx, size_t rows, size_t cols, const F *v, F *out) {
    // enough row slices to fill the chip: (cols/256) x nparts workgroups
    size_t nparts = 1;
    while (nparts < 256 && nparts * 2 <= rows && ((cols + 255) / 256) * nparts < 2048) nparts *= 2;
    F *part; HB_TRY(ctx->workspace(nparts * cols * sizeof(F), (void **)&part));
    HB_LAUNCH(ctx, "k_vecmat", k_vecmat, dim3((unsigned)((cols + 255) / 256), (unsigned)nparts), dim3(256), 0, Mx, rows, cols, v, part);
    HB_LAUNCH(ctx, "k_colsum", k_colsum, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, part, nparts, cols, out);
    return 0;
}
// out[q*m + j] = src[idx[q]*bmul + j*stride]: query replies (WHIR: 16 consecutive regrouped elements; shockwave: one column of k rows)
__global__ void k_gather_strided(const F *__restrict__ src, const uint64_t *__restrict__ idx, size_t nq, uint32_t m, size_t bmul, size_t stride, F *__restrict__ out) {
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= nq * m) return;
    size_t q = t / m, j = t % m;
    stF(out + t, ldF(src + idx[q] * bmul + j * stride));
}
int launch_gather_strided(hobbit_ctx *ctx, const F *src, const uint64_t *d_idx, size_t nq, uint32_t m, size_t bmul, size_t stride, F *out) {
    if (!nq) return 0;
    HB_LAUNCH(ctx, "k_gather_strided", k_gather_strided, dim3((unsigned)((nq * m + 255) / 256)), dim3(256), 0, src, d_idx, nq, m, bmul, stride, out);
    return 0;
}
int launch_scatter(hobbit_ctx *ctx, const uint64_t *idx, const F *val, size_t n, F *out) {
    if (!n) return 0;
    HB_LAUNCH(ctx, "k_scatter", k_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, idx, val, n, out);
    return 0;
}
int launch_axpy(hobbit_ctx *ctx, F *y, const F *x, F a, size_t n) {
    HB_LAUNCH(ctx, "k_axpy", k_axpy, dim3(grid_for(n, 256)), dim3(256), 0, y, x, a, n);
    return 0;
}

// ============================================================================================
// Sumcheck (src/sumcheck.cpp:2391-2460 two-product, 1974-2058 three-product)
//
// Per round: one streaming kernel over the tables produces per-workgroup partial sums of the round
// polynomial's coefficients, a one-workgroup kernel reduces them to 3-4 field elements, and those
// 48-64 bytes go to the HOST, which runs the MiMC transcript (3-4 x 161 strictly sequential
// cubings: ~4 us on a CPU core, ~80 us on a GPU scalar unit -- measured 5.7 of 6.7 ms at n = 2^24
// with the transcript on the device) and passes the next challenge to the next launch as a kernel
// argument.  Once the tables are down to SC_TAIL elements they are copied out and the last rounds
// run on the host.  Coefficient sums are exact field sums, so any reduction tree is bit-identical
// to the reference's sequential accumulation.
// ============================================================================================
static constexpr size_t SC_TAIL = 1024;      // table size at which the remaining rounds go to the host
static constexpr size_t SC_DOUBLE_MIN = 16 * SC_TAIL;      // tables at least this large take two rounds per round trip

template <int NC>
__device__ __forceinline__ void block_reduce_store(F (&c)[NC], F *partials) {
    __shared__ F red[NC][16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int q = 0; q < NC; q++) { F s = wave_sum(c[q]); if (lane == 0) red[q][wv] = s; }
    __syncthreads();
    if (threadIdx.x < NC) {
        F s = red[threadIdx.x][0];
        for (int w = 1; w < nw; w++) s = fadd(s, red[threadIdx.x][w]);
        stF(partials + (size_t)blockIdx.x * NC + threadIdx.x, s);
    }
}
__device__ __forceinline__ void acc_quad(F (&c)[3], const F &x0, const F &x1, const F &y0, const F &y1) {
    // (dx t + x0)(dy t + y0): a = dx dy, b = dx y0 + x0 dy, c = x0 y0   (src/polynomial.cpp:133-135)
    F dx = fsub(x1, x0), dy = fsub(y1, y0);
    c[0] = fadd(c[0], fmul(dx, dy));
    c[1] = fadd(c[1], fadd(fmul(dx, y0), fmul(x0, dy)));
    c[2] = fadd(c[2], fmul(x0, y0));
}
// round 0 of the 2-product sumcheck: polynomial only (inputs are preserved)
__global__ void __launch_bounds__(256) k_sc2_poly(const F *__restrict__ v1, const F *__restrict__ v2, size_t L, F *__restrict__ partials) {
    F c[3] = {fmake(0), fmake(0), fmake(0)};
    for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < L; j += (size_t)gridDim.x * blockDim.x)
        acc_quad(c, ldF(v1 + 2 * j), ldF(v1 + 2 * j + 1), ldF(v2 + 2 * j), ldF(v2 + 2 * j + 1));
    block_reduce_store<3>(c, partials);
}
// rounds >= 1: fold the previous tables with the challenge just derived (4 -> 2 elements per
// thread and table) and accumulate this round's polynomial from the folded pair in registers.
__global__ void __launch_bounds__(256) k_sc2_fold_poly(const F *__restrict__ s1, const F *__restrict__ s2, F *__restrict__ d1, F *__restrict__ d2,
                                                       size_t L, F r, F *__restrict__ partials) {
    F c[3] = {fmake(0), fmake(0), fmake(0)};
    for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < L; j += (size_t)gridDim.x * blockDim.x) {
        F a0 = ldF(s1 + 4 * j), a1 = ldF(s1 + 4 * j + 1), a2 = ldF(s1 + 4 * j + 2), a3 = ldF(s1 + 4 * j + 3);
        F b0 = ldF(s2 + 4 * j), b1 = ldF(s2 + 4 * j + 1), b2 = ldF(s2 + 4 * j + 2), b3 = ldF(s2 + 4 * j + 3);
        F x0 = fadd(a0, fmul(r, fsub(a1, a0))), x1 = fadd(a2, fmul(r, fsub(a3, a2)));
        F y0 = fadd(b0, fmul(r, fsub(b1, b0))), y1 = fadd(b2, fmul(r, fsub(b3, b2)));
        stF(d1 + 2 * j, x0); stF(d1 + 2 * j + 1, x1); stF(d2 + 2 * j, y0); stF(d2 + 2 * j + 1, y1);
        acc_quad(c, x0, x1, y0, y1);
    }
    block_reduce_store<3>(c, partials);
}
// Two rounds per transcript round trip.  For a quad (a0..a3) of the round-i tables, X(r, t) = x0(r) + t (x1(r) - x0(r)) with
// x0 = a0 + r (a1 - a0), x1 = a2 + r (a3 - a2) is bilinear in (r, t); G(r, t) = sum over quads of X(r, t) Y(r, t) is biquadratic and
// therefore fixed by its nine values on {0, 1, inf}^2 ("inf" = leading coefficient; leading coefficients multiply):
//   X(0,0) = a0, X(1,0) = a1, X(inf,0) = a1 - a0 | X(0,1) = a2, X(1,1) = a3, X(inf,1) = a3 - a2 | X(0,inf) = a2 - a0, X(1,inf) = a3 - a1,
//   X(inf,inf) = (a3 - a2) - (a1 - a0).
// Round i's polynomial is G(r, 0) + G(r, 1); after r = r_i the host has G(r_i, t) at t = 0, 1, inf: round i + 1's polynomial.  Nine
// products per quad instead of 2 x 4 + 4 over the two rounds, one round trip instead of two, and the tables are read once per two
// rounds: with FOLD the kernel first folds sixteen elements of the tables two levels up with (r0, r1) into the quad and stores it.
// Exact field sums: the coefficients are bit-identical to the reference's round-by-round ones (src/sumcheck.cpp:2391-2460).
// One thread per element of the round-i tables: with FOLD it folds the four elements 4g..4g+3 of the tables two levels up (64
// contiguous bytes per lane and table, the access pattern of k_sc2_fold_poly) into element g and stores it; the quad is the four
// adjacent lanes, which exchange values by DPP quad permutes.  Lane q of a quad owns X(q&1, q>>1) = a_q; lanes 1 and 3 also form the
// r = inf points, lanes 2 and 3 the t = inf points, lane 3 the (inf, inf) point.
template <int CTRL> __device__ __forceinline__ F quad_perm(const F &v) {
    F o; uint32_t w[4] = {(uint32_t)v.re, (uint32_t)(v.re >> 32), (uint32_t)v.im, (uint32_t)(v.im >> 32)};
#pragma unroll
    for (int k = 0; k < 4; k++) w[k] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[k], CTRL, 0xF, 0xF, false);
    o.re = (uint64_t)w[0] | ((uint64_t)w[1] << 32); o.im = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
    return o;
}
template <bool FOLD>
__global__ void __launch_bounds__(256) k_sc2_double(const F *__restrict__ s1, const F *__restrict__ s2, F *__restrict__ d1, F *__restrict__ d2, size_t Q, F r0,
                                                    F r1, F *__restrict__ partials) {
    F acc0 = fmake(0), acc1 = fmake(0), acc2 = fmake(0), acc3 = fmake(0);
    const int q = threadIdx.x & 3;
    for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < 4 * Q; g += (size_t)gridDim.x * blockDim.x) {
        F a, b;
        if (FOLD) {
            const F *e = s1 + 4 * g, *h = s2 + 4 * g;
            F e0 = ldF(e), e1 = ldF(e + 1), e2 = ldF(e + 2), e3 = ldF(e + 3);
            F f0 = fadd(e0, fmul(r0, fsub(e1, e0))), f1 = fadd(e2, fmul(r0, fsub(e3, e2)));
            a = fadd(f0, fmul(r1, fsub(f1, f0)));
            F h0 = ldF(h), h1 = ldF(h + 1), h2 = ldF(h + 2), h3 = ldF(h + 3);
            F k0 = fadd(h0, fmul(r0, fsub(h1, h0))), k1 = fadd(h2, fmul(r0, fsub(h3, h2)));
            b = fadd(k0, fmul(r1, fsub(k1, k0)));
            stF(d1 + g, a); stF(d2 + g, b);
        } else { a = ldF(s1 + g); b = ldF(s2 + g); }
        acc0 = fadd(acc0, fmul(a, b));                                                   // (0,0) (1,0) (0,1) (1,1) on lanes 0..3
        const F an = quad_perm<0xA0>(a), bn = quad_perm<0xA0>(b);                        // lanes 1, 3 <- lanes 0, 2
        const F da = fsub(a, an), db = fsub(b, bn);                                      // lanes 1, 3: a1 - a0, a3 - a2
        acc1 = fadd(acc1, fmul(da, db));                                                 // (inf,0) on lane 1, (inf,1) on lane 3
        const F a2 = quad_perm<0x44>(a), b2 = quad_perm<0x44>(b);                        // lanes 2, 3 <- lanes 0, 1
        acc2 = fadd(acc2, fmul(fsub(a, a2), fsub(b, b2)));                               // (0,inf) on lane 2, (1,inf) on lane 3
        const F da2 = quad_perm<0x44>(da), db2 = quad_perm<0x44>(db);                    // lane 3 <- lane 1's differences
        acc3 = fadd(acc3, fmul(fsub(da, da2), fsub(db, db2)));                           // (inf,inf) on lane 3
    }
    const F z = fmake(0);
    F c[9] = {q == 0 ? acc0 : z, q == 1 ? acc0 : z, q == 1 ? acc1 : z, q == 2 ? acc0 : z, q == 3 ? acc0 : z, q == 3 ? acc1 : z,
              q == 2 ? acc2 : z, q == 3 ? acc2 : z, q == 3 ? acc3 : z};
    block_reduce_store<9>(c, partials);
}
// reduce the per-workgroup partials to NC coefficients
template <int NC>
__global__ void __launch_bounds__(256) k_sc_reduce(const F *__restrict__ partials, int nblocks, F *__restrict__ out) {
    F c[NC];
#pragma unroll
    for (int q = 0; q < NC; q++) c[q] = fmake(0);
    for (int b = threadIdx.x; b < nblocks; b += blockDim.x)
#pragma unroll
        for (int q = 0; q < NC; q++) c[q] = fadd(c[q], ldF(partials + (size_t)b * NC + q));
    block_reduce_store<NC>(c, out);
}

// The same reduction, posting to the host mailbox: the NC coefficients go straight into coherent pinned host memory, then the
// launch's sequence number (system-scope release); the host spins on that word instead of queueing a 48-byte copy and sleeping
// in hipStreamSynchronize.  (Folding this into the round kernel itself -- last workgroup to arrive reduces -- needs an agent-scope
// release per workgroup, which on this 8-XCD part writes back each XCD's L2: measured 0.16 ms per launch, so it stays a kernel.)
template <int NC>
__global__ void __launch_bounds__(256) k_sc_reduce_post(const F *__restrict__ partials, int nblocks, Mailbox *mb, uint32_t seq) {
    F c[NC];
#pragma unroll
    for (int q = 0; q < NC; q++) c[q] = fmake(0);
    for (int b = threadIdx.x; b < nblocks; b += blockDim.x)
#pragma unroll
        for (int q = 0; q < NC; q++) c[q] = fadd(c[q], ldF(partials + (size_t)b * NC + q));
    __shared__ F red2[NC][16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int q = 0; q < NC; q++) { F sm = wave_sum(c[q]); if (lane == 0) red2[q][wv] = sm; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 0; q < NC; q++) {
            F sm = red2[q][0];
            for (int w = 1; w < nw; w++) sm = fadd(sm, red2[q][w]);
            uint64_t *o = reinterpret_cast<uint64_t *>(&mb->vals[q]);
            __hip_atomic_store(o, sm.re, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); __hip_atomic_store(o + 1, sm.im, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __hip_atomic_store(&mb->flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// host tail of the 2-product sumcheck: tables a,b of size sz (not yet folded with `rnd` when
// pending_fold), continuing at round `round` exactly as src/sumcheck.cpp:2401-2452
static void sc2_host_tail(std::vector<F> &a, std::vector<F> &b, F &rnd, bool pending_fold, int round, int rounds, MHP h_qpoly, MHP h_r) {
    size_t sz = a.size();
    auto fold = [&](std::vector<F> &v) { for (size_t j = 0; j < sz / 2; j++) v[j] = fadd(v[2 * j], fmul(rnd, fsub(v[2 * j + 1], v[2 * j]))); };
    if (pending_fold) { fold(a); fold(b); sz /= 2; }
    for (int i = round; i < rounds; i++) {
        F pa = fmake(0), pb = fmake(0), pc = fmake(0);
        for (size_t j = 0; j < sz / 2; j++) {
            F dx = fsub(a[2 * j + 1], a[2 * j]), dy = fsub(b[2 * j + 1], b[2 * j]);
            pa = fadd(pa, fmul(dx, dy)); pb = fadd(pb, fadd(fmul(dx, b[2 * j]), fmul(a[2 * j], dy))); pc = fadd(pc, fmul(a[2 * j], b[2 * j]));
        }
        rnd = mimc_hash(rnd, pa); rnd = mimc_hash(rnd, pb); rnd = mimc_hash(rnd, pc);
        h_r[i] = rnd; h_qpoly[3 * i] = pa; h_qpoly[3 * i + 1] = pb; h_qpoly[3 * i + 2] = pc;
        fold(a); fold(b); sz /= 2;
    }
}

// ---- streaming-sumcheck error terms (src/sumcheck.cpp:374-432, 1093-1136): fused multi-output dot products --
// KIND 2: compute2p (b1,b2,f1,f2) -> K1,K2;  KIND 3: compute3p (b1,gate,f1,f2,f3,beta) -> K1..K3;
// KIND 4: compute4p (b1,b2,b3,gate,f1..f4) -> K1..K4;  KIND 13: one batch of batch_prod (b1,b2,b3,f1,f2,f3) -> K1,K2,K3
struct ErrArgs { const F *t[8]; const int32_t *gate; int lk; F lr0, lr1; };
// compute3p_error_terms' selector -> gate map when has_lookups is set (src/sumcheck.cpp:413-427): 0 and 4 -> 1, 2 -> lookup_rand[0], 3 -> lookup_rand[1],
// everything else (1, and the -1 the caller parks addition gates at) -> 0
__device__ __forceinline__ F lk_gate3(int s, const F &lr0, const F &lr1) { return (s == 0 || s == 4) ? fmake(1) : s == 2 ? lr0 : s == 3 ? lr1 : fmake(0); }
template <int KIND, int NC>
__global__ void __launch_bounds__(256) k_err_terms(ErrArgs a, size_t n, F *__restrict__ partials) {
    F K[NC];
#pragma unroll
    for (int q = 0; q < NC; q++) K[q] = fmake(0);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (KIND == 2) {
            const F b1 = ldF(a.t[0] + i), b2 = ldF(a.t[1] + i), f1 = ldF(a.t[2] + i), f2 = ldF(a.t[3] + i);
            K[0] = fadd(K[0], fadd(fmul(b1, f2), fmul(b2, f1)));
            K[1] = fadd(K[1], fmul(b1, b2));
        } else if (KIND == 3) {
            const F b1 = ldF(a.t[0] + i), f1 = ldF(a.t[1] + i), f2 = ldF(a.t[2] + i), f3 = ldF(a.t[3] + i), be = ldF(a.t[4] + i);
            const F gate = a.lk ? lk_gate3(a.gate[i], a.lr0, a.lr1) : fmake((uint64_t)(int64_t)a.gate[i]);
            const F t1 = fadd(fmul(b1, f2), fmul(gate, f1)), t2 = fmul(b1, gate);
            K[0] = fadd(K[0], fadd(fmul(f3, t1), fmul(fmul(be, f1), f2)));
            K[1] = fadd(K[1], fadd(fmul(be, t1), fmul(f3, t2)));
            K[2] = fadd(K[2], fmul(t2, be));
        } else if (KIND == 4) {
            const F b1 = ldF(a.t[0] + i), b2 = ldF(a.t[1] + i), b3 = ldF(a.t[2] + i), f1 = ldF(a.t[3] + i), f2 = ldF(a.t[4] + i), f3 = ldF(a.t[5] + i),
                    f4 = ldF(a.t[6] + i);
            const F gate = a.lk ? fmake(a.gate[i] == 1 ? 1 : 0) : fsub(fmake(1), fmake((uint64_t)(int64_t)a.gate[i]));      // (:388-394)
            const F t1 = fadd(fmul(f1, b2), fmul(f2, b1)), t2 = fadd(fmul(f3, gate), fmul(f4, b3));
            const F t3 = fmul(b1, b2), t4 = fmul(gate, b3), t5 = fmul(f1, f2), t6 = fmul(f3, f4);
            K[0] = fadd(K[0], fadd(fmul(t1, t6), fmul(t2, t5)));
            K[1] = fadd(K[1], fadd(fadd(fmul(t1, t2), fmul(t3, t6)), fmul(t4, t5)));
            K[2] = fadd(K[2], fadd(fmul(t1, t4), fmul(t2, t3)));
            K[3] = fadd(K[3], fmul(t3, t4));
        } else {
            const F b1 = ldF(a.t[0] + i), b2 = ldF(a.t[1] + i), b3 = ldF(a.t[2] + i), f1 = ldF(a.t[3] + i), f2 = ldF(a.t[4] + i), f3 = ldF(a.t[5] + i);
            const F t1 = fadd(fmul(b1, f2), fmul(b2, f1)), t2 = fmul(b1, b2);
            K[0] = fadd(K[0], fadd(fmul(f3, t1), fmul(fmul(b3, f1), f2)));
            K[1] = fadd(K[1], fadd(fmul(b3, t1), fmul(f3, t2)));
            K[2] = fadd(K[2], fmul(t2, b3));
        }
    }
    block_reduce_store<NC>(K, partials);
}
// runs one error-term reduction; h_K receives the NC sums (not accumulated)
template <int KIND, int NC>
static int run_err(hobbit_ctx *ctx, const char *name, const ErrArgs &a, size_t n, MHP h_K) {
    const int MAXB = 1024;
    F *ws; HB_TRY(ctx->workspace(((size_t)MAXB * NC + NC + 4) * sizeof(F), (void **)&ws));
    F *part = ws, *coef = ws + (size_t)MAXB * NC;
    F *pin; HB_TRY(ctx->pinned(NC * sizeof(F), (void **)&pin));
    int nb = grid_for(n, 256, MAXB);
    HB_LAUNCH(ctx, name, (k_err_terms<KIND, NC>), dim3(nb), dim3(256), 0, a, n, part);
    HB_LAUNCH(ctx, "k_sc_reduce", k_sc_reduce<NC>, dim3(1), dim3(256), 0, part, nb, coef);
    HB_CHECK(ctx, hipMemcpyAsync(pin, coef, NC * sizeof(F), hipMemcpyDeviceToHost, ctx->stream));
    HB_TRY(ctx->sync());
    for (int q = 0; q < NC; q++) h_K[q] = pin[q];
    return 0;
}
int launch_err_terms(hobbit_ctx *ctx, int kind, const F *const *tables, const int32_t *gate, size_t n, MHP h_K) {
    ErrArgs a; for (int i = 0; i < 8; i++) a.t[i] = tables[i]; a.gate = gate; a.lk = ctx->has_lookups ? 1 : 0; a.lr0 = ctx->lookup_rand[0]; a.lr1 = ctx->lookup_rand[1];
    if (!n) { int nc = kind == 2 ? 2 : kind == 4 ? 4 : 3; for (int q = 0; q < nc; q++) h_K[q] = fmake(0); return 0; }
    switch (kind) {
        case 2: return run_err<2, 2>(ctx, "k_err2p", a, n, h_K);
        case 3: return run_err<3, 3>(ctx, "k_err3p", a, n, h_K);
        case 4: return run_err<4, 4>(ctx, "k_err4p", a, n, h_K);
        case 13: return run_err<13, 3>(ctx, "k_batch_prod_terms", a, n, h_K);
    }
    return ctx->fail(HOBBIT_EINVAL, "err_terms: unknown kind");
}
// fold[j] += rand * (F)sel[j]  or  rand * (1 - sel[j])   (src/sumcheck.cpp:863,867: gate selector folds)
__global__ void k_axpy_i32(F *__restrict__ y, const int32_t *__restrict__ sel, F a, int one_minus, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        F v = fmake((uint64_t)(int64_t)sel[i]);
        if (one_minus) v = fsub(fmake(1), v);
        stF(y + i, fadd(ldF(y + i), fmul(a, v)));
    }
}
int launch_axpy_i32(hobbit_ctx *ctx, F *y, const int32_t *sel, F a, int one_minus, size_t n) {
    if (!n) return 0;
    HB_LAUNCH(ctx, "k_axpy_i32", k_axpy_i32, dim3(grid_for(n, 256)), dim3(256), 0, y, sel, a, one_minus, n);
    return 0;
}

// memory / lookup fingerprints of the wiring-consistency streams (src/witness_stream.cpp:2196, 2290-2305; src/main.cpp:1039-1044):
// out[i] = addr[i] + 1 + a * value[i] (+ b * freq[i])
__global__ void k_fingerprint(const F *__restrict__ addr, const F *__restrict__ value, const F *__restrict__ freq, F a, F b, F *__restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        F v = fadd(fadd(ldF(addr + i), fmake(1)), fmul(a, ldF(value + i)));
        if (freq) v = fadd(v, fmul(b, ldF(freq + i)));
        stF(out + i, v);
    }
}
int launch_fingerprint(hobbit_ctx *ctx, const F *addr, const F *value, const F *freq, F a, F b, F *out, size_t n) {
    if (!n) return 0;
    HB_LAUNCH(ctx, "k_fingerprint", k_fingerprint, dim3(grid_for(n, 256)), dim3(256), 0, addr, value, freq, a, b, out, n);
    return 0;
}

// ---- prove_gate_consistency_lookups (src/sumcheck.cpp:503-795) device pieces ---------------------------------------
// One chunk's selector rewrites (:568-585) and lookup output column: s2 = selectors for the R call (2 -> 3), s3 = for the lookup call
// (0 -> -1, 2 -> 4), blo = lookup_rand[0] L + lookup_rand[1] R - O on lookup rows, 0 elsewhere.
__global__ void k_lkp_prepare(const int32_t *__restrict__ S, const F *__restrict__ L, const F *__restrict__ R, const F *__restrict__ O, F lr0, F lr1,
                              int32_t *__restrict__ s2, int32_t *__restrict__ s3, F *__restrict__ blo, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int s = S[i];
        const bool lk = s != 0 && s != 1;
        if (s2) s2[i] = s == 2 ? 3 : s;
        if (s3) s3[i] = s == 0 ? -1 : s == 2 ? 4 : s;
        stF(blo + i, lk ? fsub(fadd(fmul(lr0, ldF(L + i)), fmul(lr1, ldF(R + i))), ldF(O + i)) : fmake(0));
    }
}
// the four selector-derived folds (:510-536 with rnd = 1 on zeroed tables, :614-625): add_L += rnd {1, 0, lr0}[s], add_R += rnd {1, 0, lr1}[s],
// lkp += rnd [s is a lookup], mul += rnd [s == 1]
__global__ void k_lkp_sel_fold(const int32_t *__restrict__ S, F rnd, F lr0, F lr1, F *__restrict__ aL, F *__restrict__ aR, F *__restrict__ lkp, F *__restrict__ mul, size_t n) {
    const F rl0 = fmul(rnd, lr0), rl1 = fmul(rnd, lr1);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int s = S[i];
        if (s == 1) stF(mul + i, fadd(ldF(mul + i), rnd));
        else if (s == 0) { stF(aL + i, fadd(ldF(aL + i), rnd)); stF(aR + i, fadd(ldF(aR + i), rnd)); }
        else { stF(lkp + i, fadd(ldF(lkp + i), rnd)); stF(aL + i, fadd(ldF(aL + i), rl0)); stF(aR + i, fadd(ldF(aR + i), rl1)); }
    }
}
int launch_lkp_prepare(hobbit_ctx *ctx, const int32_t *S, const F *L, const F *R, const F *O, int32_t *s2, int32_t *s3, F *blo, size_t n) {
    if (!n) return 0;
    HB_LAUNCH(ctx, "k_lkp_prepare", k_lkp_prepare, dim3(grid_for(n, 256)), dim3(256), 0, S, L, R, O, ctx->lookup_rand[0], ctx->lookup_rand[1], s2, s3, blo, n);
    return 0;
}
int launch_lkp_sel_fold(hobbit_ctx *ctx, const int32_t *S, F rnd, F *aL, F *aR, F *lkp, F *mul, size_t n) {
    if (!n) return 0;
    HB_LAUNCH(ctx, "k_lkp_sel_fold", k_lkp_sel_fold, dim3(grid_for(n, 256)), dim3(256), 0, S, rnd, ctx->lookup_rand[0], ctx->lookup_rand[1], aL, aR, lkp, mul, n);
    return 0;
}

// ---- _whir_prove (src/Virgo.cpp:519-686) device pieces -----------------------------------------------
// one fold round over the half split (j, j+L): quadratic coefficients of sum (dp t + p)(db t + b) and, in the same pass,
// poly[j] += a (poly[j+L] - poly[j]), beta[j] += a (beta[j+L] - beta[j])  (the challenge a = random() does not depend on
// the polynomial, so it is drawn before the pass)
__global__ void __launch_bounds__(256) k_whir_round(F *__restrict__ poly, F *__restrict__ beta, size_t L, F a, F *__restrict__ partials) {
    F c[3] = {fmake(0), fmake(0), fmake(0)};
    for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < L; j += (size_t)gridDim.x * blockDim.x) {
        const F p0 = ldF(poly + j), p1 = ldF(poly + j + L), b0 = ldF(beta + j), b1 = ldF(beta + j + L);
        const F d1 = fsub(p1, p0), d2 = fsub(b1, b0);
        c[0] = fadd(c[0], fmul(d1, d2)); c[1] = fadd(c[1], fadd(fmul(d1, b0), fmul(p0, d2))); c[2] = fadd(c[2], fmul(p0, b0));
        stF(poly + j, fadd(p0, fmul(a, d1))); stF(beta + j, fadd(b0, fmul(a, d2)));
    }
    block_reduce_store<3>(c, partials);
}
// <a, b> over n elements: per-workgroup partials (part: >= 1024 F), then one workgroup
__global__ void __launch_bounds__(256) k_dot(const F *__restrict__ a, const F *__restrict__ b, size_t n, F *__restrict__ partials) {
    F c[1] = {fmake(0)};
    for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x) c[0] = fadd(c[0], fmul(ldF(a + j), ldF(b + j)));
    block_reduce_store<1>(c, partials);
}
int launch_dot(hobbit_ctx *ctx, const F *a, const F *b, size_t n, F *part, F *out) {
    int nb = grid_for(n, 256, 1024);
    HB_LAUNCH(ctx, "k_dot", k_dot, dim3(nb), dim3(256), 0, a, b, n, part);
    HB_LAUNCH(ctx, "k_sc_reduce", k_sc_reduce<1>, dim3(1), dim3(256), 0, part, nb, out);
    return 0;
}
int launch_whir_round(hobbit_ctx *ctx, F *poly, F *beta, size_t L, F a, F *part, F *coef) {   // coef: 3 F of this round (device)
    int nb = grid_for(L, 256, 1024);
    HB_LAUNCH(ctx, "k_whir_round", k_whir_round, dim3(nb), dim3(256), 0, poly, beta, L, a, part);
    HB_LAUNCH(ctx, "k_sc_reduce", k_sc_reduce<3>, dim3(1), dim3(256), 0, part, nb, coef);
    return 0;
}
// batched precompute_beta: reps tables of 2^v entries side by side (row stride ld); one doubling level per launch;
// z[p*v + (v-1-level)] is the challenge of table p at this level
__global__ void k_eq_step_batched(const F *__restrict__ old, F *__restrict__ nw, size_t m, size_t ld, const F *__restrict__ z, int v, int level) {
    const F *o = old + (size_t)blockIdx.y * ld; F *n = nw + (size_t)blockIdx.y * ld;
    const F r = ldF(z + (size_t)blockIdx.y * v + (v - 1 - level));
    for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < m; j += (size_t)gridDim.x * blockDim.x) {
        const F x = ldF(o + j), t = fmul(r, x);
        stF(n + 2 * j, fsub(x, t)); stF(n + 2 * j + 1, t);
    }
}
// the first h <= 11 levels of every table in one launch (one workgroup per table, table in LDS): a level per launch is latency only up there
__global__ void __launch_bounds__(256) k_eq_head_batched(F *__restrict__ out, size_t ld, const F *__restrict__ z, int v, int h) {
    __shared__ F s[2048];
    const F *zz = z + (size_t)blockIdx.x * v;
    if (threadIdx.x == 0) s[0] = fmake(1);
    __syncthreads();
    for (int i = 0; i < h; i++) {
        const uint32_t m = 1u << i;
        const F r = ldF(zz + (v - 1 - i));
        F o[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const uint32_t j = threadIdx.x + 256 * u; if (j < m) o[u] = ldF(&s[j]); }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t j = threadIdx.x + 256 * u;
            if (j < m) { const F t = fmul(r, o[u]); stF(&s[2 * j], fsub(o[u], t)); stF(&s[2 * j + 1], t); }
        }
        __syncthreads();
    }
    F *n = out + (size_t)blockIdx.x * ld;
    for (uint32_t j = threadIdx.x; j < (1u << h); j += 256) stF(n + j, ldF(&s[j]));
}
int launch_eq_head_batched(hobbit_ctx *ctx, F *out, size_t ld, const F *z, int v, int h, int reps) {
    if (h < 0 || h > 11 || h > v) return ctx->fail(HOBBIT_EINVAL, "eq_head_batched: 0 <= h <= min(v, 11)");
    HB_LAUNCH(ctx, "k_eq_step_batched", k_eq_head_batched, dim3(reps), dim3(256), 0, out, ld, z, v, h);
    return 0;
}
int launch_eq_step_batched(hobbit_ctx *ctx, const F *old, F *nw, size_t m, size_t ld, const F *z, int v, int level, int reps) {
    HB_LAUNCH(ctx, "k_eq_step_batched", k_eq_step_batched, dim3(grid_for(m, 256, 256), reps), dim3(256), 0, old, nw, m, ld, z, v, level);
    return 0;
}
__global__ void k_fill_F(F *__restrict__ p, size_t stride, size_t n, F v) {
    size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (g < n) stF(p + g * stride, v);
}
int launch_fill_F(hobbit_ctx *ctx, F *p, size_t stride, size_t n, F v) {
    if (!n) return 0;
    HB_LAUNCH(ctx, "k_fill_F", k_fill_F, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, p, stride, n, v);
    return 0;
}

// ---- 2-product sumcheck whose SECOND table is sparse (the open's P3: 5900 non-zeros of 2^25, src/PC_utils.cpp:331-339) -------------------
// The round polynomials only see the quads in which the sparse table is non-zero, so the large levels need no dense pass over it and no
// products over the dense table: per round trip one plain 4 -> 1 fold of the dense table (k_sc2_fold4) and ONE workgroup that folds the
// sparse list two levels (sorted (index, value) pairs in, one pair per surviving quad out) and evaluates G(r, t) on the surviving quads
// (k_sc2_sparse_round, which also posts to the mailbox: no separate reduction launch).  Same field sums as the dense kernels, so the
// transcript is bit-identical.  Below SC_DOUBLE_MIN the list is scattered into a dense table and the dense path takes over.
__global__ void __launch_bounds__(256) k_sc2_fold4(const F *__restrict__ s, F *__restrict__ d, size_t nout, F r0, F r1) {
    for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < nout; g += (size_t)gridDim.x * blockDim.x) {
        const F *e = s + 4 * g;
        const F e0 = ldF(e), e1 = ldF(e + 1), e2 = ldF(e + 2), e3 = ldF(e + 3);
        const F f0 = fadd(e0, fmul(r0, fsub(e1, e0))), f1 = fadd(e2, fmul(r0, fsub(e3, e2)));
        stF(d + g, fadd(f0, fmul(r1, fsub(f1, f0))));
    }
}
// gather the (up to four, consecutive in the sorted list) entries of the quad that entry e leads
__device__ __forceinline__ void sparse_quad(const uint64_t *__restrict__ idx, const F *__restrict__ val, uint32_t e, uint32_t m, uint64_t q, F (&b)[4]) {
    b[0] = b[1] = b[2] = b[3] = fmake(0);
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        if (e + k < m) {
            const uint64_t ix = idx[e + k];
            if ((ix >> 2) == q) { const F v = ldF(val + e + k); const uint32_t p = (uint32_t)(ix & 3); if (p == 0) b[0] = v; else if (p == 1) b[1] = v; else if (p == 2) b[2] = v; else b[3] = v; }
        }
    }
}
template <bool FOLD>
__global__ void __launch_bounds__(1024) k_sc2_sparse_round(const uint64_t *__restrict__ idx_in, const F *__restrict__ val_in, const uint32_t *__restrict__ m_in, uint32_t m_arg,
                                                           uint64_t *__restrict__ idx_out, F *__restrict__ val_out, uint32_t *__restrict__ m_out,
                                                           const F *__restrict__ dense, F r0, F r1, Mailbox *mb, uint32_t seq) {
    __shared__ uint32_t scan[1024];
    __shared__ F red2[9][16];
    const uint32_t t = threadIdx.x;
    uint32_t m = m_in ? *m_in : m_arg;             // (the caller's own list: its length is a launch argument)
    const uint64_t *idx = idx_in; const F *val = val_in;
    if (FOLD) {                                   // two levels: one output pair per quad that holds a non-zero
        const uint32_t per = (m + 1023) / 1024, lo = min(m, t * per), hi = min(m, lo + per);
        uint32_t c = 0;
        for (uint32_t e = lo; e < hi; e++) c += (e == 0 || (idx_in[e] >> 2) != (idx_in[e - 1] >> 2)) ? 1u : 0u;
        scan[t] = c;
        __syncthreads();
        for (uint32_t d = 1; d < 1024; d <<= 1) {   // inclusive Hillis-Steele scan
            const uint32_t v = t >= d ? scan[t - d] : 0;
            __syncthreads();
            scan[t] += v;
            __syncthreads();
        }
        uint32_t base = scan[t] - c;
        const uint32_t total = scan[1023];
        for (uint32_t e = lo; e < hi; e++) {
            const uint64_t q = idx_in[e] >> 2;
            if (e == 0 || (idx_in[e - 1] >> 2) != q) {
                F b[4]; sparse_quad(idx_in, val_in, e, m, q, b);
                const F k0 = fadd(b[0], fmul(r0, fsub(b[1], b[0]))), k1 = fadd(b[2], fmul(r0, fsub(b[3], b[2])));
                idx_out[base] = q; stF(val_out + base, fadd(k0, fmul(r1, fsub(k1, k0)))); base++;
            }
        }
        if (t == 0) *m_out = total;
        __threadfence_block();
        __syncthreads();
        m = total; idx = idx_out; val = val_out;
    }
    // G(r, t) over the quads of this level that hold a non-zero: the nine values of k_sc2_double
    F c[9];
#pragma unroll
    for (int q = 0; q < 9; q++) c[q] = fmake(0);
    {
        const uint32_t per = (m + 1023) / 1024, lo = min(m, t * per), hi = min(m, lo + per);
        for (uint32_t e = lo; e < hi; e++) {
            const uint64_t Q = idx[e] >> 2;
            if (e == 0 || (idx[e - 1] >> 2) != Q) {
                F b[4]; sparse_quad(idx, val, e, m, Q, b);
                const F *ap = dense + 4 * Q;
                const F a0 = ldF(ap), a1 = ldF(ap + 1), a2 = ldF(ap + 2), a3 = ldF(ap + 3);
                c[0] = fadd(c[0], fmul(a0, b[0])); c[1] = fadd(c[1], fmul(a1, b[1])); c[3] = fadd(c[3], fmul(a2, b[2])); c[4] = fadd(c[4], fmul(a3, b[3]));
                const F da01 = fsub(a1, a0), da23 = fsub(a3, a2), db01 = fsub(b[1], b[0]), db23 = fsub(b[3], b[2]);
                c[2] = fadd(c[2], fmul(da01, db01)); c[5] = fadd(c[5], fmul(da23, db23));
                c[6] = fadd(c[6], fmul(fsub(a2, a0), fsub(b[2], b[0]))); c[7] = fadd(c[7], fmul(fsub(a3, a1), fsub(b[3], b[1])));
                c[8] = fadd(c[8], fmul(fsub(da23, da01), fsub(db23, db01)));
            }
        }
    }
    const int lane = t & 63, wv = t >> 6;
#pragma unroll
    for (int q = 0; q < 9; q++) { F sm = wave_sum(c[q]); if (lane == 0) red2[q][wv] = sm; }
    __syncthreads();
    if (t == 0) {
        for (int q = 0; q < 9; q++) {
            F sm = red2[q][0];
            for (int w = 1; w < 16; w++) sm = fadd(sm, red2[q][w]);
            uint64_t *o = reinterpret_cast<uint64_t *>(&mb->vals[q]);
            __hip_atomic_store(o, sm.re, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); __hip_atomic_store(o + 1, sm.im, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __hip_atomic_store(&mb->flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ void __launch_bounds__(256) k_scatter_counted(const uint64_t *__restrict__ idx, const F *__restrict__ val, const uint32_t *__restrict__ m, F *__restrict__ out) {
    const uint32_t n = *m;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) stF(out + idx[i], ldF(val + i));
}

static int sumcheck2_impl(hobbit_ctx *ctx, const F *v1, const F *v2, const uint64_t *sp_idx, const F *sp_val, size_t sp_m, size_t n, F prev_r, MHP h_qpoly, MHP h_r,
                          MHP h_vr, MHP h_final);
int launch_sumcheck2(hobbit_ctx *ctx, const F *v1, const F *v2, size_t n, F prev_r, MHP h_qpoly, MHP h_r, MHP h_vr, MHP h_final) {
    return sumcheck2_impl(ctx, v1, v2, nullptr, nullptr, 0, n, prev_r, h_qpoly, h_r, h_vr, h_final);
}
// v2 given as sp_m sorted, distinct (index, value) pairs on the device (every other entry zero)
int launch_sumcheck2_sparse(hobbit_ctx *ctx, const F *v1, const uint64_t *d_idx, const F *d_val, size_t m, size_t n, F prev_r, MHP h_qpoly, MHP h_r, MHP h_vr, MHP h_final) {
    if (!m || m > ((size_t)1 << 20)) return ctx->fail(HOBBIT_EINVAL, "sumcheck2_sparse: between 1 and 2^20 non-zeros");
    return sumcheck2_impl(ctx, v1, nullptr, d_idx, d_val, m, n, prev_r, h_qpoly, h_r, h_vr, h_final);
}
static int sumcheck2_impl(hobbit_ctx *ctx, const F *v1, const F *v2, const uint64_t *sp_idx, const F *sp_val, size_t sp_m, size_t n, F prev_r, MHP h_qpoly, MHP h_r,
                          MHP h_vr, MHP h_final) {
    int rounds = 0; while (((size_t)1 << rounds) < n) rounds++;
    if (((size_t)1 << rounds) != n || n < 2) return ctx->fail(HOBBIT_EINVAL, "sumcheck2: n must be a power of two >= 2");
    const int MAXB = 1024;
    F rnd = prev_r;
    std::vector<F> ta, tb;
    if (n <= SC_TAIL) {                       // small instance: all rounds on the host
        if (sp_idx) {
            F *dv2; HB_TRY(ctx->workspace(n * sizeof(F), (void **)&dv2));
            HB_TRY(launch_zero(ctx, dv2, n * sizeof(F)));
            HB_TRY(launch_scatter(ctx, sp_idx, sp_val, sp_m, dv2));
            v2 = dv2;
        }
        ta.resize(n); tb.resize(n);
        HB_CHECK(ctx, hipMemcpyAsync(ta.data(), v1, n * sizeof(F), hipMemcpyDeviceToHost, ctx->stream));
        HB_CHECK(ctx, hipMemcpyAsync(tb.data(), v2, n * sizeof(F), hipMemcpyDeviceToHost, ctx->stream));
        HB_TRY(ctx->sync());
        sc2_host_tail(ta, tb, rnd, false, 0, rounds, h_qpoly, h_r);
    } else {
        size_t szA = n / 2, szB = n / 4;
        const bool sparse = sp_idx && n >= 16 * SC_DOUBLE_MIN;            // (shorter tables: scatter at once, dense path)
        const size_t sp_elems = sp_idx ? 2 * (sp_m + (sp_m + 1) / 2) + 8 + (sparse ? 0 : n) : 0;      // two (value | index) lists + counters (+ a dense copy)
        F *ws; HB_TRY(ctx->workspace((2 * szA + 2 * szB + (size_t)MAXB * 9 + 4 + sp_elems) * sizeof(F), (void **)&ws));
        F *A1 = ws, *A2 = A1 + szA, *B1 = A2 + szA, *B2 = B1 + szB, *part = B2 + szB;
        F *spw = part + (size_t)MAXB * 9 + 4;
        F *lval[2] = {spw, spw + sp_m + (sp_m + 1) / 2}; uint64_t *lidx[2] = {reinterpret_cast<uint64_t *>(lval[0] + sp_m), reinterpret_cast<uint64_t *>(lval[1] + sp_m)};
        uint32_t *lcnt = reinterpret_cast<uint32_t *>(spw + 2 * (sp_m + (sp_m + 1) / 2));       // [0]: the caller's count, [1], [2]: the two lists'
        if (sp_idx && !sparse) {
            F *dv2 = spw + 2 * (sp_m + (sp_m + 1) / 2) + 8;
            HB_TRY(launch_zero(ctx, dv2, n * sizeof(F)));
            HB_TRY(launch_scatter(ctx, sp_idx, sp_val, sp_m, dv2));
            v2 = dv2;
        }
        Mailbox *mb; unsigned *ticket; HB_TRY(ctx->mailbox(&mb, &ticket));
        const F *s1 = v1, *s2 = v2;
        F *d1 = A1, *d2 = A2;
        int i = 0;
        size_t cur = n;                       // size of the tables s1/s2 point at
        // ---- large tables: two rounds per round trip (k_sc2_double) ----
        if (n >= SC_DOUBLE_MIN) {
            const F *S1 = v1, *S2 = v2; size_t S_size = n;            // the materialised tables: level i - 2 (or the inputs)
            F *D1 = B1, *D2 = B2;                                     // T_2 has n/4 elements: fits B; later levels fit either
            bool pend = false; F pr0 = fmake(0), pr1 = fmake(0);      // (r_{i-2}, r_{i-1}): still to be applied to S
            const uint64_t *cidx = sp_idx; const F *cval = sp_val; const uint32_t *ccnt = nullptr; int lnext = 0;  // the sparse list at the level S1 is at
            while ((n >> i) >= SC_DOUBLE_MIN) {
                const size_t Q = (n >> i) / 4;
                const int nb = grid_for(4 * Q, 256, MAXB);
                const uint32_t seq = ++ctx->mbox_seq;
                if (sparse) {
                    if (!pend) HB_LAUNCH(ctx, "k_sc2_sparse_round", k_sc2_sparse_round<false>, dim3(1), dim3(1024), 0, cidx, cval, ccnt, (uint32_t)sp_m, (uint64_t *)nullptr,
                                         (F *)nullptr, (uint32_t *)nullptr, S1, pr0, pr1, mb, seq);
                    else {
                        HB_LAUNCH(ctx, "k_sc2_fold4", k_sc2_fold4, dim3(grid_for(4 * Q, 256, 4096)), dim3(256), 0, S1, D1, 4 * Q, pr0, pr1);      // the dense table: level i - 2 -> level i
                        HB_LAUNCH(ctx, "k_sc2_sparse_round", k_sc2_sparse_round<true>, dim3(1), dim3(1024), 0, cidx, cval, ccnt, (uint32_t)sp_m, lidx[lnext], lval[lnext],
                                  lcnt + 1 + lnext, (const F *)D1, pr0, pr1, mb, seq);
                        cidx = lidx[lnext]; cval = lval[lnext]; ccnt = lcnt + 1 + lnext; lnext ^= 1;
                        S1 = D1; S_size = 4 * Q;
                        if (D1 == B1) { D1 = A1; D2 = A2; } else { D1 = B1; D2 = B2; }
                    }
                } else {
                if (!pend) HB_LAUNCH(ctx, "k_sc2_double", k_sc2_double<false>, dim3(nb), dim3(256), 0, S1, S2, (F *)nullptr, (F *)nullptr, Q, pr0, pr1, part);
                else {
                    HB_LAUNCH(ctx, "k_sc2_double", k_sc2_double<true>, dim3(nb), dim3(256), 0, S1, S2, D1, D2, Q, pr0, pr1, part);
                    S1 = D1; S2 = D2; S_size = 4 * Q;
                    if (D1 == B1) { D1 = A1; D2 = A2; } else { D1 = B1; D2 = B2; }
                }
                HB_LAUNCH(ctx, "k_sc_reduce", k_sc_reduce_post<9>, dim3(1), dim3(256), 0, part, nb, mb, seq);
                }
                HB_TRY(ctx->mbox_wait(seq));
                F G[9]; for (int q = 0; q < 9; q++) G[q] = mb->vals[q];            // G(r, t) at (0,0) (1,0) (inf,0) | (0,1) (1,1) (inf,1) | (0,inf) (1,inf) (inf,inf)
                // round i: G(r, 0) + G(r, 1) as (a, b, c) from its values at r = inf, 1, 0
                F p0[3]; p0[0] = fadd(G[2], G[5]); p0[2] = fadd(G[0], G[3]); p0[1] = fsub(fsub(fadd(G[1], G[4]), p0[0]), p0[2]);
                for (int q = 0; q < 3; q++) { rnd = mimc_hash(rnd, p0[q]); h_qpoly[3 * i + q] = p0[q]; }
                h_r[i] = rnd;
                const F r = rnd, rr = fmul(r, r);
                F v[3];                                                             // G(r_i, t) at t = 0, 1, inf
                for (int k = 0; k < 3; k++) { const F cc = G[3 * k], aa = G[3 * k + 2], bb = fsub(fsub(G[3 * k + 1], aa), cc); v[k] = fadd(cc, fadd(fmul(r, bb), fmul(rr, aa))); }
                const F p1[3] = {v[2], fsub(fsub(v[1], v[2]), v[0]), v[0]};         // round i + 1: (a, b, c)
                for (int q = 0; q < 3; q++) { rnd = mimc_hash(rnd, p1[q]); h_qpoly[3 * (i + 1) + q] = p1[q]; }
                h_r[i + 1] = rnd;
                pend = true; pr0 = r; pr1 = rnd; i += 2;
            }
            // bridge to the round-by-round loop: it expects s = T_{i-1} and rnd = r_{i-1}.  S = T_{i-2}: fold it once with r_{i-2}
            // (the polynomial this launch also sums is round i-1's, already known: its partials are not reduced).
            if (sparse) {                                             // the second table, dense from here on: level i - 2, beside S1 (n >= 16 SC_DOUBLE_MIN: S1 is A1 or B1)
                F *m2 = (S1 == A1) ? A2 : B2;
                HB_TRY(launch_zero(ctx, m2, S_size * sizeof(F)));
                HB_LAUNCH(ctx, "k_scatter_counted", k_scatter_counted, dim3(32), dim3(256), 0, cidx, cval, ccnt, m2);
                S2 = m2;
            }
            F *b1 = (S1 == A1) ? B1 : A1, *b2 = (S1 == A1) ? B2 : A2;
            const size_t L = S_size / 4;
            HB_LAUNCH(ctx, "k_sc2_fold_poly", k_sc2_fold_poly, dim3(grid_for(L, 256, MAXB)), dim3(256), 0, S1, S2, b1, b2, L, pr0, part);
            s1 = b1; s2 = b2; cur = 2 * L;
            if (b1 == A1) { d1 = B1; d2 = B2; } else { d1 = A1; d2 = A2; }
        }
        for (;; i++) {
            size_t L = n >> (i + 1);          // pairs of the round-i tables
            int nb = grid_for(L, 256, MAXB);
            const uint32_t seq = ++ctx->mbox_seq;
            if (i == 0) HB_LAUNCH(ctx, "k_sc2_poly", k_sc2_poly, dim3(nb), dim3(256), 0, s1, s2, L, part);
            else {
                HB_LAUNCH(ctx, "k_sc2_fold_poly", k_sc2_fold_poly, dim3(nb), dim3(256), 0, s1, s2, d1, d2, L, rnd, part);
                s1 = d1; s2 = d2; cur = 2 * L;
                if (d1 == A1) { d1 = B1; d2 = B2; } else { d1 = A1; d2 = A2; }
            }
            HB_LAUNCH(ctx, "k_sc_reduce", k_sc_reduce_post<3>, dim3(1), dim3(256), 0, part, nb, mb, seq);
            HB_TRY(ctx->mbox_wait(seq));                                  // the last workgroup posts the three coefficients to the host
            for (int q = 0; q < 3; q++) { const F cq = mb->vals[q]; rnd = mimc_hash(rnd, cq); h_qpoly[3 * i + q] = cq; }
            h_r[i] = rnd;
            if (cur <= 2 * SC_TAIL || i == rounds - 1) break;   // hand the (unfolded) round-i tables to the host
        }
        ta.resize(cur); tb.resize(cur);
        HB_CHECK(ctx, hipMemcpyAsync(ta.data(), s1, cur * sizeof(F), hipMemcpyDeviceToHost, ctx->stream));
        HB_CHECK(ctx, hipMemcpyAsync(tb.data(), s2, cur * sizeof(F), hipMemcpyDeviceToHost, ctx->stream));
        HB_TRY(ctx->sync());
        sc2_host_tail(ta, tb, rnd, true, i + 1, rounds, h_qpoly, h_r);
    }
    rnd = mimc_hash(rnd, ta[0]); rnd = mimc_hash(rnd, tb[0]);          // src/sumcheck.cpp:2444-2446
    h_vr[0] = ta[0]; h_vr[1] = tb[0]; *h_final = rnd;
    return 0;
}

// ---- degree-4 gate-consistency sumcheck (src/sumcheck.cpp:875-929) --------------------------------------
// Tables t[0..5] = fold_add, fold_beta, fold_L, fold_R, fold_O, fold_mul.  Per pair, twelve coefficient sums:
// c[0..3]  cubic   add * beta * (a0 L + a1 R)      (l1*l2*l3, src/polynomial.cpp:91-93,133-135)
// c[4..8]  quartic mul * beta * L * R              (cubic * linear, :99-101)
// c[9..11] quadratic beta * O
// The host combines them with a2, a3 (:899-903), hashes (mimc_hash(coefficient, rand): coefficient is the input) and folds.
template <int NT> struct GateTabsT { const F *s[NT]; F *d[NT]; };
HB_HD void cubic_acc(F *c, const F &b0, const F &d0, const F &b1, const F &d1, const F &b2, const F &d2) {     // (l1*l2)*l3
    F qa = fmul(d0, d1), qb = fadd(fmul(d0, b1), fmul(b0, d1)), qc = fmul(b0, b1);
    c[0] = fadd(c[0], fmul(qa, d2));
    c[1] = fadd(c[1], fadd(fmul(qa, b2), fmul(qb, d2)));
    c[2] = fadd(c[2], fadd(fmul(qb, b2), fmul(qc, d2)));
    c[3] = fadd(c[3], fmul(qc, b2));
}
HB_HD void quartic_acc(F *c, const F &bm, const F &dm, const F &bb, const F &db, const F &bl, const F &dl, const F &br, const F &dr) {   // ((mul*beta)*L)*R
    F ma = fmul(dm, db), mb = fadd(fmul(dm, bb), fmul(bm, db)), mc = fmul(bm, bb);
    F ka = fmul(ma, dl), kb = fadd(fmul(ma, bl), fmul(mb, dl)), kc = fadd(fmul(mb, bl), fmul(mc, dl)), kd = fmul(mc, bl);
    c[0] = fadd(c[0], fmul(ka, dr));
    c[1] = fadd(c[1], fadd(fmul(ka, br), fmul(kb, dr)));
    c[2] = fadd(c[2], fadd(fmul(kb, br), fmul(kc, dr)));
    c[3] = fadd(c[3], fadd(fmul(kc, br), fmul(kd, dr)));
    c[4] = fadd(c[4], fmul(kd, br));
}
HB_HD void quad_acc(F *c, const F &b0, const F &d0, const F &b1, const F &d1) {
    c[0] = fadd(c[0], fmul(d0, d1));
    c[1] = fadd(c[1], fadd(fmul(d0, b1), fmul(b0, d1)));
    c[2] = fadd(c[2], fmul(b0, b1));
}
// prove_gate_consistency / _standard: tables add, beta, L, R, O, mul; twelve sums
struct GateStd {
    static constexpr int NT = 6, NS = 12, NA = 4;
    static HB_HD void acc(F (&c)[12], const F (&b)[6], const F (&e)[6], const F &a0, const F &a1) {
        F d[6];
#pragma unroll
        for (int q = 0; q < 6; q++) d[q] = fsub(e[q], b[q]);
        const F l3a = fadd(fmul(a0, d[2]), fmul(a1, d[3])), l3b = fadd(fmul(a0, b[2]), fmul(a1, b[3]));
        cubic_acc(c, b[0], d[0], b[1], d[1], l3b, l3a);
        quartic_acc(c + 4, b[5], d[5], b[1], d[1], b[2], d[2], b[3], d[3]);
        quad_acc(c + 9, b[1], d[1], b[4], d[4]);
    }
    // combine the twelve sums into the quartic (a..e)  (:899-903)
    static void combine(const F *c, CHP a, F *p) {
        p[0] = fmul(a[2], c[4]);
        p[1] = fadd(fmul(a[2], c[5]), c[0]);
        p[2] = fadd(fadd(fmul(a[2], c[6]), c[1]), fmul(a[3], c[9]));
        p[3] = fadd(fadd(fmul(a[2], c[7]), c[2]), fmul(a[3], c[10]));
        p[4] = fadd(fadd(fmul(a[2], c[8]), c[3]), fmul(a[3], c[11]));
    }
};
// prove_gate_consistency_lookups (src/sumcheck.cpp:654-729): tables add_L, add_R, L, R, O, lkp, lkp_O, mul, beta; twenty sums:
// c[0..3] add_L*beta*L, c[4..7] add_R*beta*R, c[8..11] lkp*beta*lkp_O (cubics), c[12..16] mul*beta*L*R (quartic), c[17..19] beta*O
struct GateLkp {
    static constexpr int NT = 9, NS = 20, NA = 5;
    static HB_HD void acc(F (&c)[20], const F (&b)[9], const F (&e)[9], const F &, const F &) {
        F d[9];
#pragma unroll
        for (int q = 0; q < 9; q++) d[q] = fsub(e[q], b[q]);
        cubic_acc(c, b[0], d[0], b[8], d[8], b[2], d[2]);
        cubic_acc(c + 4, b[1], d[1], b[8], d[8], b[3], d[3]);
        cubic_acc(c + 8, b[5], d[5], b[8], d[8], b[6], d[6]);
        quartic_acc(c + 12, b[7], d[7], b[8], d[8], b[2], d[2], b[3], d[3]);
        quad_acc(c + 17, b[8], d[8], b[4], d[4]);
    }
    static void combine(const F *c, CHP a, F *p) {                       // (:689-704)
        F C[4];
        for (int k = 0; k < 4; k++) C[k] = fadd(fadd(fmul(a[0], c[k]), fmul(a[1], c[4 + k])), fmul(a[4], c[8 + k]));
        p[0] = fmul(a[2], c[12]);
        p[1] = fadd(fmul(a[2], c[13]), C[0]);
        p[2] = fadd(fadd(fmul(a[2], c[14]), C[1]), fmul(a[3], c[17]));
        p[3] = fadd(fadd(fmul(a[2], c[15]), C[2]), fmul(a[3], c[18]));
        p[4] = fadd(fadd(fmul(a[2], c[16]), C[3]), fmul(a[3], c[19]));
    }
};
// round 0: polynomial only
template <class G>
__global__ void __launch_bounds__(256) k_gate_poly(GateTabsT<G::NT> t, size_t L, F a0, F a1, F *__restrict__ partials) {
    F c[G::NS];
#pragma unroll
    for (int q = 0; q < G::NS; q++) c[q] = fmake(0);
    for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < L; j += (size_t)gridDim.x * blockDim.x) {
        F b[G::NT], e[G::NT];
#pragma unroll
        for (int q = 0; q < G::NT; q++) { b[q] = ldF(t.s[q] + 2 * j); e[q] = ldF(t.s[q] + 2 * j + 1); }
        G::acc(c, b, e, a0, a1);
    }
    block_reduce_store<G::NS>(c, partials);
}
// rounds >= 1: fold the previous tables (4 -> 2 elements per thread and table) and accumulate this round's sums
template <class G>
__global__ void __launch_bounds__(256) k_gate_fold_poly(GateTabsT<G::NT> t, size_t L, F r, F a0, F a1, F *__restrict__ partials) {
    F c[G::NS];
#pragma unroll
    for (int q = 0; q < G::NS; q++) c[q] = fmake(0);
    for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < L; j += (size_t)gridDim.x * blockDim.x) {
        F b[G::NT], e[G::NT];
#pragma unroll
        for (int q = 0; q < G::NT; q++) {
            F x0 = ldF(t.s[q] + 4 * j), x1 = ldF(t.s[q] + 4 * j + 1), x2 = ldF(t.s[q] + 4 * j + 2), x3 = ldF(t.s[q] + 4 * j + 3);
            b[q] = fadd(x0, fmul(r, fsub(x1, x0))); e[q] = fadd(x2, fmul(r, fsub(x3, x2)));
            stF(t.d[q] + 2 * j, b[q]); stF(t.d[q] + 2 * j + 1, e[q]);
        }
        G::acc(c, b, e, a0, a1);
    }
    block_reduce_store<G::NS>(c, partials);
}
// one transcript step: combine the sums into the quartic (a..e), hash, check against the running sum, evaluate
template <class G>
static bool gate_round_host(const F *c, CHP a, F &rnd, F &sum, MHP poly_out, MHP r_out) {
    F p[5];
    G::combine(c, a, p);
    for (int q = 0; q < 5; q++) { rnd = mimc_hash(p[q], rnd); poly_out[q] = p[q]; }
    F s01 = fadd(fadd(fadd(p[0], p[1]), fadd(p[2], p[3])), fadd(p[4], p[4]));
    bool ok = feq(s01, sum);                                              // "Error in gate consistency 2" (:909-912, :710-713)
    sum = fadd(fmul(fadd(fmul(fadd(fmul(fadd(fmul(p[0], rnd), p[1]), rnd), p[2]), rnd), p[3]), rnd), p[4]);
    *r_out = rnd;
    return ok;
}
// inputs are preserved (the reference folds in place and afterwards only reads element 0 of each table: h_final)
template <class G>
static int gate_sumcheck_impl(hobbit_ctx *ctx, const F *const *tabs, size_t n, CHP h_a, MHP h_rand, MHP h_sum, MHP h_poly, MHP h_r, MHP h_final, int *h_check) {
    constexpr int NT = G::NT, NS = G::NS;
    int rounds = 0; while (((size_t)1 << rounds) < n) rounds++;
    if (((size_t)1 << rounds) != n || n < 2) return ctx->fail(HOBBIT_EINVAL, "gate_sumcheck: n must be a power of two >= 2");
    const int MAXB = 512;
    F rnd = *h_rand, sum = *h_sum; bool ok = true;
    const F a0 = h_a[0], a1 = h_a[1];
    std::vector<F> host[NT];
    size_t cur = n; int i = 0; bool pending = false;
    if (n > SC_TAIL) {
        size_t szA = n / 2, szB = n / 4;
        F *ws; HB_TRY(ctx->workspace((NT * (szA + szB) + (size_t)MAXB * NS + NS + 4) * sizeof(F), (void **)&ws));
        F *A = ws, *B = A + NT * szA, *part = B + NT * szB, *coef = part + (size_t)MAXB * NS;
        F *pin; HB_TRY(ctx->pinned(NS * sizeof(F), (void **)&pin));
        GateTabsT<NT> t;
        for (int q = 0; q < NT; q++) { t.s[q] = tabs[q]; t.d[q] = A + (size_t)q * szA; }
        bool toA = true;
        for (;; i++) {
            size_t L = n >> (i + 1);
            int nb = grid_for(L, 256, MAXB);
            if (i == 0) HB_LAUNCH(ctx, "k_gate_poly", (k_gate_poly<G>), dim3(nb), dim3(256), 0, t, L, a0, a1, part);
            else {
                HB_LAUNCH(ctx, "k_gate_fold_poly", (k_gate_fold_poly<G>), dim3(nb), dim3(256), 0, t, L, rnd, a0, a1, part);
                cur = 2 * L; toA = !toA;
                for (int q = 0; q < NT; q++) { t.s[q] = t.d[q]; t.d[q] = toA ? A + (size_t)q * szA : B + (size_t)q * szB; }
            }
            HB_LAUNCH(ctx, "k_sc_reduce12", k_sc_reduce<NS>, dim3(1), dim3(256), 0, part, nb, coef);
            HB_CHECK(ctx, hipMemcpyAsync(pin, coef, NS * sizeof(F), hipMemcpyDeviceToHost, ctx->stream));
            HB_TRY(ctx->sync());
            ok &= gate_round_host<G>(pin, h_a, rnd, sum, h_poly + 5 * i, h_r + i);
            if (cur <= 2 * SC_TAIL || i == rounds - 1) break;
        }
        for (int q = 0; q < NT; q++) { host[q].resize(cur); HB_CHECK(ctx, hipMemcpyAsync(host[q].data(), t.s[q], cur * sizeof(F), hipMemcpyDeviceToHost, ctx->stream)); }
        HB_TRY(ctx->sync());
        pending = true; i++;
    } else {
        for (int q = 0; q < NT; q++) { host[q].resize(n); HB_CHECK(ctx, hipMemcpyAsync(host[q].data(), tabs[q], n * sizeof(F), hipMemcpyDeviceToHost, ctx->stream)); }
        HB_TRY(ctx->sync());
    }
    auto fold = [&]() { for (int q = 0; q < NT; q++) for (size_t j = 0; j < cur / 2; j++) host[q][j] = fadd(host[q][2 * j], fmul(rnd, fsub(host[q][2 * j + 1], host[q][2 * j]))); cur /= 2; };
    if (pending) fold();
    for (; i < rounds; i++) {
        F c[NS]; for (int q = 0; q < NS; q++) c[q] = fmake(0);
        for (size_t j = 0; j < cur / 2; j++) {
            F b[NT], e[NT];
            for (int q = 0; q < NT; q++) { b[q] = host[q][2 * j]; e[q] = host[q][2 * j + 1]; }
            G::acc(c, b, e, a0, a1);
        }
        ok &= gate_round_host<G>(c, h_a, rnd, sum, h_poly + 5 * i, h_r + i);
        fold();
    }
    for (int q = 0; q < NT; q++) h_final[q] = host[q][0];
    *h_rand = rnd; *h_sum = sum; *h_check = ok ? 1 : 0;
    return 0;
}
int launch_gate_sumcheck(hobbit_ctx *ctx, const F *const tabs[6], size_t n, CHP h_a, MHP h_rand, MHP h_sum, MHP h_poly, MHP h_r, MHP h_final, int *h_check) {
    return gate_sumcheck_impl<GateStd>(ctx, tabs, n, h_a, h_rand, h_sum, h_poly, h_r, h_final, h_check);
}
// tabs: add_L, add_R, L, R, O, lkp, lkp_O, mul, beta; h_a: 5 coefficients; h_final: the nine folded values in that order
int launch_gate_lkp_sumcheck(hobbit_ctx *ctx, const F *const tabs[9], size_t n, CHP h_a, MHP h_rand, MHP h_sum, MHP h_poly, MHP h_r, MHP h_final, int *h_check) {
    return gate_sumcheck_impl<GateLkp>(ctx, tabs, n, h_a, h_rand, h_sum, h_poly, h_r, h_final, h_check);
}

// 3-product: polynomial of the current tables and fold with the PRE-round challenge in one pass
__global__ void __launch_bounds__(256) k_sc3_poly_fold(const F *__restrict__ s1, const F *__restrict__ s2, const F *__restrict__ s3,
                                                       F *__restrict__ d1, F *__restrict__ d2, F *__restrict__ d3, size_t L,
                                                       F r, F *__restrict__ partials) {
    F c[4] = {fmake(0), fmake(0), fmake(0), fmake(0)};
    for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < L; j += (size_t)gridDim.x * blockDim.x) {
        F x0 = ldF(s1 + 2 * j), x1 = ldF(s1 + 2 * j + 1), y0 = ldF(s2 + 2 * j), y1 = ldF(s2 + 2 * j + 1), z0 = ldF(s3 + 2 * j), z1 = ldF(s3 + 2 * j + 1);
        F dx = fsub(x1, x0), dy = fsub(y1, y0), dz = fsub(z1, z0);
        // (l1*l2)*l3 with the reference's operator grouping (src/polynomial.cpp:91-93,133-135)
        F qa = fmul(dx, dy), qb = fadd(fmul(dx, y0), fmul(x0, dy)), qc = fmul(x0, y0);
        c[0] = fadd(c[0], fmul(qa, dz));
        c[1] = fadd(c[1], fadd(fmul(qa, z0), fmul(qb, dz)));
        c[2] = fadd(c[2], fadd(fmul(qb, z0), fmul(qc, dz)));
        c[3] = fadd(c[3], fmul(qc, z0));
        stF(d1 + j, fadd(x0, fmul(r, dx))); stF(d2 + j, fadd(y0, fmul(r, dy))); stF(d3 + j, fadd(z0, fmul(r, dz)));
    }
    block_reduce_store<4>(c, partials);
}
// cubic round polynomial only / fold only (batch_3product_sumcheck hashes first, then folds: src/sumcheck.cpp:327-352)
__global__ void __launch_bounds__(256) k_sc3_poly(const F *__restrict__ s1, const F *__restrict__ s2, const F *__restrict__ s3, size_t L, F *__restrict__ partials) {
    F c[4] = {fmake(0), fmake(0), fmake(0), fmake(0)};
    for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < L; j += (size_t)gridDim.x * blockDim.x) {
        F x0 = ldF(s1 + 2 * j), x1 = ldF(s1 + 2 * j + 1), y0 = ldF(s2 + 2 * j), y1 = ldF(s2 + 2 * j + 1), z0 = ldF(s3 + 2 * j), z1 = ldF(s3 + 2 * j + 1);
        F dx = fsub(x1, x0), dy = fsub(y1, y0), dz = fsub(z1, z0);
        F qa = fmul(dx, dy), qb = fadd(fmul(dx, y0), fmul(x0, dy)), qc = fmul(x0, y0);
        c[0] = fadd(c[0], fmul(qa, dz));
        c[1] = fadd(c[1], fadd(fmul(qa, z0), fmul(qb, dz)));
        c[2] = fadd(c[2], fadd(fmul(qb, z0), fmul(qc, dz)));
        c[3] = fadd(c[3], fmul(qc, z0));
    }
    block_reduce_store<4>(c, partials);
}
__global__ void __launch_bounds__(256) k_fold3(const F *__restrict__ s1, const F *__restrict__ s2, const F *__restrict__ s3, F *__restrict__ d1, F *__restrict__ d2,
                                               F *__restrict__ d3, size_t L, F r) {
    for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < L; j += (size_t)gridDim.x * blockDim.x) {
        F x0 = ldF(s1 + 2 * j), x1 = ldF(s1 + 2 * j + 1), y0 = ldF(s2 + 2 * j), y1 = ldF(s2 + 2 * j + 1), z0 = ldF(s3 + 2 * j), z1 = ldF(s3 + 2 * j + 1);
        stF(d1 + j, fadd(x0, fmul(r, fsub(x1, x0)))); stF(d2 + j, fadd(y0, fmul(r, fsub(y1, y0)))); stF(d3 + j, fadd(z0, fmul(r, fsub(z1, z0))));
    }
}
int launch_sc3_poly(hobbit_ctx *ctx, const F *s1, const F *s2, const F *s3, size_t L, F *part, F *coef) {
    int nb = grid_for(L, 256, 1024);
    HB_LAUNCH(ctx, "k_sc3_poly", k_sc3_poly, dim3(nb), dim3(256), 0, s1, s2, s3, L, part);
    HB_LAUNCH(ctx, "k_sc_reduce4", k_sc_reduce<4>, dim3(1), dim3(256), 0, part, nb, coef);
    return 0;
}
int launch_fold3(hobbit_ctx *ctx, const F *s1, const F *s2, const F *s3, F *d1, F *d2, F *d3, size_t L, F r) {
    HB_LAUNCH(ctx, "k_fold3", k_fold3, dim3(grid_for(L, 256)), dim3(256), 0, s1, s2, s3, d1, d2, d3, L, r);
    return 0;
}
// one layer of the multiplication tree (src/sumcheck.cpp:88-110): in1[j] = x[2j], in2[j] = x[2j+1], tr[j] = x[2j]*x[2j+1]
__global__ void __launch_bounds__(256) k_mul_layer(const F *__restrict__ x, size_t n_out, F *__restrict__ in1, F *__restrict__ in2, F *__restrict__ tr) {
    for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < n_out; j += (size_t)gridDim.x * blockDim.x) {
        const F a = ldF(x + 2 * j), b = ldF(x + 2 * j + 1);
        stF(in1 + j, a); stF(in2 + j, b); stF(tr + j, fmul(a, b));
    }
}
int launch_mul_layer(hobbit_ctx *ctx, const F *x, size_t n_out, F *in1, F *in2, F *tr) {
    if (!n_out) return 0;
    HB_LAUNCH(ctx, "k_mul_layer", k_mul_layer, dim3(grid_for(n_out, 256)), dim3(256), 0, x, n_out, in1, in2, tr);
    return 0;
}

// host rounds of the 3-product sumcheck on tables of size sz (src/sumcheck.cpp:1981-2037)
static void sc3_host_tail(std::vector<F> &a, std::vector<F> &b, std::vector<F> &c3, F &rnd, int round, int rounds, MHP h_cpoly, MHP h_r) {
    size_t sz = a.size();
    for (int i = round; i < rounds; i++) {
        F pa = fmake(0), pb = fmake(0), pc = fmake(0), pd = fmake(0);
        for (size_t j = 0; j < sz / 2; j++) {
            F x0 = a[2 * j], y0 = b[2 * j], z0 = c3[2 * j];
            F dx = fsub(a[2 * j + 1], x0), dy = fsub(b[2 * j + 1], y0), dz = fsub(c3[2 * j + 1], z0);
            F qa = fmul(dx, dy), qb = fadd(fmul(dx, y0), fmul(x0, dy)), qc = fmul(x0, y0);
            pa = fadd(pa, fmul(qa, dz)); pb = fadd(pb, fadd(fmul(qa, z0), fmul(qb, dz)));
            pc = fadd(pc, fadd(fmul(qb, z0), fmul(qc, dz))); pd = fadd(pd, fmul(qc, z0));
            a[j] = fadd(x0, fmul(rnd, dx)); b[j] = fadd(y0, fmul(rnd, dy)); c3[j] = fadd(z0, fmul(rnd, dz));
        }
        h_r[i] = rnd;
        rnd = mimc_hash(rnd, pa); rnd = mimc_hash(rnd, pb); rnd = mimc_hash(rnd, pc); rnd = mimc_hash(rnd, pd);
        h_cpoly[4 * i] = pa; h_cpoly[4 * i + 1] = pb; h_cpoly[4 * i + 2] = pc; h_cpoly[4 * i + 3] = pd;
        sz /= 2;
    }
}
int launch_sumcheck3(hobbit_ctx *ctx, const F *v1, const F *v2, const F *v3, size_t n, F prev_r, MHP h_cpoly, MHP h_r, MHP h_vr, MHP h_final) {
    int rounds = 0; while (((size_t)1 << rounds) < n) rounds++;
    if (((size_t)1 << rounds) != n || n < 2) return ctx->fail(HOBBIT_EINVAL, "sumcheck3: n must be a power of two >= 2");
    const int MAXB = 1024;
    F rnd = prev_r;
    const F *s1 = v1, *s2 = v2, *s3 = v3;
    size_t cur = n;
    int i = 0;
    if (n > SC_TAIL) {
        size_t szA = n / 2, szB = n / 4;
        F *ws; HB_TRY(ctx->workspace((3 * szA + 3 * szB + (size_t)MAXB * 4 + 4) * sizeof(F), (void **)&ws));
        F *A = ws, *B = A + 3 * szA, *part = B + 3 * szB, *coef = part + (size_t)MAXB * 4;
        F *pin; HB_TRY(ctx->pinned(4 * sizeof(F), (void **)&pin));
        F *dst = A; size_t dsz = szA;
        for (; cur > SC_TAIL; i++) {
            size_t L = cur / 2;
            int nb = grid_for(L, 256, MAXB);
            HB_LAUNCH(ctx, "k_sc3_poly_fold", k_sc3_poly_fold, dim3(nb), dim3(256), 0, s1, s2, s3, dst, dst + dsz, dst + 2 * dsz, L, rnd, part);
            HB_LAUNCH(ctx, "k_sc_reduce4", k_sc_reduce<4>, dim3(1), dim3(256), 0, part, nb, coef);
            HB_CHECK(ctx, hipMemcpyAsync(pin, coef, 4 * sizeof(F), hipMemcpyDeviceToHost, ctx->stream));
            HB_TRY(ctx->sync());
            h_r[i] = rnd;                                           // randomness[i] = pre-round challenge
            for (int q = 0; q < 4; q++) { rnd = mimc_hash(rnd, pin[q]); h_cpoly[4 * i + q] = pin[q]; }
            s1 = dst; s2 = dst + dsz; s3 = dst + 2 * dsz; cur = L;
            if (dst == A) { dst = B; dsz = szB; } else { dst = A; dsz = szA; }
        }
    }
    std::vector<F> ta(cur), tb(cur), tc(cur);
    HB_CHECK(ctx, hipMemcpyAsync(ta.data(), s1, cur * sizeof(F), hipMemcpyDeviceToHost, ctx->stream));
    HB_CHECK(ctx, hipMemcpyAsync(tb.data(), s2, cur * sizeof(F), hipMemcpyDeviceToHost, ctx->stream));
    HB_CHECK(ctx, hipMemcpyAsync(tc.data(), s3, cur * sizeof(F), hipMemcpyDeviceToHost, ctx->stream));
    HB_TRY(ctx->sync());
    sc3_host_tail(ta, tb, tc, rnd, i, rounds, h_cpoly, h_r);
    rnd = mimc_hash(rnd, ta[0]); rnd = mimc_hash(rnd, tb[0]);      // v3[0] is not hashed (src/sumcheck.cpp:2043-2046)
    h_vr[0] = ta[0]; h_vr[1] = tb[0]; h_vr[2] = tc[0]; *h_final = rnd;
    return 0;
}

// ============================================================================================
// Elastic_PC open helpers (src/Elastic_PC.cpp:59-111 update_reply, :510-517 zero-chunk test; src/PC_utils.cpp:422-496)
// ============================================================================================
// *flag |= (some element of v is non-zero)
__global__ void k_any_nonzero(const F *__restrict__ v, size_t n, int *__restrict__ flag) {
    int nz = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const F a = ldF(v + i); nz |= (a.re | a.im) != 0; }
    if (__any(nz) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}
int launch_any_nonzero(hobbit_ctx *ctx, const F *v, size_t n, int *d_flag) {
    HB_LAUNCH(ctx, "k_any_nonzero", k_any_nonzero, dim3(grid_for(n, 256, 1024)), dim3(256), 0, v, n, d_flag);
    return 0;
}
// G[c*ldG + j] = T[j*ld + cols[c]], j < nrows: the queried columns of a row-major matrix, one per output row
__global__ void k_gather_cols(const F *__restrict__ T, size_t ld, uint32_t nrows, const uint32_t *__restrict__ cols, uint32_t ncols, F *__restrict__ G, size_t ldG) {
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= (size_t)ncols * nrows) return;
    const uint32_t c = (uint32_t)(t / nrows), j = (uint32_t)(t % nrows);
    stF(G + (size_t)c * ldG + j, ldF(T + (size_t)j * ld + cols[c]));
}
int launch_gather_cols(hobbit_ctx *ctx, const F *T, size_t ld, uint32_t nrows, const uint32_t *d_cols, uint32_t ncols, F *G, size_t ldG) {
    if (!ncols || !nrows) return 0;
    HB_LAUNCH(ctx, "k_gather_cols", k_gather_cols, dim3((unsigned)(((size_t)ncols * nrows + 255) / 256)), dim3(256), 0, T, ld, nrows, d_cols, ncols, G, ldG);
    return 0;
}
// out[cols[i] + j*ld] = vals[i], j < nrows: a column indicator weighted per column (recursive_prover_RS's second beta, src/PC_utils.cpp:489-496)
__global__ void k_spread_cols(const uint32_t *__restrict__ cols, const F *__restrict__ vals, uint32_t ncols, uint32_t nrows, size_t ld, F *__restrict__ out) {
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= (size_t)ncols * nrows) return;
    const uint32_t j = (uint32_t)(t / ncols), i = (uint32_t)(t % ncols);
    stF(out + (size_t)j * ld + cols[i], ldF(vals + i));
}
int launch_spread_cols(hobbit_ctx *ctx, const uint32_t *d_cols, const F *d_vals, uint32_t ncols, uint32_t nrows, size_t ld, F *out) {
    if (!ncols || !nrows) return 0;
    HB_LAUNCH(ctx, "k_spread_cols", k_spread_cols, dim3((unsigned)(((size_t)ncols * nrows + 255) / 256)), dim3(256), 0, d_cols, d_vals, ncols, nrows, ld, out);
    return 0;
}

// ============================================================================================
// Streaming provers: product layers of the stream (src/witness_stream.cpp:2413-2510 read_mul_tree_layer / read_mul_tree_data),
// partial evaluations (src/sumcheck.cpp:1327-1340, 949-959)
// ============================================================================================
// out[t] = prod_{j < seg} in[t*seg + j]
__global__ void k_seg_prod(const F *__restrict__ in, uint32_t seg, size_t n_out, F *__restrict__ out) {
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= n_out) return;
    const F *p = in + t * seg;
    F a = ldF(p);
    for (uint32_t j = 1; j < seg; j++) a = fmul(a, ldF(p + j));
    stF(out + t, a);
}
int launch_seg_prod(hobbit_ctx *ctx, const F *in, uint32_t seg, size_t n_out, F *out) {
    if (!n_out) return 0;
    HB_LAUNCH(ctx, "k_seg_prod", k_seg_prod, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, in, seg, n_out, out);
    return 0;
}
// a[j] = v[2j], b[j] = v[2j+1]
__global__ void k_deinterleave(const F *__restrict__ v, size_t n, F *__restrict__ a, F *__restrict__ b) {
    const size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (j >= n) return;
    stF(a + j, ldF(v + 2 * j)); stF(b + j, ldF(v + 2 * j + 1));
}
int launch_deinterleave(hobbit_ctx *ctx, const F *v, size_t n, F *a, F *b) {
    if (!n) return 0;
    HB_LAUNCH(ctx, "k_deinterleave", k_deinterleave, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, v, n, a, b);
    return 0;
}
// sum_j a[j] * b[j*sb] (* c[j] when c != NULL)
__global__ void __launch_bounds__(256) k_dot_gen(const F *__restrict__ a, const F *__restrict__ b, size_t sb, const F *__restrict__ c, size_t n, F *__restrict__ partials) {
    F s[1] = {fmake(0)};
    for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x) {
        F t = fmul(ldF(a + j), ldF(b + j * sb));
        if (c) t = fmul(t, ldF(c + j));
        s[0] = fadd(s[0], t);
    }
    block_reduce_store<1>(s, partials);
}
int launch_dot_gen(hobbit_ctx *ctx, const F *a, const F *b, size_t sb, const F *c, size_t n, F *part, F *out) {
    int nb = grid_for(n, 256, 1024);
    HB_LAUNCH(ctx, "k_dot_gen", k_dot_gen, dim3(nb), dim3(256), 0, a, b, sb, c, n, part);
    HB_LAUNCH(ctx, "k_sc_reduce", k_sc_reduce<1>, dim3(1), dim3(256), 0, part, nb, out);
    return 0;
}
// sum_j a[j] * F(sel[j])  (one_minus: 1 - F(sel[j]))
__global__ void __launch_bounds__(256) k_dot_i32(const F *__restrict__ a, const int32_t *__restrict__ sel, int one_minus, size_t n, F *__restrict__ partials) {
    F s[1] = {fmake(0)};
    for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x) {
        F v = fmake((uint64_t)(int64_t)sel[j]);
        if (one_minus) v = fsub(fmake(1), v);
        s[0] = fadd(s[0], fmul(ldF(a + j), v));
    }
    block_reduce_store<1>(s, partials);
}
int launch_dot_i32(hobbit_ctx *ctx, const F *a, const int32_t *sel, int one_minus, size_t n, F *part, F *out) {
    int nb = grid_for(n, 256, 1024);
    HB_LAUNCH(ctx, "k_dot_i32", k_dot_i32, dim3(nb), dim3(256), 0, a, sel, one_minus, n, part);
    HB_LAUNCH(ctx, "k_sc_reduce", k_sc_reduce<1>, dim3(1), dim3(256), 0, part, nb, out);
    return 0;
}
// y[j] = F(sel[j]) or 1 - F(sel[j])
__global__ void k_i32_to_F(const int32_t *__restrict__ sel, int one_minus, size_t n, F *__restrict__ y) {
    const size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (j >= n) return;
    F v = fmake((uint64_t)(int64_t)sel[j]);
    if (one_minus) v = fsub(fmake(1), v);
    stF(y + j, v);
}
int launch_i32_to_F(hobbit_ctx *ctx, const int32_t *sel, int one_minus, size_t n, F *y) {
    if (!n) return 0;
    HB_LAUNCH(ctx, "k_i32_to_F", k_i32_to_F, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, sel, one_minus, n, y);
    return 0;
}

// ============================================================================================
// Fill / copy as ordinary kernels.  The runtime's own hipMemsetAsync / device-to-device hipMemcpyAsync come with a 40-100 us idle gap in
// front of their internal kernels on this driver (rocprofv3 kernel trace of one open: 10 fills = 1.0 ms, 12 copies = 0.5 ms of GPU idle
// time); a plain launch has the usual 3-6 us.
// ============================================================================================
__global__ void k_zero16(uint4 *__restrict__ p, size_t n16) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(0, 0, 0, 0);
}
__global__ void k_copy16(uint4 *__restrict__ d, const uint4 *__restrict__ s, size_t n16) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
int launch_zero(hobbit_ctx *ctx, void *p, size_t bytes) {
    if (!bytes) return 0;
    if ((bytes & 15) || ((uintptr_t)p & 15)) { hipError_t e = hipMemsetAsync(p, 0, bytes, ctx->stream); return e == hipSuccess ? 0 : ctx->hip(e, "hipMemsetAsync"); }
    HB_LAUNCH(ctx, "k_zero16", k_zero16, dim3(grid_for(bytes / 16, 256, 8192)), dim3(256), 0, reinterpret_cast<uint4 *>(p), bytes / 16);
    return 0;
}
int launch_copy(hobbit_ctx *ctx, void *d, const void *s, size_t bytes) {
    if (!bytes) return 0;
    if ((bytes & 15) || ((uintptr_t)d & 15) || ((uintptr_t)s & 15)) {
        hipError_t e = hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, ctx->stream); return e == hipSuccess ? 0 : ctx->hip(e, "hipMemcpyAsync");
    }
    HB_LAUNCH(ctx, "k_copy16", k_copy16, dim3(grid_for(bytes / 16, 256, 8192)), dim3(256), 0, reinterpret_cast<uint4 *>(d), reinterpret_cast<const uint4 *>(s), bytes / 16);
    return 0;
}

}  // namespace hobbit
