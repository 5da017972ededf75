// hobbit_ctx.hpp -- context object behind the C ABI (include/hobbit_hip.h): device, stream,
// error string, per-kernel HIP-event profiler, cached twiddle tables and the uploaded expander
// graphs in their device (gather / sliced-ELL) form.
#pragma once
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include "hobbit_field.hpp"
#include "../../include/hobbit_hip.h"

namespace hobbit {

// One SpMV step of the recursive expander encode, in gather form over the codeword buffer:
//   cw[out_off + t] = sum_e w_e * cw[in_off + idx_e],  t in [0, out_len)
// Outputs are ordered by in-degree and cut into slices of ENC_SW; a slice stores its edges k-major
// (edge k of its l-th output at slice_ptr + k*ENC_SW + l), padded with zero-weight records to a
// multiple of ENC_SPLIT past the slice's widest row (sorting keeps that within ~10 % of the real
// edge count: 76 288 records for the 69 084 edges of n = 4096, against 124 928 unsorted), so a
// wavefront reads 64 consecutive records per instruction and needs no per-lane bounds.
// slice_out maps (slice, l) back to the output index (0xFFFFFFFF = padding lane).
// slice geometry shared by the host preprocessing and k_encode (see hobbit_kernels.hip)
static constexpr uint32_t ENC_SW = 16;                 // outputs per slice
static constexpr uint32_t ENC_SPLIT = 64 / ENC_SW;     // lane groups sharing one output's edges
static constexpr uint32_t ENC_UNROLL = 4;              // edge records in flight per lane

struct EncStep {
    uint32_t in_off, out_off, out_len, n_slices;
    uint32_t slice_base;   // index of this step's first slice in slice_ptr/slice_width
    uint32_t sw;           // outputs per slice of this step: 64 (one output per lane, no cross-lane combine) for the wide steps,
                           // ENC_SW (lane groups share an output) for the narrow ones, where a slice per wave would leave waves idle
    uint32_t out_base;     // index of this step's first entry in slice_out (sw entries per slice)
};
static constexpr uint32_t ENC_WIDE_MIN = 128;          // steps with at least this many outputs use sw = 64 (one fold per lane and no cross-lane
                                                       // combine: the narrow form's per-slice fold + shuffles cost more than its shorter rows save --
                                                       // middle pass at 2^28: 2.61 ms at 512, 2.36 at 256, 2.28 at 128, 2.46 at 32)

struct HostGraph {          // one uploaded level (src/expanders.h:7-16)
    long long L = 0, R = 0;
    int degree = 0;
    std::vector<long long> nbr;
    std::vector<F> w;
};

// One WIDE SpMV step in "fat" form for the persistent encode kernels (hobbit_kernels.hip, k_enc_fat): a workgroup stays on its CU for the
// whole launch and walks the columns; every lane of its `ncons` consumer waves owns `nout` outputs of the step for good, and the edge
// records of those outputs -- 32-bit weight, 16-bit byte offset of the input inside the step's window -- live in REGISTERS, loaded once per
// launch (position j of a lane has cap[j] register slots; a wave uses the first w[wave][j] of them, a multiple of 4; unused slots have
// weight 0).  Outputs are dealt to lanes in order of in-degree, serpentine over the positions, so that a lane's heavy output is paired
// with a light one.  Built by hobbit_graph_finalize when the step's degrees fit the caps the kernel was compiled for.
struct FatStep {
    bool ok = false;
    uint32_t nout = 0, ncons = 0, cap[3] = {0, 0, 0};
    uint32_t in_off = 0, in_len = 0, out_off = 0, out_len = 0;
    uint32_t *d_wt = nullptr, *d_ot = nullptr, *d_oidx = nullptr, *d_w = nullptr;
    size_t slots_used = 0;                   // sum over waves and positions of the widths (x 64 = padded edge count)
};
static constexpr uint32_t FAT_A_NOUT = 2, FAT_A_CONS = 7, FAT_A_CAP0 = 72, FAT_A_CAP1 = 48;          // C_0 of n = 4096: 864 outputs, in-degree 42.7 +- 6.5
static constexpr uint32_t FAT_C1_NOUT = 1, FAT_C1_CONS = 3, FAT_C1_CAP0 = 64;                        // C_1: 182 outputs, in-degree 42.7 (max ~62): three waves, a few workgroups per CU
static constexpr uint32_t FAT_D_NOUT = 3, FAT_D_CONS = 8, FAT_D_CAP0 = 28, FAT_D_CAP1 = 16, FAT_D_CAP2 = 12;   // D_0: 1463 outputs, in-degree 12.2 +- 3.5

// The NARROW dependent steps between the first and the last one (C_1 .. D_1 of n = 4096: 182, 38, 8, 19, 66, 309 outputs) for the persistent
// kernel k_enc_mid: one workgroup of MID_WAVES waves per CU walks the columns; step s gives each of its outputs to 2^lg[s] adjacent lanes of one
// wave, which split its in-edges round-robin, keep their share of the records in registers (cap[s] slots per lane) and combine by shuffles.
static constexpr uint32_t MID_WAVES = 16, MID_MAX_STEPS = 6;
static constexpr uint32_t MID_CAP[MID_MAX_STEPS] = {16, 4, 8, 8, 4, 16};       // C_1 (4 lanes per output), C_2 (16), C_3 (8), D_3 (2), D_2 (8), D_1 (2)
struct MidCode {
    bool ok = false;
    uint32_t nsteps = 0, win_off = 0, win_len = 0, in_len = 0, st_lo = 0;     // window [win_off, win_off + win_len) of the codeword; the first in_len come from memory, [st_lo, win_len) go back
    uint32_t lg[MID_MAX_STEPS] = {}, R[MID_MAX_STEPS] = {}, out_rel[MID_MAX_STEPS] = {};
    uint32_t *d_wt = nullptr, *d_ot = nullptr, *d_oidx = nullptr, *d_w = nullptr;
};

struct DeviceCode {         // finalized code for one message length n
    long long n = 0, len = 0;
    bool small_weights = true;               // all weights real and < 2^32
    std::vector<EncStep> steps;
    EncStep *d_steps = nullptr;
    uint32_t *d_slice_ptr = nullptr, *d_slice_width = nullptr, *d_slice_out = nullptr;   // per slice: offset, records per output, output ids
    uint2 *d_edges32 = nullptr;              // {idx, w32}          (small_weights)
    uint32_t *d_eidx = nullptr; F *d_ew = nullptr;   // general weights
    size_t n_edges_padded = 0, n_edges = 0;
    FatStep fatA, fatC1, fatD;               // first, second and last step in fat form (deep codes only: n = 4096)
    MidCode mid;                             // the steps between them
    // H^T in CSR (evaluate_parity_matrix), built on first use
    uint32_t *d_pm_rowptr = nullptr, *d_pm_idx = nullptr; F *d_pm_w = nullptr; size_t pm_rows = 0;
};

struct ProfEntry { std::vector<std::pair<hipEvent_t, hipEvent_t>> ev; double ms = 0; long long launches = 0; };

}  // namespace hobbit

namespace hobbit { struct Mailbox { F vals[16]; uint32_t flag; uint32_t pad[3]; }; }

struct hobbit_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    // second stream + events: the commit's layout change (HBM-bound) of one chunk group runs beside the next group's row FFT (VALU-bound).
    // Event slots: [0, 56) the commit pipeline's chunk groups (tensorcode_chunks clamps its group count to 56), 60-63 open_impl's cross-stream
    // fences, 64/65 the pipeline's opening / closing brackets.
    hipStream_t side = nullptr; hipEvent_t side_ev[66] = {};
    // commit_impl <-> launch_encode: "do not materialise the zero tail of the codewords" (asked / done); see hobbit_commitment::rows_valid
    bool enc_skip_tail = false, enc_tail_skipped = false;
    // host -> device streaming of a caller's pageable polynomial (hobbit_commit_standard_host): copy stream, two pinned staging pieces, events
    hipStream_t up_stream = nullptr; void *up_pin[2] = {nullptr, nullptr}; hipEvent_t up_done[2] = {}; hipEvent_t up_ready[64] = {};
    static constexpr size_t UP_PIECE = (size_t)64 << 20;
    int up_init() {
        if (up_stream) return 0;
        if (hipStreamCreateWithFlags(&up_stream, hipStreamNonBlocking) != hipSuccess) { err = "upload stream create failed"; return HOBBIT_EHIP; }
        for (int i = 0; i < 2; i++) {
            if (hipHostMalloc(&up_pin[i], UP_PIECE, hipHostMallocDefault) != hipSuccess) { err = "upload staging hipHostMalloc failed"; return HOBBIT_ENOMEM; }
            if (hipEventCreateWithFlags(&up_done[i], hipEventDisableTiming) != hipSuccess) { err = "upload event create failed"; return HOBBIT_EHIP; }
        }
        for (auto &e : up_ready) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { err = "upload event create failed"; return HOBBIT_EHIP; }
        return 0;
    }
    int side_init() {
        if (side) return 0;
        if (hipStreamCreateWithFlags(&side, hipStreamNonBlocking) != hipSuccess) { err = "side stream create failed"; return HOBBIT_EHIP; }
        for (auto &e : side_ev) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { err = "side event create failed"; return HOBBIT_EHIP; }
        return 0;
    }
    // helper context (own stream, scratch, staging, mailbox) for the one part of the open that is independent of what follows it:
    // shockwave_prove(C_c) runs there on a second host thread beside P5 and shockwave_prove(C_f) (open_impl)
    hobbit_ctx *helper = nullptr, *helper2 = nullptr;      // helper2: stream + scratch of the inner commitments, queued from the main thread
    std::string err;
    // profiler
    int prof_on = 0;
    std::map<std::string, hobbit::ProfEntry> prof;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    // twiddles: logn -> device table of 2^(logn-1) forward (and inverse) roots
    std::map<int, hobbit::F *> tw_fwd, tw_inv;
    std::map<int, hobbit::F *> tw2d_fwd;          // inter-stage twiddles of the long transforms, [n1][k2] = w^(n1 k2), read coalesced
    std::map<int, hobbit::F *> tw_r8;             // tables of k_fft_r8 per log2(N) (forward): radix-8 passes, then the radix-R tail
    // radix-8 per-pass tables of the FFT-4096 kernel ([7][8] | [7][64] | [7][512]), fwd / inv
    hobbit::F *tw8[2] = {nullptr, nullptr};
    hobbit::F tw8_w8[2], tw8_w83[2]; int tw8_w4_plus_i[2] = {0, 0};
    // graphs
    std::map<std::pair<int, int>, hobbit::HostGraph> graphs;   // (dep, kind)
    hobbit::DeviceCode code;
    // the reference's globals has_lookups / lookup_rand[0..1] (src/main.cpp:67,70), set through hobbit_set_lookups
    bool has_lookups = false; hobbit::F lookup_rand[2];
    // scratch
    void *ws = nullptr; size_t ws_bytes = 0;
    // second scratch: the row-major FFT output of a tensor code before its transpose
    void *ws2 = nullptr; size_t ws2_bytes = 0;
    int workspace2(size_t bytes, void **p) {
        if (bytes > ws2_bytes) {
            if (ws2) { hipStreamSynchronize(stream); hipFree(ws2); ws2 = nullptr; ws2_bytes = 0; }
            if (hipMalloc(&ws2, bytes) != hipSuccess) { err = "workspace2 hipMalloc failed"; return HOBBIT_ENOMEM; }
            ws2_bytes = bytes;
        }
        *p = ws2; return 0;
    }
    // Size-keyed free list for the device buffers of short-lived objects (a streaming commit or opening allocates five to nine buffers and
    // releases them at its end: at the MLP config's shape the hipFree calls cost as much as the whole commit, 0.9 ms).  One stream per
    // context, so a buffer handed back while kernels still read it is safe to hand out again: the next user is ordered behind them.
    std::multimap<size_t, void *> pool; size_t pool_bytes = 0;
    std::map<void *, size_t> pooled;             // live hobbit_malloc allocations that came from the pool (pointer -> size)
    int pool_get(size_t bytes, void **p) {
        auto it = pool.find(bytes);
        if (it != pool.end()) { *p = it->second; pool_bytes -= bytes; pool.erase(it); return 0; }
        if (hipMalloc(p, bytes) != hipSuccess) {             // make room: drop everything parked, try once more
            pool_drain();
            if (hipMalloc(p, bytes) != hipSuccess) { *p = nullptr; err = "device allocation failed"; return HOBBIT_ENOMEM; }
        }
        return 0;
    }
    void pool_put(size_t bytes, void *p) {
        if (!p) return;
        if (pool.size() >= 48 || pool_bytes + bytes > ((size_t)8 << 30)) { hipStreamSynchronize(stream); hipFree(p); return; }
        pool.emplace(bytes, p); pool_bytes += bytes;
    }
    void pool_drain() {
        if (pool.empty()) return;
        hipStreamSynchronize(stream);
        for (auto &kv : pool) hipFree(kv.second);
        pool.clear(); pool_bytes = 0;
    }
    // small pinned host buffer for the per-round coefficient read-back of the sumchecks
    void *pin = nullptr; size_t pin_bytes = 0;
    int pinned(size_t bytes, void **p) {
        if (bytes > pin_bytes) {
            if (pin) hipHostFree(pin);
            if (hipHostMalloc(&pin, bytes < 4096 ? 4096 : bytes) != hipSuccess) { pin = nullptr; pin_bytes = 0; err = "hipHostMalloc failed"; return HOBBIT_ENOMEM; }
            pin_bytes = bytes < 4096 ? 4096 : bytes;
        }
        *p = pin; return 0;
    }
    // Pinned staging arena of the open path.  hipMemcpyAsync to or from PAGEABLE host memory is not asynchronous on this runtime: a
    // device-to-host copy into a caller's buffer first waits for everything queued on the stream (rocprofv3: 100-170 us of GPU idle time
    // after every such copy, 2 ms per open).  So host inputs are copied here and uploaded from here, results land here and reach the
    // caller's buffers (`deferred`) after the closing synchronisation of the outermost library call (`stage_depth`).
    uint8_t *stage = nullptr; size_t stage_cap = 0, stage_off = 0; int stage_depth = 0;
    struct Deferred { void *dst; const void *src; size_t bytes; };
    std::vector<Deferred> deferred;
    void *stage_alloc(size_t bytes) {
        if (!stage) { if (hipHostMalloc((void **)&stage, (size_t)24 << 20) != hipSuccess) { stage = nullptr; return nullptr; } stage_cap = (size_t)24 << 20; }
        const size_t a = (stage_off + 63) & ~(size_t)63;
        if (a + bytes > stage_cap) return nullptr;
        stage_off = a + bytes; return stage + a;
    }
    // pinned, device-visible buffer for small per-launch constant tables (read in place by kernels)
    void *pinc = nullptr; size_t pinc_bytes = 0;
    int pinned_const(size_t bytes, void **p) {
        if (bytes > pinc_bytes) {
            hipStreamSynchronize(stream);
            if (pinc) hipHostFree(pinc);
            size_t want = bytes < 65536 ? 65536 : bytes;
            if (hipHostMalloc(&pinc, want) != hipSuccess) { pinc = nullptr; pinc_bytes = 0; err = "hipHostMalloc failed"; return HOBBIT_ENOMEM; }
            pinc_bytes = want;
        }
        *p = pinc; return 0;
    }
    // arena of the open phase (tables that live across several sumchecks), kept between calls
    void *ws3 = nullptr; size_t ws3_bytes = 0;
    int workspace3(size_t bytes, void **p) {
        if (bytes > ws3_bytes) {
            if (ws3) { hipStreamSynchronize(stream); hipFree(ws3); ws3 = nullptr; ws3_bytes = 0; }
            if (hipMalloc(&ws3, bytes) != hipSuccess) { err = "workspace3 hipMalloc failed"; return HOBBIT_ENOMEM; }
            ws3_bytes = bytes;
        }
        *p = ws3; return 0;
    }
    // scratch of the inner-PCS provers (whir / shockwave), separate from the open arena that holds their inputs
    void *ws4 = nullptr; size_t ws4_bytes = 0;
    int workspace4(size_t bytes, void **p) {
        if (bytes > ws4_bytes) {
            if (ws4) { hipStreamSynchronize(stream); hipFree(ws4); ws4 = nullptr; ws4_bytes = 0; }
            if (hipMalloc(&ws4, bytes) != hipSuccess) { err = "workspace4 hipMalloc failed"; return HOBBIT_ENOMEM; }
            ws4_bytes = bytes;
        }
        *p = ws4; return 0;
    }
    // one retired commitment's buffers, kept for the next commit of the same shape (a 2^28 commit
    // owns 16.5 GiB; re-allocating it per call would dominate a repeated-commit loop)
    void *spare_tensor = nullptr; size_t spare_tensor_bytes = 0;
    void *spare_levels = nullptr; size_t spare_levels_bytes = 0;

    int fail(int code, const std::string &msg) { err = msg; return code; }
    int hip(hipError_t e, const char *what) {
        if (e == hipSuccess) return 0;
        err = std::string(what) + ": " + hipGetErrorString(e);
        return HOBBIT_EHIP;
    }
    // prof_on: 0 off, 1 every launch, 2 only the commit's bulk kernels (a handful of launches per step: the event brackets then
    // cost nothing measurable, whereas bracketing the ~500 small launches of an open adds milliseconds of host time per step)
    std::vector<hipEvent_t> ev_pool;
    hipEvent_t ev_get() { if (!ev_pool.empty()) { hipEvent_t e = ev_pool.back(); ev_pool.pop_back(); return e; } hipEvent_t e; hipEventCreate(&e); return e; }
    static bool prof_bulk(const char *n) {
        return !strncmp(n, "k_leaf_chain", 12) || !strncmp(n, "k_fft4096", 9) || !strncmp(n, "k_encode", 8) || !strncmp(n, "k_enc_", 6) || !strcmp(n, "k_transpose") ||
               !strncmp(n, "k_inner_digests", 15) || !strncmp(n, "k_chain_digests", 15) || !strcmp(n, "k_aggregate");
    }
    bool prof_cur = false;
    void prof_begin(const char *name) {
        prof_cur = prof_on == 1 || (prof_on == 2 && prof_bulk(name));
        if (!prof_cur) return;
        hipEvent_t a = ev_get(), b = ev_get();
        hipEventRecord(a, stream);
        prof[name].ev.emplace_back(a, b);
    }
    void prof_end(const char *name) {
        if (!prof_cur) return;
        auto &e = prof[name]; hipEventRecord(e.ev.back().second, stream); e.launches++;
    }
    void prof_collect() {
        for (auto &kv : prof) {
            for (auto &p : kv.second.ev) {
                hipEventSynchronize(p.second);
                float ms = 0; hipEventElapsedTime(&ms, p.first, p.second);
                kv.second.ms += ms; ev_pool.push_back(p.first); ev_pool.push_back(p.second);
            }
            kv.second.ev.clear();
        }
    }
    // ---- mailbox: the one-workgroup reduction kernel of a sumcheck round writes its few coefficients straight into coherent pinned
    // host memory and then a sequence number; the host spins on that word (HOBBIT_MBOX=sync: sleeps in hipStreamSynchronize)
    // instead of queueing a copy and synchronising.
    hobbit::Mailbox *mbox = nullptr; unsigned *d_ticket = nullptr; uint32_t mbox_seq = 0;
    int mailbox(hobbit::Mailbox **m, unsigned **ticket) {
        if (!mbox) {
            if (hipHostMalloc((void **)&mbox, 4096, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) { mbox = nullptr; err = "mailbox hipHostMalloc failed"; return HOBBIT_ENOMEM; }
            memset((void *)mbox, 0, 4096);
            if (hipMalloc((void **)&d_ticket, 256) != hipSuccess) { err = "mailbox ticket alloc failed"; return HOBBIT_ENOMEM; }
            if (hipMemsetAsync(d_ticket, 0, 256, stream) != hipSuccess) { err = "mailbox ticket memset failed"; return HOBBIT_EHIP; }
        }
        *m = mbox; *ticket = d_ticket; return 0;
    }
    // wait until the kernel tagged `seq` has posted; bounded: gives up (error) once the stream has drained without the post
    // Drain the stream.  hipStreamSynchronize sleeps on an interrupt: in the rocprofv3 trace of an open nine such waits were each followed
    // by ~92 us of GPU idle time before the next kernel.  The default polls the stream's completion instead (a wait longer than 1 ms falls
    // back to the sleeping call; HOBBIT_SYNC=sleep: always sleep).  Alternating same-box runs of commit + open at 2^28: 39.45 / 39.75 ms
    // sleeping, 39.18 / 39.44 ms polling.  (The other configs of scripts/bench_configs.py do not move; what looked like a regression of
    // the 2^24 sumcheck there was the order of the runs -- the first process on a fresh box is the fast one whichever mode it uses.)
    int sync_mode = -1;
    // time this context's host thread spent waiting for the device (stream drains, mailbox spins): HOBBIT_TRACE=host prints it per stage,
    // the rest of a stage's time is host work the GPU may be waiting for
    uint64_t wait_ns = 0;
    struct WaitClock {
        hobbit_ctx *c; std::chrono::steady_clock::time_point t0;
        explicit WaitClock(hobbit_ctx *ctx) : c(ctx), t0(std::chrono::steady_clock::now()) {}
        ~WaitClock() { c->wait_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
    };
    int sync() {
        if (sync_mode < 0) { const char *e = getenv("HOBBIT_SYNC"); sync_mode = (e && !strcmp(e, "sleep")) ? 0 : 1; }
        WaitClock wc(this);
        if (sync_mode == 0) return hip(hipStreamSynchronize(stream), "stream sync");
        const auto t_start = wc.t0;
        for (uint32_t it = 1;; it++) {
            const hipError_t q = hipStreamQuery(stream);
            if (q == hipSuccess) return 0;
            if (q != hipErrorNotReady) return hip(q, "stream sync");
            if ((it & 0x3F) == 0 && std::chrono::steady_clock::now() - t_start > std::chrono::milliseconds(1)) return hip(hipStreamSynchronize(stream), "stream sync");
        }
    }
    int mbox_mode = -1;
    int mbox_wait(uint32_t seq) {
        if (mbox_mode < 0) { const char *e = getenv("HOBBIT_MBOX"); mbox_mode = (e && !strcmp(e, "sync")) ? 0 : 1; }
        WaitClock wc(this);
        if (mbox_mode == 0) {                    // sleep in the runtime; the post is complete when the kernel is
            int r = hip(hipStreamSynchronize(stream), "mailbox wait");
            if (r) return r;
            if (mbox->flag != seq) { err = "mailbox: kernel finished without posting"; return HOBBIT_EHIP; }
            return 0;
        }
        // the runtime queues launches lazily: a query makes it submit what is pending before we start spinning
        hipError_t q = hipStreamQuery(stream);
        if (q != hipSuccess && q != hipErrorNotReady) return hip(q, "mailbox wait");
        const auto t_start = std::chrono::steady_clock::now();
        for (uint64_t it = 1;; it++) {
            if (__atomic_load_n(&mbox->flag, __ATOMIC_ACQUIRE) == seq) return 0;
            if ((it & 0xFFFF) == 0) {
                if (std::chrono::steady_clock::now() - t_start > std::chrono::seconds(60)) { err = "mailbox: no post within 60 s (kernel hung?)"; return HOBBIT_EHIP; }
                q = hipStreamQuery(stream);
                if (q == hipSuccess) { if (__atomic_load_n(&mbox->flag, __ATOMIC_ACQUIRE) == seq) return 0; err = "mailbox: kernel finished without posting"; return HOBBIT_EHIP; }
                if (q != hipErrorNotReady) return hip(q, "mailbox wait");
            }
        }
    }
    // a caller may lend `workspace` another buffer for the calls it queues on a stream of its own (open_impl: the query answers on the
    // inner commitments' stream), so that they do not share scratch with what runs on the context's stream at the same time
    void *ws_lent = nullptr; size_t ws_lent_bytes = 0;
    int workspace(size_t bytes, void **p) {
        if (ws_lent) {
            if (bytes > ws_lent_bytes) { err = "lent workspace too small"; return HOBBIT_ESTATE; }
            *p = ws_lent; return 0;
        }
        if (bytes > ws_bytes) {
            if (ws) { hipStreamSynchronize(stream); hipFree(ws); ws = nullptr; ws_bytes = 0; }
            hipError_t e = hipMalloc(&ws, bytes);
            if (e != hipSuccess) { err = "workspace hipMalloc failed"; return HOBBIT_ENOMEM; }
            ws_bytes = bytes;
        }
        *p = ws; return 0;
    }
};

// launch + optional per-kernel event bracket; checks the launch error
#define HB_LAUNCH(ctx, name, kern, grid, block, lds, ...)                                   \
    do {                                                                                     \
        (ctx)->prof_begin(name);                                                             \
        hipLaunchKernelGGL(kern, grid, block, lds, (ctx)->stream, __VA_ARGS__);              \
        (ctx)->prof_end(name);                                                               \
        hipError_t e__ = hipGetLastError();                                                  \
        if (e__ != hipSuccess) return (ctx)->hip(e__, name);                                 \
    } while (0)
#define HB_CHECK(ctx, call) do { int r__ = (ctx)->hip((call), #call); if (r__) return r__; } while (0)
#define HB_TRY(expr) do { int r__ = (expr); if (r__) return r__; } while (0)
