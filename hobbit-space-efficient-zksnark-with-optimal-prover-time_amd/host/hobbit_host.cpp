// hobbit_host.cpp -- bodies of the reference-shaped host interface (hobbit_host.hpp): marshal
// std::vector arguments to device buffers and call the C ABI.  See the header for the mapping.
#include "hobbit_host.hpp"
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <set>

int tensor_row_size = 128;        /* src/main.cpp:31 */
size_t BUFFER_SPACE = 0;          /* src/main.cpp:38 */
bool linear_time = false;         /* src/Our_PC.cpp:21 */
int aggregation_queries = 1900;   /* src/Elastic_PC.cpp:10 */
graph _C[100], D[100];

static hobbit_ctx *g_ctx = nullptr;
static hobbit_commitment *g_commit = nullptr;
static int g_commit_K = 0, g_commit_trs = 0; static size_t g_commit_cols = 0;
static void *g_poly_dev = nullptr; static size_t g_poly_n = 0;      // device copy of the committed polynomial, kept for open_standard
static uint8_t g_commit_root[32];
// One older commitment stays alive beside the current one: prove_circuit_standard (src/main.cpp:985-1087) commits to the circuit polynomial
// and to the witness before it opens either.  open_standard finds its commitment by the root of the Commitment_MT it is handed.
struct CommitRec { hobbit_commitment *c = nullptr; int K = 0, trs = 0; size_t cols = 0; void *poly = nullptr; size_t n = 0; uint8_t root[32]; };
static CommitRec g_prev;
static void free_prev() {
    if (g_prev.c) hobbit_commitment_free(g_prev.c);
    if (g_prev.poly) hobbit_free(hobbit_host_ctx(), g_prev.poly);
    g_prev = CommitRec();
}
static void swap_with_prev() {
    CommitRec cur; cur.c = g_commit; cur.K = g_commit_K; cur.trs = g_commit_trs; cur.cols = g_commit_cols; cur.poly = g_poly_dev; cur.n = g_poly_n; memcpy(cur.root, g_commit_root, 32);
    g_commit = g_prev.c; g_commit_K = g_prev.K; g_commit_trs = g_prev.trs; g_commit_cols = g_prev.cols; g_poly_dev = g_prev.poly; g_poly_n = g_prev.n; memcpy(g_commit_root, g_prev.root, 32);
    g_prev = cur;
}
// make the commitment whose tree is `MT` the current one (no-op when it already is, or when the caller passes no tree)
static void select_commitment(const vector<vector<_hash>> &MT) {
    if (MT.empty() || MT.back().empty() || !g_commit) return;
    if (!memcmp(MT.back()[0].arr, g_commit_root, 32)) return;
    if (g_prev.c && !memcmp(MT.back()[0].arr, g_prev.root, 32)) { swap_with_prev(); return; }
    printf("Error: open_standard on a commitment this process no longer holds (the mirror keeps the last two)\n"); exit(-1);
}
static hobbit_host_open_transcript g_open;
hobbit_host_open_transcript &hobbit_host_last_open() { return g_open; }

#define HCHK(call) do { int rc__ = (call); if (rc__ != 0) { printf("Error in %s: %s\n", #call, hobbit_last_error(g_ctx)); exit(-1); } } while (0)

// HOBBIT_HOST_TIMING=1: every streaming entry point prints its wall time and the share spent in the stream readers / the uploads
static const bool g_timing = getenv("HOBBIT_HOST_TIMING") != nullptr;
static double g_t_read = 0, g_t_up = 0;
static inline double now_s() { return std::chrono::duration_cast<std::chrono::duration<double>>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct PhaseTimer {
    const char *name; double t0, r0, u0;
    explicit PhaseTimer(const char *n) : name(n), t0(now_s()), r0(g_t_read), u0(g_t_up) {}
    ~PhaseTimer() { if (g_timing) printf("[hobbit] %-28s wall %9.3f ms  stream readers %9.3f ms  uploads %8.3f ms\n", name, 1e3 * (now_s() - t0), 1e3 * (g_t_read - r0), 1e3 * (g_t_up - u0)); }
};
#define TIMED_READ(stmt) do { double t__ = now_s(); stmt; g_t_read += now_s() - t__; } while (0)
#define TIMED_UP(stmt) do { double t__ = now_s(); stmt; g_t_up += now_s() - t__; } while (0)

hobbit_ctx *hobbit_host_ctx() {
    if (!g_ctx) {
        const char *d = getenv("HOBBIT_DEVICE");
        int rc = hobbit_ctx_create(d ? atoi(d) : 0, &g_ctx);
        if (rc != 0) { printf("Error: no usable MI355X / HIP device (rc=%d); libhobbit_hip has no CPU fallback\n", rc); exit(-1); }
    }
    return g_ctx;
}
void hobbit_host_shutdown() {
    if (g_ctx) free_prev();
    if (g_commit) { hobbit_commitment_free(g_commit); g_commit = nullptr; }
    if (g_poly_dev) { hobbit_free(g_ctx, g_poly_dev); g_poly_dev = nullptr; g_poly_n = 0; }
    if (g_ctx) { hobbit_ctx_destroy(g_ctx); g_ctx = nullptr; }
}
hobbit_commitment *hobbit_host_last_commitment() { return g_commit; }

static inline const hobbit_F *hF(const F *p) { return reinterpret_cast<const hobbit_F *>(p); }
static inline hobbit_F *hF(F *p) { return reinterpret_cast<hobbit_F *>(p); }
static_assert(sizeof(F) == sizeof(hobbit_F), "fieldElement layout");

struct DevBuf {   // RAII device buffer
    void *p = nullptr;
    DevBuf(size_t bytes) { HCHK(hobbit_malloc(hobbit_host_ctx(), bytes, &p)); }
    DevBuf(const void *h, size_t bytes) : DevBuf(bytes) { if (bytes) HCHK(hobbit_memcpy_h2d(g_ctx, p, h, bytes)); }
    ~DevBuf() { if (p) hobbit_free(g_ctx, p); }
    void to_host(void *h, size_t bytes, size_t off = 0) { if (bytes) HCHK(hobbit_memcpy_d2h(g_ctx, h, (char *)p + off, bytes)); }
};

// ---- field (src/fieldElement.cpp:24-96, 206-209) ---------------------------------------------------
namespace virgo {
static const unsigned long long MODP = 2305843009213693951ULL;
fieldElement::fieldElement(long long x) { real = x >= 0 ? x : MODP + x; img = 0; }
fieldElement::fieldElement(long long x, long long y) { real = x >= 0 ? x : MODP + x; img = y >= 0 ? y : MODP + y; }
fieldElement fieldElement::operator+(const fieldElement &o) const { fieldElement r; r.real = real + o.real; if (r.real >= MODP) r.real -= MODP; r.img = img + o.img; if (r.img >= MODP) r.img -= MODP; return r; }
fieldElement fieldElement::operator-(const fieldElement &o) const { fieldElement r; r.real = real >= o.real ? real - o.real : real + MODP - o.real; r.img = img >= o.img ? img - o.img : img + MODP - o.img; return r; }
fieldElement fieldElement::operator-() const { return zero() - *this; }
fieldElement fieldElement::operator*(const fieldElement &o) const { fieldElement r; hobbit_f_mul_host(hF(this), hF(&o), hF(&r), 1); return r; }
fieldElement fieldElement::inv() const { fieldElement r; hobbit_f_inv_host(hF(this), hF(&r), 1); return r; }
}  // namespace virgo

// ---- transcript -----------------------------------------------------------------------------------
void init_hash() {}                                                 // constants are compiled in (src/mimc.cpp:11-19)
F mimc_hash(F input, F k) { F r; hobbit_mimc(hF(&input), hF(&k), hF(&r)); return r; }

// ---- utils ----------------------------------------------------------------------------------------
vector<F> generate_randomness(int size) {                           // src/utils.cpp:873-883 (host, libc)
    vector<F> x; x.reserve(size);
    F c;
    for (int i = 0; i < size; i++) { if (i % 100 == 0) c = F(random()); x.push_back(c + F(rand())); }
    return x;
}
void precompute_beta(vector<F> r, vector<F> &B) {                   // src/utils.cpp:251-296
    B.resize((size_t)1 << r.size());
    DevBuf d(B.size() * sizeof(F));
    HCHK(hobbit_eq_table(hobbit_host_ctx(), hF(r.data()), (int)r.size(), (hobbit_F *)d.p));
    d.to_host(B.data(), B.size() * sizeof(F));
}
F evaluate_vector(vector<F> v, vector<F> r) {                       // src/utils.cpp:789-802
    DevBuf d(v.data(), v.size() * sizeof(F));
    F out;
    HCHK(hobbit_eval_vector(hobbit_host_ctx(), (const hobbit_F *)d.p, v.size(), hF(r.data()), hF(&out)));
    return out;
}
void _fft(F *arr, int logn, bool flag) {                            // src/utils.cpp:605-673
    // NOTE: the reference caches twiddles by length only, so an inverse call right after a forward call
    // of the same length reuses FORWARD twiddles.  This mirror always uses the twiddles of
    // the requested direction; the commit path only ever calls the forward transform.
    size_t len = (size_t)1 << logn;
    if (logn > 12) { printf("Error: _fft length 2^%d not supported on the device yet\n", logn); exit(-1); }
    DevBuf d(arr, len * sizeof(F));
    HCHK(hobbit_fft_batch(hobbit_host_ctx(), (hobbit_F *)d.p, logn, 1, len, flag ? 1 : 0));
    d.to_host(arr, len * sizeof(F));
}
void fft(vector<F> &arr, int logn, bool flag) { _fft(arr.data(), logn, flag); }   // src/utils.cpp:467-527

// ---- expander code --------------------------------------------------------------------------------
static graph generate_random_expander_store(long long L, long long R, long long d) {   // src/expanders.h:20-47
    graph ret; ret.degree = (int)d;
    ret.neighbor.resize(L); ret.weight.resize(L); ret.r_neighbor.resize(R); ret.r_weight.resize(R);
    for (long long i = 0; i < L; ++i) {
        ret.neighbor[i].resize(d); ret.weight[i].resize(d);
        for (long long j = 0; j < d; ++j) {
            long long target = rand() % R;
            F weight = random();
            ret.neighbor[i][j] = target; ret.weight[i][j] = weight;
            ret.r_neighbor[target].push_back(i); ret.r_weight[target].push_back(weight);
        }
    }
    ret.L = L; ret.R = R;
    return ret;
}
static void upload_graph(int dep, int kind, const graph &g) {
    vector<long long> nbr((size_t)g.L * g.degree); vector<F> w((size_t)g.L * g.degree);
    for (long long i = 0; i < g.L; i++) for (int j = 0; j < g.degree; j++) { nbr[i * g.degree + j] = g.neighbor[i][j]; w[i * g.degree + j] = g.weight[i][j]; }
    HCHK(hobbit_graph_upload(hobbit_host_ctx(), dep, kind, g.L, g.R, g.degree, nbr.data(), hF(w.data())));
}
static const double k_alpha = 0.211, k_r = 1.72;                    // src/parameter.h:4-8
static const int k_cn = 9, k_dn = 12, k_distance_threshold = (int)(1.0 / 0.07) - 1;
static long long expander_rec(long long n, int dep) {               // src/expanders.h:78-92
    if (n <= k_distance_threshold) return n;
    _C[dep] = generate_random_expander_store(n, (long long)(k_alpha * n), k_cn);
    long long L = expander_rec((long long)(k_alpha * n), dep + 1);
    D[dep] = generate_random_expander_store(L, (long long)(n * (k_r - 1) - L), k_dn);
    upload_graph(dep, 0, _C[dep]); upload_graph(dep, 1, D[dep]);
    return n + L + (long long)(n * (k_r - 1) - L);
}
// Upload whatever graphs the host arrays _C / D hold for a code of length n (walk of src/expanders.h:78-92 without drawing): for a build in
// which a copy of the reference's own expander_init_store (inline in src/expanders.h) drew them and nothing told the device.
static long long upload_rec(long long n, int dep) {
    if (n <= k_distance_threshold) return n;
    if (_C[dep].L != n) { printf("Error: no expander graph of %lld left nodes at depth %d in _C (expander_init_store has not run for this size)\n", n, dep); exit(-1); }
    long long L = upload_rec((long long)(k_alpha * n), dep + 1);
    upload_graph(dep, 0, _C[dep]); upload_graph(dep, 1, D[dep]);
    return n + L + (long long)(n * (k_r - 1) - L);
}
void hobbit_host_upload_graphs(long long n) {
    HCHK(hobbit_graph_reset(hobbit_host_ctx()));
    long long total = upload_rec(n, 0), len = 0;
    HCHK(hobbit_graph_finalize(g_ctx, n, &len));
    if (len != total) { printf("Error in hobbit_host_upload_graphs\n"); exit(-1); }
}
long long expander_init_store(long long n, int dep) {
    if (dep == 0) HCHK(hobbit_graph_reset(hobbit_host_ctx()));
    long long total = expander_rec(n, dep);
    if (dep == 0) { long long len = 0; HCHK(hobbit_graph_finalize(g_ctx, n, &len)); if (len != total) { printf("Error in expander_init_store\n"); exit(-1); } }
    return total;
}
int encode_monolithic(const F *src, F *dst, long long n, int dep) {  // src/linear_code_encode.h:62-119
    if (dep != 0) { printf("Error: encode_monolithic recursion is internal to the device kernel\n"); exit(-1); }
    DevBuf s(src, (size_t)n * sizeof(F)), d((size_t)2 * n * sizeof(F));
    long long len = n;
    HCHK(hobbit_graph_finalize(hobbit_host_ctx(), n, &len));         // (re)binds the uploaded levels to this n; returns n+L+R
    HCHK(hobbit_encode_batch(g_ctx, (const hobbit_F *)s.p, (hobbit_F *)d.p, n, 1, (size_t)n, (size_t)2 * n));
    d.to_host(dst, (size_t)len * sizeof(F));                        // the reference writes n+L+R elements of dst
    return (int)len;
}

// ---- hashes / Merkle ------------------------------------------------------------------------------
void blake3_hash(uint8_t *src, uint8_t *dst) {                      // src/Blake3_hash.cpp:5-10
    DevBuf s(src, 64), d(32);
    HCHK(hobbit_blake3_64(hobbit_host_ctx(), (const uint8_t *)s.p, (uint8_t *)d.p, 1));
    d.to_host(dst, 32);
}
namespace merkle_tree {
_hash hash_double_field_element_merkle_damgard_blake(virgo::fieldElement x, virgo::fieldElement y, virgo::fieldElement z, virgo::fieldElement w, _hash &prev_hash) {
    F e[4] = {x, y, z, w};                                          // src/merkle_tree.cpp:62-87
    DevBuf s(e, 64), p(prev_hash.arr, 32);
    HCHK(hobbit_hash_md(hobbit_host_ctx(), (const hobbit_F *)s.p, (const uint8_t *)p.p, (uint8_t *)p.p, 1));
    _hash r; p.to_host(r.arr, 32); return r;
}
namespace merkle_tree_prover {
static void unflatten(DevBuf &lv, size_t n, vector<vector<_hash>> &hashes) {
    size_t levels = (size_t)log2((double)n) + 1;
    if (hashes.size() < levels) hashes.resize(levels);
    size_t off = 0;
    for (size_t l = 0, sz = n; l < levels; l++, sz /= 2) { hashes[l].resize(sz); lv.to_host(hashes[l].data(), 32 * sz, 32 * off); off += sz; }
}
void MT_commit_Blake(F *leafs, vector<vector<_hash>> &hashes, int N) {   // src/merkle_tree.cpp:193-221
    DevBuf s(leafs, (size_t)N * sizeof(F)), lv((size_t)64 * (N / 4));
    HCHK(hobbit_mt_commit_blake(hobbit_host_ctx(), (const hobbit_F *)s.p, (size_t)N, (uint8_t *)lv.p));
    unflatten(lv, (size_t)N / 4, hashes);
}
void create_tree_blake(int ele_num, vector<vector<_hash>> &hashes, const int, bool) {   // src/merkle_tree.cpp:255-287 (left|left kept)
    DevBuf lv((size_t)64 * ele_num);
    HCHK(hobbit_memcpy_h2d(hobbit_host_ctx(), lv.p, hashes[0].data(), (size_t)32 * ele_num));
    HCHK(hobbit_merkle_levels(g_ctx, (uint8_t *)lv.p, (size_t)ele_num, 1));
    unflatten(lv, (size_t)ele_num, hashes);
}
vector<_hash> open_tree_blake(vector<vector<_hash>> &MT_hashes, vector<size_t> c, int collumns) {   // src/merkle_tree.cpp:308-324
    int pos = (int)((c[1] / 4) * collumns + c[0]);
    if ((size_t)pos >= MT_hashes[0].size()) { printf("Error %d,%d\n", pos, (int)MT_hashes[0].size()); exit(-1); }
    vector<_hash> path;
    for (size_t i = 0; i + 1 < MT_hashes.size(); i++) { path.push_back(MT_hashes[i][2 * (pos / 2) + (1 - (pos % 2))]); pos = pos / 2; }
    return path;
}
}  // namespace merkle_tree_prover
}  // namespace merkle_tree

// ---- tensor code / commit -------------------------------------------------------------------------
void compute_tensorcode(vector<F> &message, vector<vector<F>> &tensor) {       // src/PC_utils.cpp:66-123
    size_t M = message.size(), cols = 2 * M / tensor_row_size, rows2 = 2 * (size_t)tensor_row_size;
    DevBuf m(message.data(), M * sizeof(F)), t(cols * rows2 * sizeof(F));
    HCHK(hobbit_tensorcode(hobbit_host_ctx(), (const hobbit_F *)m.p, M, tensor_row_size, linear_time ? 1 : 0, (hobbit_F *)t.p));
    vector<F> cm(cols * rows2);                                     // codeword-major on the device
    t.to_host(cm.data(), cm.size() * sizeof(F));
    tensor.resize(rows2);
    for (size_t r = 0; r < rows2; r++) { tensor[r].resize(cols); for (size_t c = 0; c < cols; c++) tensor[r][c] = cm[c * rows2 + r]; }
}
void hobbit_host_materialize_tensor(vector<vector<vector<F>>> &_tensor) {
    if (!g_commit) return;
    _tensor.resize(g_commit_K);
    for (int i = 0; i < g_commit_K; i++) {
        _tensor[i].resize(2 * (size_t)g_commit_trs);
        for (int r = 0; r < 2 * g_commit_trs; r++) { _tensor[i][r].resize(g_commit_cols); HCHK(hobbit_commitment_tensor_row(g_ctx, g_commit, i, r, hF(_tensor[i][r].data()))); }
    }
}
void commit_standard(vector<F> &poly, _hash &comm, vector<vector<_hash>> &MT_hashes, vector<vector<vector<F>>> &_tensor, int K) {   // src/Our_PC.cpp:146-171
    (void)comm;                                                     // the reference never writes it either
    size_t N = poly.size(), M = N / K;
    const double t_in = now_s();
    auto lap = [&](const char *what) { if (g_timing) printf("[hobbit] commit_standard: %-34s at %9.3f ms\n", what, 1e3 * (now_s() - t_in)); };
    // the previous commitment (with the device copy of its polynomial) moves to the one-deep stash; what was there is released
    hobbit_host_ctx();
    lap("context ready");
    free_prev();
    if (g_commit) { swap_with_prev(); g_commit = nullptr; g_poly_dev = nullptr; g_poly_n = 0; }
    // the device copy of poly is kept: open_standard receives the same vector and would otherwise pay the PCIe upload again
    HCHK(hobbit_malloc(hobbit_host_ctx(), N * sizeof(F), &g_poly_dev)); g_poly_n = N;
    if (getenv("HOBBIT_HOST_BLOCKING_UPLOAD")) {                    // the round-2 form, for A/B: one blocking copy of the whole vector, then the commit
        HCHK(hobbit_memcpy_h2d(g_ctx, g_poly_dev, poly.data(), N * sizeof(F)));
        HCHK(hobbit_commit_standard(g_ctx, (const hobbit_F *)g_poly_dev, N, K, tensor_row_size, linear_time ? 1 : 0, &g_commit));
    } else                                                          // chunk group g + 1 crosses PCIe while group g's row FFT / layout change run
        HCHK(hobbit_commit_standard_host(g_ctx, hF(poly.data()), (hobbit_F *)g_poly_dev, N, K, tensor_row_size, linear_time ? 1 : 0, &g_commit));
    HCHK(hobbit_sync(g_ctx));
    lap("upload + commit kernels done");
    g_commit_K = K; g_commit_trs = tensor_row_size; g_commit_cols = 2 * M / tensor_row_size;
    // MT_hashes (the reference's return value: every level of the tree, 64 B per leaf): each level straight into its own vector
    size_t levels = (size_t)log2((double)M) + 1, off = 0;
    MT_hashes.resize(levels);
    const uint8_t *d_lv = (const uint8_t *)hobbit_commitment_levels_dev(g_commit);
    for (size_t l = 0, sz = M; l < levels; l++, sz /= 2) { MT_hashes[l].resize(sz); HCHK(hobbit_memcpy_d2h(g_ctx, MT_hashes[l].data(), d_lv + 32 * off, 32 * sz)); off += sz; }
    memcpy(g_commit_root, MT_hashes.back()[0].arr, 32);
    lap("MT_hashes read back");
    _tensor.clear(); _tensor.resize(K);
    const char *mat = getenv("HOBBIT_MATERIALIZE_TENSOR");
    if ((mat && atoi(mat)) || (size_t)4 * N * sizeof(F) <= ((size_t)256 << 20)) hobbit_host_materialize_tensor(_tensor);
}
void prove_gate_consistency_standard(vector<F> &arr_L, vector<F> &arr_R, vector<F> &arr_O, vector<F> &add_gate, vector<F> r, double &vt, double &ps) {
    (void)vt; (void)ps;                                              // the reference does not touch them either (src/sumcheck.cpp:434-501)
    const size_t n = arr_L.size(); const int rounds = (int)log2((double)n);
    vector<F> mul_gate(n), beta;
    for (size_t i = 0; i < n; i++) mul_gate[i] = F(1) - add_gate[i];
    precompute_beta(r, beta);
    DevBuf dA(add_gate.data(), n * sizeof(F)), dB(beta.data(), n * sizeof(F)), dL(arr_L.data(), n * sizeof(F)), dR(arr_R.data(), n * sizeof(F)),
        dO(arr_O.data(), n * sizeof(F)), dM(mul_gate.data(), n * sizeof(F));
    F a[4] = {F(1), F(1), F(1), F(0) - F(1)}, rnd = F(213), sum = F(0), fin[6];
    vector<F> poly(5 * (size_t)rounds), rr(rounds);
    int ok = 0;
    HCHK(hobbit_gate_sumcheck(hobbit_host_ctx(), (const hobbit_F *)dA.p, (const hobbit_F *)dB.p, (const hobbit_F *)dL.p, (const hobbit_F *)dR.p, (const hobbit_F *)dO.p,
                              (const hobbit_F *)dM.p, n, hF(a), hF(&rnd), hF(&sum), hF(poly.data()), hF(rr.data()), hF(fin), &ok));
    if (!ok) printf("Error in gate consistency 2\n");               // the reference prints and continues (:482-485)
    add_gate[0] = fin[0]; arr_L[0] = fin[2]; arr_R[0] = fin[3]; arr_O[0] = fin[4];
}
// proof-size accounting of verify_claim_opt_blake (src/merkle_tree.cpp:326-361): 32 B per sibling not yet seen
static void path_ps(size_t n_leaves, int depth, const vector<size_t> &pos, double &ps) {
    vector<bool> visited(2 * n_leaves + 2, false);
    for (size_t p : pos) {
        size_t pe = n_leaves + p;
        for (int i = 0; i < depth; i++) {
            if (visited[pe ^ 1]) break;
            visited[pe ^ 1] = true; pe /= 2; visited[pe] = true;
            ps += 32.0 / 1024.0;
        }
    }
}
static void sumcheck2_ps(int rounds, double &ps) { for (int i = 0; i < rounds; i++) ps += 3 * sizeof(F) / 1024.0; ps += 2 * sizeof(F) / 1024.0; }   // src/sumcheck.cpp:2431,2449
struct SpBuffers {   // host buffers behind one hobbit_shockwave_out
    hobbit_host_shockwave_transcript &t; hobbit_shockwave_out o;
    SpBuffers(hobbit_host_shockwave_transcript &tt, size_t N, int k) : t(tt) {
        size_t w = N / k, W = 2 * w; int lgW = (int)log2((double)W), lw = (int)log2((double)w);
        t.I.assign(240, 0); t.q1.assign(3 * lgW, F(0)); t.r1.assign(lgW, F(0)); t.vr1.assign(2, F(0)); t.q2.assign(3 * lgW, F(0)); t.r2.assign(lgW, F(0)); t.vr2.assign(2, F(0));
        t.wq.assign(3 * (lw + 8), F(0)); t.wa.assign(lw + 8, F(0)); t.wroots.assign(32 * (lw + 1), 0); t.wscal.assign(2, F(0));
        t.reply.assign(240 * (size_t)k, F(0)); t.paths.assign(240 * (size_t)lgW * 32, 0);
        t.wqidx.assign(256, 0); t.wqreply.assign(256 * 16, F(0)); t.wqpaths.assign(256 * 24 * 32, 0); t.wfinal.assign(32, F(0)); t.wqn.assign(8, 0);
        o = hobbit_shockwave_out{t.I.data(), hF(t.q1.data()), hF(t.r1.data()), hF(t.vr1.data()), hF(&t.fin1), hF(t.q2.data()), hF(t.r2.data()), hF(t.vr2.data()), hF(&t.fin2),
                                 hF(t.wq.data()), hF(t.wa.data()), t.wroots.data(), hF(t.wscal.data()), t.wchecks, t.whir_root, &t.iters,
                                 hF(t.reply.data()), t.paths.data(), t.wqidx.data(), hF(t.wqreply.data()), t.wqpaths.data(), hF(t.wfinal.data()), t.wqn.data()};
    }
};
// ps of one shockwave_prove (src/Virgo.cpp:435-517) in the reference's order of accumulation
static void shockwave_ps(const hobbit_host_shockwave_transcript &t, size_t N, int k, double &ps) {
    size_t w = N / k, W = 2 * w; int lgW = (int)log2((double)W);
    sumcheck2_ps(lgW, ps); sumcheck2_ps(lgW, ps);                                         // P1, prove_fft
    // :479 tests aggr.size()/2 after prove_fft doubled aggr in place (src/sumcheck.cpp:2984-2985): the original width, and :482 adds the doubled one
    if (w > 256) {                                                                        // _whir_prove (:519-686)
        size_t q = 0;
        for (int it = 1; it <= t.iters; it++) {
            for (int i = 0; i < 4; i++) ps += (3 * sizeof(F)) / 1024.0;
            size_t size = it == 1 ? 2 * w : (2 * w) >> (it - 1);                          // layer the queries of this round read
            auto answer = [&](int round) {
                int n = t.wqn[round];
                ps += 16.0 * n * sizeof(F) / 1024.0;
                vector<size_t> pos(t.wqidx.begin() + q, t.wqidx.begin() + q + n);
                path_ps(size / 4, (int)log2((double)(size / 4)), pos, ps);
                q += n;
            };
            if (it < t.iters) answer(it - 1);
            else { ps += (w >> (4 * t.iters)) * 2 * sizeof(F) / 1024.0; answer(it - 1); }
        }
    } else ps += 2 * w * sizeof(F) / 1024.0;
    ps += 240.0 * k * sizeof(F) / 1024.0;
    vector<size_t> pos(t.I.begin(), t.I.end());
    path_ps(W, lgW, pos, ps);
}
// src/Our_PC.cpp:604-692.  The prover side runs on the device (hobbit_open_standard: libc draws in the reference's order);
// the reference's verifier emulation is reduced to its proof-size accounting (ps); vt is the time that accounting took here.
static hobbit_host_elastic_transcript g_eopen;
hobbit_host_elastic_transcript &hobbit_host_last_elastic_open() { return g_eopen; }
// open_standard's linear_time == false branch (src/Our_PC.cpp:609-611, 651-652: 790 queries, recursive_prover_RS); the messages land in
// hobbit_host_last_elastic_open() (same transcript shape as Elastic_PC::open option 1, replies queries x K)
static void open_standard_rs(vector<F> &poly, vector<F> &x, vector<vector<_hash>> &Commitment_MT, int K, double &vt, double &ps) {
    const size_t N = poly.size(), M = N / K;
    const int queries = 790; aggregation_queries = queries;
    const int trs = g_commit_trs; const size_t cols = g_commit_cols;
    const int logc = (int)log2((double)cols), logr = (int)log2((double)(2 * trs)), logt = logr - 1, depth = (int)log2((double)M);
    if (!g_poly_dev || g_poly_n != N || getenv("HOBBIT_HOST_REUPLOAD")) {
        if (g_poly_dev) hobbit_free(hobbit_host_ctx(), g_poly_dev);
        HCHK(hobbit_malloc(hobbit_host_ctx(), N * sizeof(F), &g_poly_dev)); g_poly_n = N;
        HCHK(hobbit_memcpy_h2d(g_ctx, g_poly_dev, poly.data(), N * sizeof(F)));
    }
    hobbit_host_elastic_transcript &t = g_eopen;
    const int maxr = 11 + logr + logr + (logt + logc) + logc;
    t.queries = queries; t.cols.assign(queries, 0); t.rows.assign(queries, 0); t.reply.assign((size_t)queries * K, F(0)); t.paths.assign((size_t)queries * depth * 32, 0);
    t.qpoly.assign(3 * (size_t)maxr, F(0)); t.r.assign(maxr, F(0)); t.vr.assign(8, F(0)); t.fin.assign(4, F(0)); t.rx.assign(logc + logt, F(0));
    SpBuffers bf(t.sp_f, M, 32);
    hobbit_elastic_open_out o{t.cols.data(), t.rows.data(), hF(&t.rv0), hF(t.reply.data()), &t.reply_len, t.paths.data(), t.cf_root, &t.ncols,
                              hF(t.qpoly.data()), hF(t.r.data()), hF(t.vr.data()), hF(t.fin.data()), t.checks, hF(t.rx.data()), &bf.o};
    HCHK(hobbit_open_standard_rs(g_ctx, (const hobbit_F *)g_poly_dev, N, g_commit, hF(x.data()), queries, &o));
    if (!t.checks[0] || !t.checks[1]) { printf("Error in fft\n"); exit(-1); }                                   // src/sumcheck.cpp:3016-3019
    if (t.sp_f.iters && !(t.sp_f.wchecks[0] && t.sp_f.wchecks[1])) { printf("Error in final verification step\n"); exit(-1); }
    printf(">>OK\n");
    auto t0 = std::chrono::steady_clock::now();
    size_t np2 = 1; while (np2 < (size_t)t.ncols) np2 <<= 1;
    t.rounds = (int)log2((double)(np2 * 2 * trs)) + logr + (logt + logc) + logc;
    ps += (double)((size_t)queries * K * sizeof(F)) / 1024.0;                                                   // (:648)
    printf(">> %lf Kb\n", (double)((size_t)queries * K * sizeof(F)) / 1024.0);
    sumcheck2_ps((int)log2((double)(np2 * 2 * trs)), ps); sumcheck2_ps(logr, ps); sumcheck2_ps(logt + logc, ps); sumcheck2_ps(logc, ps);   // P0, P2, P3, P5
    shockwave_ps(t.sp_f, M, 32, ps);
    double MT_ps = 0.0;
    vector<size_t> pos(queries);
    for (int i = 0; i < queries; i++) pos[i] = (size_t)(t.rows[i] / 4) * cols + t.cols[i];
    path_ps(Commitment_MT.empty() ? M : Commitment_MT[0].size(), depth, pos, MT_ps);                            // (:676-679)
    printf("Opening proofs : %lf\n", MT_ps);
    ps += MT_ps;
    vt += std::chrono::duration_cast<std::chrono::duration<double>>(std::chrono::steady_clock::now() - t0).count();
}
void open_standard(vector<F> &poly, vector<F> x, vector<vector<_hash>> &Commitment_MT, vector<vector<vector<F>>> &_tensor, int K, double &vt, double &ps) {
    (void)_tensor;
    select_commitment(Commitment_MT);
    if (!g_commit || g_commit_K != K) { printf("Error: open_standard without a matching commit_standard\n"); exit(-1); }
    const size_t N = poly.size(), M = N / K;
    BUFFER_SPACE = M;
    if (!linear_time) { open_standard_rs(poly, x, Commitment_MT, K, vt, ps); return; }
    const int queries = 5900; aggregation_queries = queries;
    const int trs = g_commit_trs; const size_t cols = g_commit_cols;
    const int R1 = (int)log2((double)(2 * trs)), logc = (int)log2((double)cols), R3 = R1 + logc, depth = (int)log2((double)M);
    if (!g_poly_dev || g_poly_n != N || getenv("HOBBIT_HOST_REUPLOAD")) {
        if (g_poly_dev) hobbit_free(hobbit_host_ctx(), g_poly_dev);
        HCHK(hobbit_malloc(hobbit_host_ctx(), N * sizeof(F), &g_poly_dev)); g_poly_n = N;
        HCHK(hobbit_memcpy_h2d(g_ctx, g_poly_dev, poly.data(), N * sizeof(F)));
    }
    hobbit_host_open_transcript &t = g_open;
    t.queries = queries; t.rounds = R1 + logc + 2 * R3 + logc;
    t.cols.assign(queries, 0); t.rows.assign(queries, 0); t.reply.assign((size_t)queries * K, F(0)); t.paths.assign((size_t)queries * depth * 32, 0);
    t.qpoly.assign(3 * (size_t)t.rounds, F(0)); t.r.assign(t.rounds, F(0)); t.vr.assign(10, F(0)); t.fin.assign(5, F(0)); t.scalars.assign(5, F(0));
    SpBuffers bc(t.sp_c, (size_t)trs * cols, 32), bf(t.sp_f, M, 32);
    hobbit_open_out o{t.cols.data(), t.rows.data(), hF(t.reply.data()), t.paths.data(), hF(t.qpoly.data()), hF(t.r.data()), hF(t.vr.data()), hF(t.fin.data()),
                      hF(t.scalars.data()), t.checks, t.roots, &bc.o, &bf.o};
    HCHK(hobbit_open_standard(g_ctx, (const hobbit_F *)g_poly_dev, N, g_commit, hF(x.data()), queries, &o));
    if (!t.checks[0]) { printf("Error recursion 1\n"); exit(-1); }                         // src/PC_utils.cpp:323-326
    if (!t.checks[1]) { printf("Error recursion 2\n"); exit(-1); }                         // :364-367
    if (!t.checks[2]) { printf("Error in fft\n"); exit(-1); }                              // src/sumcheck.cpp:3016-3019
    for (auto *sp : {&t.sp_c, &t.sp_f}) if (sp->iters && !(sp->wchecks[0] && sp->wchecks[1])) { printf("Error in final verification step\n"); exit(-1); }   // src/Virgo.cpp:562-565, 648-651
    printf(">>OK\n");
    auto t0 = std::chrono::steady_clock::now();
    ps += (double)((size_t)queries * K * sizeof(F)) / 1024.0;                             // :653
    printf(">> %lf Kb\n", (double)((size_t)queries * K * sizeof(F)) / 1024.0);
    sumcheck2_ps(R1, ps); sumcheck2_ps(logc, ps); sumcheck2_ps(R3, ps); sumcheck2_ps(R3, ps);   // P1..P4
    shockwave_ps(t.sp_c, (size_t)trs * cols, 32, ps);
    sumcheck2_ps(logc, ps);                                                                // P5
    shockwave_ps(t.sp_f, M, 32, ps);
    double MT_ps = 0.0;
    vector<size_t> pos(queries);
    for (int i = 0; i < queries; i++) pos[i] = (size_t)(t.rows[i] / 4) * cols + t.cols[i];
    path_ps(Commitment_MT.empty() ? M : Commitment_MT[0].size(), depth, pos, MT_ps);       // :676-679
    printf("Opening proofs : %lf\n", MT_ps);
    ps += MT_ps;
    vt += std::chrono::duration_cast<std::chrono::duration<double>>(std::chrono::steady_clock::now() - t0).count();
}
void _aggregate_axpy(vector<F> &poly, vector<F> beta1, vector<F> &aggregated_vector, int K) {    // src/Our_PC.cpp:258-272
    size_t M = poly.size() / K;
    DevBuf p(poly.data(), poly.size() * sizeof(F)), a(M * sizeof(F));
    HCHK(hobbit_aggregate(hobbit_host_ctx(), (const hobbit_F *)p.p, poly.size(), hF(beta1.data()), K, (hobbit_F *)a.p));
    aggregated_vector.resize(M);
    a.to_host(aggregated_vector.data(), M * sizeof(F));
}
void _compute_aggregation_reply(vector<vector<size_t>> &I, vector<vector<F>> &reply, vector<vector<vector<F>>> &_tensor, int K) {   // src/Our_PC.cpp:291-305
    reply.resize(aggregation_queries);
    for (auto &r : reply) r.resize(K);
    if (!_tensor.empty() && !_tensor[0].empty()) {                  // materialised: the reference's own loop
        for (int i = 0; i < K; i++) for (size_t k = 0; k < I.size(); k++) reply[k][i] = _tensor[i][I[k][1]][I[k][0]];
        return;
    }
    vector<uint32_t> rows(I.size()), cols(I.size());
    for (size_t k = 0; k < I.size(); k++) { cols[k] = (uint32_t)I[k][0]; rows[k] = (uint32_t)I[k][1]; }
    vector<F> flat(I.size() * K);
    HCHK(hobbit_commitment_gather(hobbit_host_ctx(), g_commit, rows.data(), cols.data(), I.size(), hF(flat.data())));
    for (size_t k = 0; k < I.size(); k++) for (int i = 0; i < K; i++) reply[k][i] = flat[k * K + i];
}

// ---- sumchecks ------------------------------------------------------------------------------------
struct proof generate_2product_sumcheck_proof(vector<F> &_v1, vector<F> &_v2, F previous_r, double &vt, double &ps) {   // src/sumcheck.cpp:2391-2460
    size_t n = _v1.size();
    int rounds = (int)log2((double)n);
    DevBuf a(_v1.data(), n * sizeof(F)), b(_v2.data(), n * sizeof(F));
    vector<F> q(3 * (size_t)rounds), r(rounds); F vr[2], fin;
    HCHK(hobbit_sumcheck2(hobbit_host_ctx(), (const hobbit_F *)a.p, (const hobbit_F *)b.p, n, hF(&previous_r), hF(q.data()), hF(r.data()), hF(vr), hF(&fin)));
    struct proof Pr;
    for (int i = 0; i < rounds; i++) Pr.q_poly.push_back(quadratic_poly(q[3 * i], q[3 * i + 1], q[3 * i + 2]));
    Pr.randomness.push_back(r); Pr.vr = {vr[0], vr[1]}; Pr.final_rand = fin;
    ps += rounds * 3 * sizeof(F) / 1024.0 + 2 * sizeof(F) / 1024.0;       // same proof-size accounting (src/sumcheck.cpp:2431,2449)
    (void)vt;
    return Pr;
}
struct proof _generate_3product_sumcheck_proof(vector<F> &v1, vector<F> &v2, vector<F> &v3, F previous_r, double &vt, double &ps) {   // src/sumcheck.cpp:1974-2058
    size_t n = v1.size();
    int rounds = (int)log2((double)n);
    DevBuf a(v1.data(), n * sizeof(F)), b(v2.data(), n * sizeof(F)), c(v3.data(), n * sizeof(F));
    vector<F> q(4 * (size_t)rounds), r(rounds); F vr[3], fin;
    HCHK(hobbit_sumcheck3(hobbit_host_ctx(), (const hobbit_F *)a.p, (const hobbit_F *)b.p, (const hobbit_F *)c.p, n, hF(&previous_r), hF(q.data()), hF(r.data()), hF(vr), hF(&fin)));
    struct proof Pr;
    for (int i = 0; i < rounds; i++) Pr.c_poly.push_back(cubic_poly(q[4 * i], q[4 * i + 1], q[4 * i + 2], q[4 * i + 3]));
    Pr.randomness.push_back(r); Pr.vr = {vr[0], vr[1], vr[2]}; Pr.final_rand = fin;
    // the reference folds v1..v3 in place; callers only rely on element 0 afterwards
    v1[0] = vr[0]; v2[0] = vr[1]; v3[0] = vr[2];
    ps += rounds * 5 * sizeof(F) / 1024.0 + 3 * sizeof(F) / 1024.0;       // src/sumcheck.cpp:2029,2047
    (void)vt;
    return Pr;
}

// ---- code-membership / FFT-as-sumcheck ---------------------------------------------------------------
static struct proof proof_from2(int rounds, const vector<F> &q, const vector<F> &r, const F vr[2], const F &fin) {
    struct proof Pr;
    for (int i = 0; i < rounds; i++) Pr.q_poly.push_back(quadratic_poly(q[3 * i], q[3 * i + 1], q[3 * i + 2]));
    Pr.randomness.push_back(r); Pr.vr = {vr[0], vr[1]}; Pr.final_rand = fin;
    return Pr;
}
int evaluate_parity_matrix(vector<F> &A, vector<F> &beta1, int Offset, int n, int dep, int &lvl) {   // src/sumcheck.cpp:2888-2929
    if (Offset != 0 || dep != 0 || lvl != 0) { printf("Error: evaluate_parity_matrix recursion is internal to the device path\n"); exit(-1); }
    DevBuf b(beta1.data(), beta1.size() * sizeof(F)), a(A.size() * sizeof(F));
    HCHK(hobbit_parity_matrix(hobbit_host_ctx(), (const hobbit_F *)b.p, A.size(), n, (hobbit_F *)a.p));
    vector<F> t(A.size()); a.to_host(t.data(), t.size() * sizeof(F));
    for (size_t i = 0; i < A.size(); i++) A[i] += t[i];          // the reference accumulates into A
    long long len = 0; HCHK(hobbit_graph_finalize(g_ctx, n, &len));
    return (int)len;
}
proof prove_linear_code(vector<F> &codeword, int n, double &vt, double &ps) {       // src/sumcheck.cpp:3223-3235
    int k = (int)log2((double)codeword.size());
    vector<F> r1 = generate_randomness(k);
    DevBuf cw(codeword.data(), codeword.size() * sizeof(F));
    vector<F> q(3 * (size_t)k), r(k); F vr[2], fin;
    HCHK(hobbit_prove_linear_code(hobbit_host_ctx(), (const hobbit_F *)cw.p, codeword.size(), n, hF(r1.data()), hF(q.data()), hF(r.data()), hF(vr), hF(&fin)));
    proof P = proof_from2(k, q, r, vr, fin);
    ps += k * 3 * sizeof(F) / 1024.0 + 2 * sizeof(F) / 1024.0; (void)vt;
    if (P.q_poly[0].eval(0) + P.q_poly[0].eval(1) != F(0)) printf("Error in codeword\n");
    P.randomness.push_back(r1);
    return P;
}
struct proof prove_fft(vector<F> &m, vector<F> r, F previous_sum, double &vt, double &ps) {   // src/sumcheck.cpp:2975-2987
    size_t s = m.size(); int k = (int)r.size();
    DevBuf dm(m.data(), s * sizeof(F));
    m.resize(2 * s, F(0));                                        // the reference pads its argument in place
    vector<F> q(3 * (size_t)k), rr(k); F vr[2], fin;
    HCHK(hobbit_prove_fft(hobbit_host_ctx(), (const hobbit_F *)dm.p, s, hF(r.data()), hF(q.data()), hF(rr.data()), hF(vr), hF(&fin)));
    struct proof Pr = proof_from2(k, q, rr, vr, fin);
    ps += k * 3 * sizeof(F) / 1024.0 + 2 * sizeof(F) / 1024.0; (void)vt;
    if (previous_sum != (Pr.q_poly[0].eval(0) + Pr.q_poly[0].eval(1))) printf("Error in fft\n");
    Pr.randomness[0].pop_back();
    return Pr;
}
struct proof prove_fft_matrix(vector<vector<F>> M, vector<F> r, F previous_sum, double &vt, double &ps) {   // src/sumcheck.cpp:2989-3027
    size_t rows = M.size(), cols = M[0].size();
    vector<F> flat(rows * cols);
    for (size_t i = 0; i < rows; i++) memcpy((void *)(flat.data() + i * cols), M[i].data(), cols * sizeof(F));
    DevBuf dM(flat.data(), flat.size() * sizeof(F));
    int k2 = (int)log2((double)(2 * cols)), k1 = (int)log2((double)rows);
    vector<F> q(3 * (size_t)k2), rr(k2); F vr[2], fin;
    HCHK(hobbit_prove_fft_matrix(hobbit_host_ctx(), (const hobbit_F *)dM.p, rows, cols, hF(r.data()), hF(q.data()), hF(rr.data()), hF(vr), hF(&fin)));
    struct proof Pr = proof_from2(k2, q, rr, vr, fin);
    ps += k2 * 3 * sizeof(F) / 1024.0 + 2 * sizeof(F) / 1024.0; (void)vt;
    if (previous_sum != (Pr.q_poly[0].eval(0) + Pr.q_poly[0].eval(1))) { printf("Error in fft\n"); exit(-1); }
    for (int i = 0; i < k1; i++) Pr.randomness[0].push_back(r[i + k2]);
    return Pr;
}
void phiGInit(vector<F> &phi_g, const vector<F>::const_iterator &rx, const F &scale, int n, bool isIFFT) {   // src/utils.cpp:694-755
    vector<F> r(rx, rx + n);
    DevBuf g(((size_t)1 << n) * sizeof(F));
    HCHK(hobbit_phi_g(hobbit_host_ctx(), hF(r.data()), n, hF(&scale), isIFFT ? 1 : 0, (hobbit_F *)g.p));
    if (phi_g.size() < ((size_t)1 << n)) phi_g.resize((size_t)1 << n);
    g.to_host(phi_g.data(), ((size_t)1 << n) * sizeof(F));
}
vector<vector<F>> transpose(vector<vector<F>> M) {                // src/utils.cpp (host helper)
    vector<vector<F>> T(M[0].size(), vector<F>(M.size()));
    for (size_t i = 0; i < M.size(); i++) for (size_t j = 0; j < M[0].size(); j++) T[j][i] = M[i][j];
    return T;
}
vector<F> prepare_matrix(vector<vector<F>> M, vector<F> r) {       // src/utils.cpp:758-775: V[i] = fold of row i
    size_t n = M.size(), c = M[0].size();
    vector<F> flat(n * c);                                         // device op folds the ROW index of columns: feed M^T
    for (size_t i = 0; i < n; i++) for (size_t j = 0; j < c; j++) flat[j * n + i] = M[i][j];
    DevBuf d(flat.data(), flat.size() * sizeof(F)), o(n * sizeof(F));
    HCHK(hobbit_prepare_matrix_cols(hobbit_host_ctx(), (const hobbit_F *)d.p, c, n, hF(r.data()), (int)r.size(), (hobbit_F *)o.p));
    vector<F> V(n); o.to_host(V.data(), n * sizeof(F));
    return V;
}

// ---- Elastic_PC streaming commit (src/Elastic_PC.cpp:174-285, 728-771) ------------------------------
// HOBBIT_HOST_REFERENCE_BUILD: the mirror is loaded in FRONT of the reference's own objects (tests/test_mlp_end_to_end.py: symbol
// interposition, nothing of the reference is recompiled or edited) -- the witness generator's stream readers are then the reference's,
// serving every stream, and the stand-alone readers below (synthetic default streams only) are left out.
#ifndef HOBBIT_HOST_REFERENCE_BUILD
void read_stream_PC(stream_descriptor &fd, F *v, int size) {       // src/witness_stream.cpp:2405-2411 (default branch)
    if (fd.name == "PC_layer" || fd.name == "witness" || fd.name == "circuit") { printf("Error: stream '%s' belongs to the witness generator (out of scope)\n", fd.name.c_str()); exit(-1); }
    F n = F(322322);
    for (int i = 0; i < size; i++) { v[i] = n; n = n * n + F(i); }
}
#endif
void init_commitment(bool mod) {                                    // src/Elastic_PC.cpp:728-734
    linear_time = mod;
    tensor_row_size = (int)(BUFFER_SPACE / (1ULL << 11));
    if (tensor_row_size == 0) tensor_row_size = 16;
}
static int pc_layer_chunk(stream_descriptor &raw, int layer, size_t B, void *d_out);
void commit(stream_descriptor fd, _hash &comm, vector<vector<_hash>> &MT_hashes) {   // src/Elastic_PC.cpp:174-285
    (void)comm; PhaseTimer pt__("commit");
    if (fd.size / BUFFER_SPACE < 4) printf("Decrease buffer size %d\n", (int)(fd.size / BUFFER_SPACE));
    hobbit_elastic *e = nullptr;
    HCHK(hobbit_elastic_begin(hobbit_host_ctx(), BUFFER_SPACE, tensor_row_size, linear_time ? 1 : 0, 1, &e));
    vector<F> buff(BUFFER_SPACE);
    DevBuf d(BUFFER_SPACE * sizeof(F));
    for (size_t i = 0; i < fd.size / BUFFER_SPACE; i++) {
        if (fd.name == "PC_layer") {
            // read_stream_PC's PC_layer branch (src/witness_stream.cpp:2357-2364): read_mul_tree_layer on the descriptor itself, whose
            // name read_stream does not know -> the product layer fd.layer of the DEFAULT stream; the products are taken on the device
            stream_descriptor raw = fd; raw.name = "test";
            if (pc_layer_chunk(raw, (int)fd.layer, BUFFER_SPACE, d.p) != 0) { printf("Error: PC_layer chunk\n"); exit(-1); }
        } else {
            TIMED_READ(read_stream_PC(fd, buff.data(), (int)BUFFER_SPACE));
            TIMED_UP(HCHK(hobbit_memcpy_h2d(g_ctx, d.p, buff.data(), BUFFER_SPACE * sizeof(F))));
        }
        HCHK(hobbit_elastic_push(g_ctx, e, (const hobbit_F *)d.p));
    }
    size_t T = 4 * BUFFER_SPACE;
    DevBuf lv(64 * T);
    HCHK(hobbit_elastic_finish(g_ctx, e, (uint8_t *)lv.p));
    hobbit_elastic_free(e);
    size_t levels = (size_t)log2((double)T) + 1, off = 0;
    MT_hashes.resize(levels);
    for (size_t l = 0, sz = T; l < levels; l++, sz /= 2) { MT_hashes[l].resize(sz); lv.to_host(MT_hashes[l].data(), 32 * sz, 32 * off); off += sz; }
}
#ifndef HOBBIT_HOST_REFERENCE_BUILD
void read_stream(stream_descriptor &fd, vector<F> &v, int size) {   // src/witness_stream.cpp:2106, default branch :2348-2352
    if (fd.name == "input" || fd.name == "circuit" || fd.name == "witness" || fd.name == "transcript_stream" || fd.name.find("lookup") == 0 ||
        fd.name.find("wiring_consistency_check") == 0) { printf("Error: stream '%s' belongs to the witness generator (out of scope)\n", fd.name.c_str()); exit(-1); }
    for (int i = 0; i < size; i++) v[i] = F((i % 1024) + 1);
}
#endif
// src/Elastic_PC.cpp:625-726 under linear_time (RS x expander; test_Elastic_PC option 2): 5900 queries, aggregate()'s expander branch,
// update_reply_spielman as the reference is built (include/hobbit_hip.h: hobbit_elastic_open_begin_lin), recursive_prover_Spielman_stream
static void open_linear_time(stream_descriptor fd, vector<F> x, vector<vector<_hash>> &Commitment_MT, double &vt, double &ps) {
    PhaseTimer pt__("open");
    const int queries = 5900; aggregation_queries = queries;
    const size_t B = BUFFER_SPACE, K = fd.size / B; const int trs = tensor_row_size;
    const size_t cols = 2 * B / (size_t)trs;
    const int logc = (int)log2((double)cols), R1 = (int)log2((double)(2 * trs)), logt = R1 - 1, depth = (int)log2((double)(4 * B));
    hobbit_elastic_open *e = nullptr;
    HCHK(hobbit_elastic_open_begin_lin(hobbit_host_ctx(), fd.size, B, trs, hF(x.data()), queries, &e));
    hobbit_host_elastic_transcript &t = g_eopen;
    t.lin = true;
    hobbit_elastic_open_dims(e, &t.ncols, &t.nrem, &t.np);
    vector<F> buff(B); DevBuf d(B * sizeof(F));
    stream_descriptor fd_a = fd, fd_r = fd;                          // aggregate and compute_aggregation_reply each take the descriptor BY VALUE (:316, :487)
    for (size_t i = 0; i < K; i++) {                                 // aggregate (:327-334)
        TIMED_READ(read_stream(fd_a, buff, (int)B));
        TIMED_UP(HCHK(hobbit_memcpy_h2d(g_ctx, d.p, buff.data(), B * sizeof(F))));
        HCHK(hobbit_elastic_open_aggregate_push(g_ctx, e, (const hobbit_F *)d.p));
    }
    HCHK(hobbit_elastic_open_aggregate_finish(g_ctx, e));
    printf("%d\n", (int)((size_t)t.nrem * 2 * (size_t)trs));         // aggregate() prints the flattened aux_commit's size (:408)
    for (size_t i = 0; i < K; i++) {                                 // compute_aggregation_reply (:506-531)
        TIMED_READ(read_stream(fd_r, buff, (int)B));
        TIMED_UP(HCHK(hobbit_memcpy_h2d(g_ctx, d.p, buff.data(), B * sizeof(F))));
        HCHK(hobbit_elastic_open_reply_push(g_ctx, e, (const hobbit_F *)d.p));
    }
    size_t tot = 0; for (auto &l : Commitment_MT) tot += l.size();
    DevBuf lv(32 * tot);
    { size_t off = 0; for (auto &l : Commitment_MT) { HCHK(hobbit_memcpy_h2d(g_ctx, (uint8_t *)lv.p + 32 * off, l.data(), 32 * l.size())); off += l.size(); } }
    const int R3 = (int)log2((double)t.np);
    t.rounds = R1 + logc + R3 + logc;
    t.queries = queries; t.cols.assign(queries, 0); t.rows.assign(queries, 0); t.reply.assign((size_t)queries * K, F(0)); t.paths.assign((size_t)queries * depth * 32, 0);
    t.qpoly.assign(3 * (size_t)t.rounds, F(0)); t.r.assign(t.rounds, F(0)); t.vr.assign(8, F(0)); t.fin.assign(4, F(0)); t.rx.assign(logc + logt - 1, F(0)); t.scal.assign(3, F(0));
    SpBuffers bf(t.sp_f, B, 32), bc(t.sp_c, t.np, 32);
    hobbit_elastic_open_out o{t.cols.data(), t.rows.data(), hF(&t.rv0), hF(t.reply.data()), &t.reply_len, t.paths.data(), t.cf_root, &t.ncols,
                              hF(t.qpoly.data()), hF(t.r.data()), hF(t.vr.data()), hF(t.fin.data()), t.checks, hF(t.rx.data()), &bf.o,
                              t.cc_root, &t.nrem, hF(t.scal.data()), &bc.o};
    HCHK(hobbit_elastic_open_finish(g_ctx, e, tot == 8 * B - 1 ? (const uint8_t *)lv.p : nullptr, &o));
    hobbit_elastic_open_free(e);
    if (!t.checks[0]) { printf("Error in fft\n"); exit(-1); }                                                   // src/sumcheck.cpp:3016-3019
    for (auto *sp : {&t.sp_c, &t.sp_f}) if (sp->iters && !(sp->wchecks[0] && sp->wchecks[1])) { printf("Error in final verification step\n"); exit(-1); }
    auto t0 = std::chrono::steady_clock::now();
    double MT_ps = 0.0;
    vector<size_t> pos(queries);
    for (int i = 0; i < queries; i++) pos[i] = (size_t)(t.rows[i] / 4) * cols + t.cols[i];
    path_ps(4 * B, depth, pos, MT_ps);                                                                          // (:684-688)
    ps += (double)((size_t)queries * (size_t)t.reply_len * sizeof(F)) / 1024.0;                                 // (:701)
    sumcheck2_ps(R1, ps); sumcheck2_ps(logc, ps); sumcheck2_ps(R3, ps);                                         // P1, P2, P3 (src/PC_utils.cpp:221-248)
    shockwave_ps(t.sp_c, t.np, 32, ps);                                                                         // (:252)
    sumcheck2_ps(logc, ps);                                                                                     // P5 (:266)
    shockwave_ps(t.sp_f, B, 32, ps);                                                                            // (:269)
    ps += MT_ps;
    vt += std::chrono::duration_cast<std::chrono::duration<double>>(std::chrono::steady_clock::now() - t0).count();
    for (auto &l : Commitment_MT) { l.clear(); vector<_hash>(l).swap(l); }                                      // (:691-696)
    Commitment_MT.clear();
    printf("PC : ps = %lf, vt = %lf\n", ps, vt);
}
// src/Elastic_PC.cpp:625-726, !linear_time (RS x RS).  Prover side on the device (hobbit_elastic_open_*: libc draws in the reference's
// order, the stream re-read twice through read_stream as the reference does); the verifier emulation is reduced to its ps accounting.
void open(stream_descriptor fd, vector<F> x, vector<vector<_hash>> &Commitment_MT, double &vt, double &ps) {
    if (linear_time) { open_linear_time(fd, x, Commitment_MT, vt, ps); return; }
    g_eopen.lin = false;
    PhaseTimer pt__("open");
    const int queries = 700; aggregation_queries = queries;
    const size_t B = BUFFER_SPACE, K = fd.size / B; const int trs = tensor_row_size;
    const int logc = 12, logr = (int)log2((double)(2 * trs)), logt = logr - 1, depth = (int)log2((double)(4 * B));
    hobbit_elastic_open *e = nullptr;
    HCHK(hobbit_elastic_open_begin(hobbit_host_ctx(), fd.size, B, trs, hF(x.data()), queries, &e));
    vector<F> buff(B); DevBuf d(B * sizeof(F));
    stream_descriptor fd_a = fd, fd_r = fd;                          // aggregate and compute_aggregation_reply each take the descriptor BY VALUE (:316, :487)
    for (size_t i = 0; i < K; i++) {                                 // aggregate (:327-334)
        TIMED_READ(read_stream(fd_a, buff, (int)B));
        TIMED_UP(HCHK(hobbit_memcpy_h2d(g_ctx, d.p, buff.data(), B * sizeof(F))));
        HCHK(hobbit_elastic_open_aggregate_push(g_ctx, e, (const hobbit_F *)d.p));
    }
    HCHK(hobbit_elastic_open_aggregate_finish(g_ctx, e));
    for (size_t i = 0; i < K; i++) {                                 // compute_aggregation_reply (:506-531)
        TIMED_READ(read_stream(fd_r, buff, (int)B));
        TIMED_UP(HCHK(hobbit_memcpy_h2d(g_ctx, d.p, buff.data(), B * sizeof(F))));
        HCHK(hobbit_elastic_open_reply_push(g_ctx, e, (const hobbit_F *)d.p));
    }
    // the commitment tree, flat, back on the device for the paths
    size_t tot = 0; for (auto &l : Commitment_MT) tot += l.size();
    DevBuf lv(32 * tot);
    { size_t off = 0; for (auto &l : Commitment_MT) { HCHK(hobbit_memcpy_h2d(g_ctx, (uint8_t *)lv.p + 32 * off, l.data(), 32 * l.size())); off += l.size(); } }
    hobbit_host_elastic_transcript &t = g_eopen;
    const int maxr = 11 + logr + logr + (logt + logc) + logc;
    t.queries = queries; t.cols.assign(queries, 0); t.rows.assign(queries, 0); t.reply.assign((size_t)queries * K, F(0)); t.paths.assign((size_t)queries * depth * 32, 0);
    t.qpoly.assign(3 * (size_t)maxr, F(0)); t.r.assign(maxr, F(0)); t.vr.assign(8, F(0)); t.fin.assign(4, F(0)); t.rx.assign(logc + logt, F(0));
    SpBuffers bf(t.sp_f, B, 32);
    hobbit_elastic_open_out o{t.cols.data(), t.rows.data(), hF(&t.rv0), hF(t.reply.data()), &t.reply_len, t.paths.data(), t.cf_root, &t.ncols,
                              hF(t.qpoly.data()), hF(t.r.data()), hF(t.vr.data()), hF(t.fin.data()), t.checks, hF(t.rx.data()), &bf.o};
    HCHK(hobbit_elastic_open_finish(g_ctx, e, tot == 8 * B - 1 ? (const uint8_t *)lv.p : nullptr, &o));
    hobbit_elastic_open_free(e);
    if (!t.checks[0] || !t.checks[1]) { printf("Error in fft\n"); exit(-1); }                                   // src/sumcheck.cpp:3016-3019
    if (t.sp_f.iters && !(t.sp_f.wchecks[0] && t.sp_f.wchecks[1])) { printf("Error in final verification step\n"); exit(-1); }
    auto t0 = std::chrono::steady_clock::now();
    size_t np2 = 1; while (np2 < (size_t)t.ncols) np2 <<= 1;
    t.rounds = (int)log2((double)(np2 * 2 * trs)) + logr + (logt + logc) + logc;
    double MT_ps = 0.0;
    vector<size_t> pos(queries);
    for (int i = 0; i < queries; i++) pos[i] = (size_t)(t.rows[i] / 4) * 4096 + t.cols[i];
    path_ps(4 * B, depth, pos, MT_ps);                                                                          // (:684-688)
    ps += (double)((size_t)queries * (size_t)t.reply_len * sizeof(F)) / 1024.0;                                 // (:701)
    sumcheck2_ps((int)log2((double)(np2 * 2 * trs)), ps); sumcheck2_ps(logr, ps); sumcheck2_ps(logt + logc, ps); sumcheck2_ps(logc, ps);   // P0, P2, P3, P5
    shockwave_ps(t.sp_f, B, 32, ps);
    ps += MT_ps;
    vt += std::chrono::duration_cast<std::chrono::duration<double>>(std::chrono::steady_clock::now() - t0).count();
    for (auto &l : Commitment_MT) { l.clear(); vector<_hash>(l).swap(l); }                                      // (:691-696)
    Commitment_MT.clear();
    printf("PC : ps = %lf, vt = %lf\n", ps, vt);
}
void test_Elastic_PC(size_t N, int option) {                        // src/Elastic_PC.cpp:736-771
    _hash comm; vector<vector<_hash>> MT_hashes;
    stream_descriptor commit_data; commit_data.name = "test"; commit_data.size = N;
    if (option == 1) { linear_time = false; tensor_row_size = (int)(BUFFER_SPACE / (1ULL << 11)); }
    else if (option == 2) { linear_time = true; int K = (int)(N / BUFFER_SPACE); tensor_row_size = (int)(N / (K * 1ULL << 14)); printf("> %d\n", tensor_row_size); expander_init_store(tensor_row_size); }
    else { printf("Error: option %d (the Brakedown streaming baseline) is a comparison baseline, not built\n", option); exit(-1); }
    auto start = std::chrono::steady_clock::now();
    commit(commit_data, comm, MT_hashes);
    auto end = std::chrono::steady_clock::now();
    double elapsed = std::chrono::duration_cast<std::chrono::duration<double>>(end - start).count();
    std::cout << "Commit time: " << elapsed << " seconds" << std::endl;
    if (option == 1) printf("Commitment finished\n");
    printf("root ");
    for (int i = 0; i < 32; i++) printf("%02x", MT_hashes.back()[0].arr[i]);
    printf("\n");
    double vt = 0.0, ps = 0.0;
    start = std::chrono::steady_clock::now();
    open(commit_data, generate_randomness((int)log2((double)N)), MT_hashes, vt, ps);
    end = std::chrono::steady_clock::now();
    elapsed += std::chrono::duration_cast<std::chrono::duration<double>>(end - start).count();
    if (option == 1) printf("Total prover time : %lf\n", elapsed);                                              // (:759)
    else std::cout << "Total time: " << elapsed << " seconds" << std::endl;                                      // (:783)
}
void test_Elastic_PC_commit(size_t N, int option) {                 // commit phase only (kept for callers that time the commit alone)
    _hash comm; vector<vector<_hash>> MT_hashes;
    stream_descriptor commit_data; commit_data.name = "test"; commit_data.size = N;
    if (option == 1) { linear_time = false; tensor_row_size = (int)(BUFFER_SPACE / (1ULL << 11)); }
    else { linear_time = true; int K = (int)(N / BUFFER_SPACE); tensor_row_size = (int)(N / (K * 1ULL << 14)); printf("> %d\n", tensor_row_size); expander_init_store(tensor_row_size); }
    auto start = std::chrono::steady_clock::now();
    commit(commit_data, comm, MT_hashes);
    auto end = std::chrono::steady_clock::now();
    std::cout << "Commit time: " << std::chrono::duration_cast<std::chrono::duration<double>>(end - start).count() << " seconds" << std::endl;
    printf("root ");
    for (int i = 0; i < 32; i++) printf("%02x", MT_hashes.back()[0].arr[i]);
    printf("\n");
}

// ---- remaining reference-named entry points of the path ------------------------------------------------------
shockwave_data *C_f = nullptr, *C_c = nullptr;                     // src/PC_utils.cpp:6-7
F *scratch[2][100];                                                // src/linear_code_encode.cpp:3
bool __encode_initialized = false;                                 // src/linear_code_encode.cpp:4
double routine_time = 0.0, sc_vt = 0.0;                            // src/sumcheck.cpp:27,29

mul_tree_proof prove_multiplication_tree_new(vector<vector<F>> &input, F previous_r, vector<F> prev_x, double &vt, double &ps) {   // src/sumcheck.cpp:35-257
    size_t vectors = input.size(), size = input[0].size();
    for (auto &v : input) if (v.size() != size) { printf("Error in mul tree sumcheck, no equal size vectors\n"); exit(-1); }
    int depth = (int)log2((double)size);
    if (((size_t)1 << depth) != size) { depth++; size = (size_t)1 << depth; for (auto &v : input) v.resize(size, F(1)); }          // (:48-54)
    if (vectors != ((size_t)1 << (int)log2((double)vectors))) {                                                                        // (:56-64)
        size_t nv = (size_t)1 << ((int)log2((double)vectors) + 1);
        for (size_t i = vectors; i < nv; i++) input.push_back(vector<F>(size, F(0)));
        vectors = nv;
    }
    vector<F> flat(vectors * size);
    for (size_t j = 0; j < vectors; j++) memcpy((void *)(flat.data() + j * size), input[j].data(), size * sizeof(F));
    DevBuf d(flat.data(), flat.size() * sizeof(F));
    const int lt = (int)log2((double)(vectors * size)); size_t nr = 0; for (int i = 0; i < lt; i++) nr += (size_t)i;
    vector<F> q(4 * (nr + 1)), r(nr + 1), vr(3 * (size_t)depth), fin(depth), final_r(lt), out(vectors); F oe, fe; int layers = 0;
    HCHK(hobbit_mul_tree(hobbit_host_ctx(), (const hobbit_F *)d.p, vectors, size, hF(&previous_r), prev_x.empty() ? nullptr : hF(prev_x.data()), hF(q.data()), hF(r.data()),
                         hF(vr.data()), hF(fin.data()), hF(final_r.data()), hF(&oe), hF(&fe), &layers));
    mul_tree_proof P; P.size = size; P.initial_randomness = previous_r; P.out_eval = oe; P.final_eval = fe; P.final_r = final_r;
    size_t qo = 0, ro = 0; int rl = vectors == 1 ? 1 : (int)log2((double)vectors);
    for (int l = 0; l < layers; l++) {
        struct proof p;
        for (int i = 0; i < rl; i++) p.c_poly.push_back(cubic_poly(q[qo + 4 * i], q[qo + 4 * i + 1], q[qo + 4 * i + 2], q[qo + 4 * i + 3]));
        p.randomness.push_back(vector<F>(r.begin() + ro, r.begin() + ro + rl)); p.vr = {vr[3 * l], vr[3 * l + 1], vr[3 * l + 2]}; p.final_rand = fin[l];
        P.proofs.push_back(p); qo += 4 * (size_t)rl; ro += (size_t)rl; rl++;
        ps += (p.c_poly.size() * 5 + 3) * sizeof(F) / 1024.0;
    }
    for (int i = 0; i < depth; i++) P.individual_randomness.push_back(final_r[i]);                                                     // (:221-229)
    for (int i = depth; i < lt; i++) P.global_randomness.push_back(final_r[i]);
    P.output.resize(vectors);                                                                                                           // products per vector
    for (size_t j = 0; j < vectors; j++) { F m = F(1); for (size_t i = 0; i < size; i++) m = m * input[j][i]; P.output[j] = m; }
    (void)vt;
    return P;
}
struct proof batch_3product_sumcheck(vector<vector<F>> &arr1, vector<vector<F>> &arr2, vector<vector<F>> &arr3, vector<F> a, double &vt, double &ps) {   // src/sumcheck.cpp:275-372
    const int batches = (int)a.size();
    vector<size_t> lens(batches); size_t tot = 0, L = 0;
    for (int j = 0; j < batches; j++) { lens[j] = arr1[j].size(); tot += lens[j]; L = lens[j] > L ? lens[j] : L; }
    vector<F> t1(tot), t2(tot), t3(tot); size_t o = 0;
    for (int j = 0; j < batches; j++) { memcpy((void *)(t1.data() + o), arr1[j].data(), lens[j] * sizeof(F)); memcpy((void *)(t2.data() + o), arr2[j].data(), lens[j] * sizeof(F));
                                        memcpy((void *)(t3.data() + o), arr3[j].data(), lens[j] * sizeof(F)); o += lens[j]; }
    DevBuf d1(t1.data(), tot * sizeof(F)), d2(t2.data(), tot * sizeof(F)), d3(t3.data(), tot * sizeof(F));
    const int rounds = (int)log2((double)L);
    vector<F> q(4 * (size_t)rounds), r(rounds), vr(3 * (size_t)batches);
    HCHK(hobbit_batch_3product_sumcheck(hobbit_host_ctx(), (const hobbit_F *)d1.p, (const hobbit_F *)d2.p, (const hobbit_F *)d3.p, lens.data(), batches, hF(a.data()),
                                        hF(q.data()), hF(r.data()), hF(vr.data())));
    struct proof P;
    for (int i = 0; i < rounds; i++) P.c_poly.push_back(cubic_poly(q[4 * i], q[4 * i + 1], q[4 * i + 2], q[4 * i + 3]));
    P.randomness.push_back(r); P.vr = vr;
    ps += rounds * 4 * sizeof(F) / 1024.0 + (P.vr.size() - P.vr.size() / 3) * sizeof(F) / 1024.0; (void)vt;
    // the reference folds its inputs in place; callers rely on element 0 afterwards (= vr)
    for (int j = 0; j < batches; j++) { arr1[j][0] = vr[3 * j]; arr2[j][0] = vr[3 * j + 1]; arr3[j][0] = vr[3 * j + 2]; }
    return P;
}
void _compute_tensorcode(F *message, F **tensor, int size) {                                                   // src/PC_utils.cpp:9-64
    vector<F> m(message, message + size); vector<vector<F>> t;
    compute_tensorcode(m, t);
    for (size_t i = 0; i < t.size(); i++) memcpy((void *)tensor[i], t[i].data(), t[i].size() * sizeof(F));
}
shockwave_data::~shockwave_data() {
    if (k > 0) { for (int i = 0; i < k; i++) { delete[] encoded_matrix[i]; delete[] matrix[i]; } delete[] encoded_matrix; delete[] matrix; }
    for (void *p : {d_matrix, d_enc, d_levels}) if (p && g_ctx) hobbit_free(g_ctx, p);
}
shockwave_data *shockwave_commit(vector<F> &poly, int k) {                                                     // src/Virgo.cpp:120-157
    shockwave_data *d = new shockwave_data; d->k = k; d->N = (int)poly.size();
    const size_t N = poly.size(), w = N / k, W = 2 * w;
    HCHK(hobbit_malloc(hobbit_host_ctx(), N * sizeof(F), &d->d_matrix)); HCHK(hobbit_malloc(g_ctx, 2 * N * sizeof(F), &d->d_enc)); HCHK(hobbit_malloc(g_ctx, 64 * W, &d->d_levels));
    HCHK(hobbit_memcpy_h2d(g_ctx, d->d_matrix, poly.data(), N * sizeof(F)));
    HCHK(hobbit_shockwave_commit(g_ctx, (const hobbit_F *)d->d_matrix, N, k, (hobbit_F *)d->d_enc, (uint8_t *)d->d_levels));
    d->matrix = new F *[k]; d->encoded_matrix = new F *[k];
    for (int i = 0; i < k; i++) {
        d->matrix[i] = new F[w]; d->encoded_matrix[i] = new F[W];
        memcpy((void *)d->matrix[i], poly.data() + (size_t)i * w, w * sizeof(F));
        HCHK(hobbit_memcpy_d2h(g_ctx, d->encoded_matrix[i], (const char *)d->d_enc + (size_t)i * W * sizeof(F), W * sizeof(F)));
    }
    const int levels = (int)log2((double)W) + 1; d->MT.resize(levels); size_t off = 0;
    for (int l = 0, sz = (int)W; l < levels; l++, sz /= 2) { d->MT[l].resize(sz); HCHK(hobbit_memcpy_d2h(g_ctx, d->MT[l].data(), (const char *)d->d_levels + 32 * off, 32 * (size_t)sz)); off += sz; }
    return d;
}
static hobbit_host_shockwave_transcript g_sw;
hobbit_host_shockwave_transcript &hobbit_host_last_shockwave() { return g_sw; }
void shockwave_prove(shockwave_data *data, vector<F> x, double &vt, double &ps) {                              // src/Virgo.cpp:435-517
    SpBuffers b(g_sw, (size_t)data->N, data->k);
    HCHK(hobbit_shockwave_prove(hobbit_host_ctx(), (const hobbit_F *)data->d_matrix, (const hobbit_F *)data->d_enc, (const uint8_t *)data->d_levels, (size_t)data->N, data->k,
                                hF(x.data()), (int)x.size(), &b.o));
    if (g_sw.iters && !(g_sw.wchecks[0] && g_sw.wchecks[1])) { printf("Error in final verification step\n"); exit(-1); }               // src/Virgo.cpp:562-565, 648-651
    shockwave_ps(g_sw, (size_t)data->N, data->k, ps); (void)vt;
    delete data;                                                                                                                      // (:515)
}
// the streaming multiplication-tree prover over read_stream: the stream is re-generated on the host exactly as the reference does, every
// read is uploaded and handed to the library as a chunk source
struct HostStream { stream_descriptor fd; vector<F> buf; DevBuf *dev = nullptr; size_t cap = 0; size_t small_n = 0; };
#ifdef HOBBIT_HOST_REFERENCE_BUILD
#include <dlfcn.h>
static int &seval_ring_len() {                                     // the Seval ring's length, `int BUFFER_SPACE_tr` (src/main.cpp:58); a data symbol of
    static int *p = (int *)dlsym(RTLD_DEFAULT, "BUFFER_SPACE_tr"); //  the library loaded AFTER this one, so looked up at run time
    if (!p) { printf("Error: BUFFER_SPACE_tr not found (reference build of the mirror without the reference loaded)\n"); exit(-1); }
    return *p;
}
#endif
static int host_stream_source(void *user, size_t n, const hobbit_F **out) {
    HostStream *hs = (HostStream *)user;
    if (n == 0) { hs->fd.pos = 0; hs->fd.idx = 0; hs->fd.stage = 0; hs->fd.offset = 0; hs->fd.finished = false; return 0; }          // reset_stream (src/witness_stream.cpp:228-234)
    auto t0 = std::chrono::steady_clock::now();
    if (hs->buf.size() != n) hs->buf.resize(n);                     // (some branches of the reference's read_stream size their work by v.size())
#ifdef HOBBIT_HOST_REFERENCE_BUILD
    // a product layer no longer than BUFFER_SPACE is read with the oracle's ring shortened to match (src/sumcheck.cpp:1802-1808): the
    // reference's trace readers copy a whole ring into the buffer without a bound
    const bool small = hs->small_n && n == hs->small_n;
    if (small) seval_ring_len() = (int)(n / 16);
    TIMED_READ(read_stream(hs->fd, hs->buf, (int)n));
    if (small) seval_ring_len() = (int)(BUFFER_SPACE / 8);
#else
    TIMED_READ(read_stream(hs->fd, hs->buf, (int)n));
#endif
    if (hs->cap < n) { delete hs->dev; hs->dev = new DevBuf(n * sizeof(F)); hs->cap = n; }
    { double t__ = now_s(); if (hobbit_memcpy_h2d(g_ctx, hs->dev->p, hs->buf.data(), n * sizeof(F)) != 0) return 1; g_t_up += now_s() - t__; }
    routine_time += std::chrono::duration_cast<std::chrono::duration<double>>(std::chrono::steady_clock::now() - t0).count();       // "streaming time" (src/sumcheck.cpp:1185-1188)
    *out = (const hobbit_F *)hs->dev->p;
    return 0;
}
static int pc_layer_chunk(stream_descriptor &raw, int layer, size_t B, void *d_out) {
    HostStream hs; hs.fd = raw;
    int rc = hobbit_read_mul_tree_layer(hobbit_host_ctx(), host_stream_source, &hs, B, layer, (hobbit_F *)d_out);
    if (rc == 0) rc = hobbit_sync(g_ctx);               // the source's upload buffer dies with hs
    delete hs.dev;
    return rc;
}
// src/sumcheck.cpp:983-1003
void commit_layers(stream_descriptor fd, vector<stream_descriptor> &fd_com, vector<vector<vector<_hash>>> &MT_hashes, int batches, int layer_id, int distance) {
    printf("%lld,%d\n", (long long)fd.size, (int)(1ULL << layer_id));
    const size_t size = fd.size / (1ULL << layer_id);
    if (batches - 1 <= 0) return;
    fd_com.resize(batches - 1); MT_hashes.resize(batches - 1);
    for (int i = 0; i < batches - 1; i++) {
        fd_com[i] = stream_descriptor(); fd_com[i].name = "PC_layer"; fd_com[i].size = size / (1ULL << (distance * i)); fd_com[i].layer = layer_id + i * distance;
        _hash comm;
        init_commitment(false);
        printf("Committing to: %d\n", (int)fd_com[i].size);
        if (fd_com[i].size > BUFFER_SPACE) commit(fd_com[i], comm, MT_hashes[i]);
    }
}
// src/sumcheck.cpp:1005-1011.  Elastic_PC::open reads its two passes through read_stream, which serves a "PC_layer" descriptor from its
// default branch -- the raw default stream, not the product layer the commitment was computed over (the reference as it is).
void open_layers(vector<stream_descriptor> &fd_com, vector<vector<vector<_hash>>> &MT_hashes, double &vt, double &ps) {
    for (size_t i = 0; i < fd_com.size(); i++)
        if (fd_com[i].size > BUFFER_SPACE) open(fd_com[i], generate_randomness((int)log2((double)fd_com[i].size)), MT_hashes[i], vt, ps);
}
struct S3Buffers {   // host buffers behind one hobbit_stream3_out
    vector<F> nc, nr, c1, r1, vr1, q2, r2, vr2, fin2, R; int checks[3] = {0, 0, 0}; hobbit_stream3_out o;
    S3Buffers(size_t fd_size, size_t B, int batches, int layer_id) {
        const size_t size = fd_size >> layer_id; const int logB = (int)log2((double)B), lR = (int)log2((double)(size / (2 * B))); const int ld = 1 + logB + lR;
        nc.assign(batches, F(0)); nr.assign((size_t)batches * ld, F(0)); c1.assign(4 * (size_t)logB, F(0)); r1.assign(logB, F(0)); vr1.assign(3 * (size_t)batches, F(0));
        q2.assign(3 * (size_t)lR, F(0)); r2.assign(lR, F(0)); vr2.assign(2, F(0)); fin2.assign(1, F(0)); R.assign(size / (2 * B), F(0));
        o = hobbit_stream3_out{hF(nc.data()), hF(nr.data()), ld, hF(c1.data()), hF(r1.data()), hF(vr1.data()), hF(q2.data()), hF(r2.data()), hF(vr2.data()), hF(fin2.data()), hF(R.data()), checks};
    }
};
static void s3_check(const S3Buffers &b, double &ps, int batches, int rounds1, int rounds2) {
    if (!b.checks[1]) { printf("Error in sumcheck 1\n"); exit(-1); }                                            // src/sumcheck.cpp:1280-1283
    if (!b.checks[2]) { printf("Error in sumcheck 2\n"); exit(-1); }                                            // (:1362-1365)
    ps += (1 + batches) * sizeof(F) / 1024.0;                                                                   // (:1212)
    ps += (double)(((size_t)1 << rounds2) - 1) * (1 + batches) * sizeof(F) / 1024.0;                            // one batch_prod per half chunk after the first (:1128; R ends 2^rounds2 long)
    ps += rounds1 * 4 * sizeof(F) / 1024.0 + 2 * batches * sizeof(F) / 1024.0;                                  // batch_3product_sumcheck
    ps += rounds2 * 3 * sizeof(F) / 1024.0 + 2 * sizeof(F) / 1024.0;                                            // the closing 2-product sumcheck
}
void generate_3product_sumcheck_beta_stream_batch_optimized(stream_descriptor fd, vector<vector<F>> r, int batches, int distance, int layer_id, vector<F> old_claims,
                                                            vector<F> &new_claims, vector<vector<F>> &new_r, double &vt, double &ps) {   // src/sumcheck.cpp:1150-1393
    HostStream hs; hs.fd = fd;
    size_t rlen = 0; for (auto &v : r) rlen = v.size() > rlen ? v.size() : rlen;
    vector<F> rr((size_t)batches * rlen, F(0));
    for (int i = 0; i < batches; i++) memcpy((void *)(rr.data() + (size_t)i * rlen), r[i].data(), r[i].size() * sizeof(F));
    S3Buffers b(fd.size, BUFFER_SPACE, batches, layer_id);
    HCHK(hobbit_sumcheck3_stream_batch(hobbit_host_ctx(), host_stream_source, &hs, fd.size, BUFFER_SPACE, hF(rr.data()), (int)rlen, batches, distance, layer_id,
                                       hF(old_claims.data()), (int)old_claims.size(), &b.o));
    delete hs.dev;
    for (size_t i = 0; i < old_claims.size(); i++) if (!b.checks[0]) { printf("Error in sumcheck 0 %d\n", (int)i); break; }   // the reference prints and continues (:1246-1251)
    const int logB = (int)log2((double)BUFFER_SPACE), lR = (int)log2((double)((fd.size >> layer_id) / (2 * BUFFER_SPACE)));
    s3_check(b, ps, batches, logB, lR); (void)vt;
    new_claims = b.nc; new_r.resize(batches);
    for (int i = 0; i < batches; i++) new_r[i].assign(b.nr.begin() + (size_t)i * b.o.new_r_ld, b.nr.begin() + (size_t)i * b.o.new_r_ld + 1 + (logB - i * distance) + lR);
}
vector<F> prove_multiplication_tree_stream_shallow(stream_descriptor fd, int vectors, int size, F previous_r, int distance, vector<F> prev_x, bool naive, double &vt, double &ps) {   // src/sumcheck.cpp:1746-1915
    PhaseTimer pt__("multiplication tree (stream)");
    const size_t total = (size_t)size * vectors, B = BUFFER_SPACE;
    int layers = total > 2 * B ? (int)log2((double)(total / (2 * B))) : 0;
    if (layers % distance != 0 && layers > distance) layers = distance + layers - (layers % distance);
    vector<stream_descriptor> fd_com; vector<vector<vector<_hash>>> MT_layers;
    if (total > 2 * B && !naive) commit_layers(fd, fd_com, MT_layers, layers / distance, distance - 1, distance);        // (:1791-1795)
    HostStream hs; hs.fd = fd;
    const size_t n1 = total > 2 * B ? fd.size >> layers : total; const size_t sz = n1 / vectors;
    if (total > 2 * B && n1 <= B) hs.small_n = 2 * n1;              // read_mul_tree_layer's read length for that layer (src/witness_stream.cpp:2436)
    const int lt = (int)log2((double)n1), depth = (int)log2((double)sz); size_t nr = 0; for (int i = 0; i < lt; i++) nr += (size_t)i;
    vector<F> out(vectors), q(4 * (nr + 1)), r(nr + 1), vr(3 * (size_t)depth + 3), fin(depth + 1), final_r(lt); F oe, fe; int tl = 0, nst = 0, sl = 0;
    vector<S3Buffers *> bufs; vector<hobbit_stream3_out> steps;
    if (total > 2 * B) {
        if (layers <= distance || naive) for (int i = layers - 1; i >= 0; i--) bufs.push_back(new S3Buffers(fd.size, B, 1, i));
        else for (int i = distance - 1; i >= 0; i--) bufs.push_back(new S3Buffers(fd.size, B, layers / distance, i));
        for (auto *b : bufs) steps.push_back(b->o);
    }
    vector<F> claims0(16);
    hobbit_mul_stream_out mo{hF(out.data()), hF(q.data()), hF(r.data()), hF(vr.data()), hF(fin.data()), hF(final_r.data()), hF(&oe), hF(&fe), &tl,
                             steps.data(), (int)steps.size(), &nst, hF(claims0.data()), &sl};
    HCHK(hobbit_mul_tree_stream_shallow(hobbit_host_ctx(), host_stream_source, &hs, fd.size, B, vectors, (size_t)size, hF(&previous_r), distance,
                                        prev_x.empty() ? nullptr : hF(prev_x.data()), naive ? 1 : 0, &mo));
    delete hs.dev;
    const int logB = (int)log2((double)B);
    for (int l = 0, rl = vectors == 1 ? 1 : (int)log2((double)vectors); l < tl; l++, rl++)                        // the in-memory tree's layers (:1772/:1820 ->
        ps += ((size_t)rl * 5 + 3) * sizeof(F) / 1024.0;                                                         //  _generate_3product_sumcheck_proof, :2031,2048)
    for (int i = 0; i < nst; i++) {
        printf(layers <= distance || naive ? "OK %d\n" : "~OK %d\n", nst - 1 - i);                                // (:1861, :1889)
        if (!bufs[i]->checks[0]) printf("Error in sumcheck 0 0\n");
        const int batches = (int)bufs[i]->nc.size();
        s3_check(*bufs[i], ps, batches, logB, (int)bufs[i]->r2.size());
    }
    for (auto *b : bufs) delete b;
    if (total > 2 * B && !naive) open_layers(fd_com, MT_layers, vt, ps);                                                      // (:1910-1912)
    return out;
}

// ---- prove_circuit_standard (src/main.cpp:985-1087) ---------------------------------------------------
size_t circuit_size = 0;
void (*hobbit_read_trace_hook)(stream_descriptor &, vector<F> &, vector<F> &, vector<F> &, vector<int> &) = nullptr;
void (*hobbit_read_memory_hook)(stream_descriptor &, vector<F> &, vector<F> &, vector<F> &) = nullptr;
#ifndef HOBBIT_HOST_REFERENCE_BUILD
void reset_stream(stream_descriptor &fd) { fd.pos = 0; fd.idx = 0; fd.stage = 0; fd.offset = 0; fd.finished = false; }
#endif
// (prove_circuit_standard itself is main.cpp's: a driver relinked against this mirror keeps its own copy.  tests/test_mlp_end_to_end.py runs the
// reference's compiled one over these functions; rounds 1-2 carried a restatement of it here, dropped in round 3.)

// ---- prove_gate_consistency / _lookups (src/sumcheck.cpp:796-975, 503-795) over the witness generator's read_trace ----------------
bool has_lookups = false;
vector<F> lookup_rand;
int tensor_code = 1;
static hobbit_host_gate_transcript g_gate;
hobbit_host_gate_transcript &hobbit_host_last_gate() { return g_gate; }
struct HostTrace {                                                   // hobbit_trace_source over hobbit_read_trace_hook
    stream_descriptor fd; size_t B;
    vector<F> bL, bR, bO; vector<int> bS;
    DevBuf dL, dR, dO, dS;
    HostTrace(const stream_descriptor &f, size_t b) : fd(f), B(b), bL(b), bR(b), bO(b), bS(b), dL(b * sizeof(F)), dR(b * sizeof(F)), dO(b * sizeof(F)), dS(b * sizeof(int32_t)) {}
};
static int host_trace_source(void *user, size_t n, const hobbit_F **L, const hobbit_F **R, const hobbit_F **O, const int32_t **S) {
    HostTrace *t = (HostTrace *)user;
    if (n == 0) { reset_stream(t->fd); return 0; }                                                              // (:871 / :643)
    if (n != t->B || !hobbit_read_trace_hook) return -1;
    TIMED_READ(hobbit_read_trace_hook(t->fd, t->bL, t->bR, t->bO, t->bS));
    double t__ = now_s();
    if (hobbit_memcpy_h2d(g_ctx, t->dL.p, t->bL.data(), n * sizeof(F)) || hobbit_memcpy_h2d(g_ctx, t->dR.p, t->bR.data(), n * sizeof(F)) ||
        hobbit_memcpy_h2d(g_ctx, t->dO.p, t->bO.data(), n * sizeof(F)) || hobbit_memcpy_h2d(g_ctx, t->dS.p, t->bS.data(), n * sizeof(int32_t))) return -1;
    g_t_up += now_s() - t__;
    *L = (const hobbit_F *)t->dL.p; *R = (const hobbit_F *)t->dR.p; *O = (const hobbit_F *)t->dO.p; *S = (const int32_t *)t->dS.p;
    return 0;
}
static void gate_stream_common(stream_descriptor &tr, vector<F> &r, bool lookups, double &vt, double &ps) {
    if (!hobbit_read_trace_hook) { printf("Error: the streaming gate provers need the witness generator's read_trace (hobbit_read_trace_hook)\n"); exit(-1); }
    PhaseTimer pt__(lookups ? "gate consistency (lookups)" : "gate consistency");
    hobbit_host_ctx();
    const size_t B = BUFFER_SPACE, nch = tr.size / B;
    const int logB = (int)log2((double)B), lR = (int)log2((double)nch);
    const int nt = lookups ? 9 : 6, na = lookups ? 5 : 4, nb = lookups ? 8 : 6;
    hobbit_host_gate_transcript &t = g_gate;
    t.lookups = lookups;
    t.R.assign(nch, F(0)); t.a.assign(na, F(0)); t.poly.assign(5 * (size_t)logB, F(0)); t.gr.assign(logB, F(0)); t.fin.assign(nt, F(0));
    t.Peval.assign((size_t)nb * nch, F(0)); t.b.assign(nb, F(0)); t.q2.assign(3 * (size_t)lR, F(0)); t.r2.assign(lR, F(0)); t.vr2.assign(2, F(0));
    for (int &c : t.checks) c = 1;
    HostTrace src(tr, B);
    if (lookups) {
        if (!has_lookups || lookup_rand.size() < 2) { printf("Error: prove_gate_consistency_lookups needs has_lookups and lookup_rand (src/main.cpp:888,910)\n"); exit(-1); }
        HCHK(hobbit_set_lookups(g_ctx, 1, hF(lookup_rand.data())));
        hobbit_gate_lkp_stream_out o{hF(t.R.data()), hF(t.a.data()), hF(t.poly.data()), hF(t.gr.data()), hF(t.fin.data()), hF(t.Peval.data()), hF(t.b.data()),
                                     hF(t.q2.data()), hF(t.r2.data()), hF(t.vr2.data()), hF(&t.fin2), t.checks};
        HCHK(hobbit_gate_consistency_lookups_stream(g_ctx, host_trace_source, &src, nch, B, hF(r.data()), &o));
        HCHK(hobbit_set_lookups(g_ctx, has_lookups ? 1 : 0, hF(lookup_rand.data())));
    } else {
        hobbit_gate_stream_out o{hF(t.R.data()), hF(t.a.data()), hF(t.poly.data()), hF(t.gr.data()), hF(t.fin.data()), hF(t.Peval.data()), hF(t.b.data()),
                                 hF(t.q2.data()), hF(t.r2.data()), hF(t.vr2.data()), hF(&t.fin2), t.checks};
        HCHK(hobbit_gate_consistency_stream(g_ctx, host_trace_source, &src, nch, B, hF(r.data()), &o));
    }
    if (!t.checks[0]) { printf("Error in gate consistency 1\n"); exit(-1); }                                   // (:840 / :588, :549)
    if (!t.checks[1]) { printf("Error in gate consistency 2\n"); exit(-1); }                                   // (:912 / :711)
    if (!t.checks[2]) { printf("Error in gate consistency 3\n"); exit(-1); }                                   // (:968 / :787)
    if (lookups && !t.checks[3]) { printf("ERRRROR\n"); exit(-1); }                                            // (:638-641)
    if (lookups && !t.checks[4]) printf("ERRRROR\n");                                                          // (:650-652: the reference only prints)
    // proof-size accounting of the reference (:827,860,919,971 / :555,612,719,783) + the closing 2-product sumcheck's own
    ps += (lookups ? 5 : 4) * sizeof(F) / 1024.0;
    ps += (double)(nch - 1) * (lookups ? 15 : 12) * sizeof(F) / 1024.0;
    ps += (double)logB * 5 * sizeof(F) / 1024.0;
    sumcheck2_ps(lR, ps);
    ps += 5 * sizeof(F) / 1024.0;
    (void)vt;
}
void prove_gate_consistency(stream_descriptor tr, vector<F> r, double &vt, double &ps) { gate_stream_common(tr, r, false, vt, ps); }
void prove_gate_consistency_lookups(stream_descriptor tr, vector<F> r, double &vt, double &ps) { gate_stream_common(tr, r, true, vt, ps); }

// ---- proof wire format (hobbit_host.hpp) ---------------------------------------------------------------
namespace {
struct Wr {
    std::vector<uint8_t> b;
    void raw(const void *p, size_t n) { const uint8_t *q = (const uint8_t *)p; b.insert(b.end(), q, q + n); }
    void u32(uint32_t v) { raw(&v, 4); }
    void u64(uint64_t v) { raw(&v, 8); }
    void sec(uint32_t tag, const void *p, size_t n) { u32(tag); u64(n); raw(p, n); }
    template <class T> void vec(uint32_t tag, const std::vector<T> &v) { sec(tag, v.data(), v.size() * sizeof(T)); }
};
struct Rd {
    const uint8_t *p; size_t n, at = 0; bool ok = true;
    Rd(const uint8_t *q, size_t m) : p(q), n(m) {}
    bool raw(void *d, size_t k) { if (!ok || k > n - at) return ok = false; memcpy(d, p + at, k); at += k; return true; }
    uint32_t u32() { uint32_t v = 0; raw(&v, 4); return v; }
    uint64_t u64() { uint64_t v = 0; raw(&v, 8); return v; }
    bool next(uint32_t &tag, const uint8_t *&pay, size_t &len) {
        if (at == n) return false;
        tag = u32(); const uint64_t l = u64();
        if (!ok || l > n - at) { ok = false; return false; }
        pay = p + at; len = (size_t)l; at += len; return true;
    }
};
template <class T> bool take(const uint8_t *pay, size_t len, std::vector<T> &v) { if (len % sizeof(T)) return false; v.resize(len / sizeof(T)); if (len) memcpy((void *)v.data(), pay, len); return true; }
enum : uint32_t { T_COLS = 1, T_ROWS, T_REPLY, T_PATHS, T_QPOLY, T_R, T_VR, T_FIN, T_SCAL, T_ROOTS, T_SPC, T_SPF, T_RV0, T_CFROOT, T_RX, T_NCOLS,
                  S_I = 100, S_Q1, S_R1, S_VR1, S_FIN1, S_Q2, S_R2, S_VR2, S_FIN2, S_WQ, S_WA, S_WSCAL, S_REPLY, S_PATHS, S_WROOTS, S_WHIRROOT, S_WQIDX, S_WQN, S_WQREPLY,
                  S_WQPATHS, S_WFINAL, S_ITERS };
std::vector<uint8_t> ser_sp(const hobbit_host_shockwave_transcript &t) {
    Wr w;
    w.vec(S_I, t.I); w.vec(S_Q1, t.q1); w.vec(S_R1, t.r1); w.vec(S_VR1, t.vr1); w.sec(S_FIN1, &t.fin1, sizeof(F)); w.vec(S_Q2, t.q2); w.vec(S_R2, t.r2); w.vec(S_VR2, t.vr2);
    w.sec(S_FIN2, &t.fin2, sizeof(F)); w.vec(S_WQ, t.wq); w.vec(S_WA, t.wa); w.vec(S_WSCAL, t.wscal); w.vec(S_REPLY, t.reply); w.vec(S_PATHS, t.paths); w.vec(S_WROOTS, t.wroots);
    w.sec(S_WHIRROOT, t.whir_root, 32); w.vec(S_WQIDX, t.wqidx); w.vec(S_WQN, t.wqn); w.vec(S_WQREPLY, t.wqreply); w.vec(S_WQPATHS, t.wqpaths); w.vec(S_WFINAL, t.wfinal);
    const uint32_t it = (uint32_t)t.iters; w.sec(S_ITERS, &it, 4);
    return w.b;
}
bool de_sp(const uint8_t *buf, size_t n, hobbit_host_shockwave_transcript &t) {
    Rd r(buf, n); uint32_t tag; const uint8_t *pay; size_t len; bool ok = true;
    while (ok && r.next(tag, pay, len)) switch (tag) {
        case S_I: ok = take(pay, len, t.I); break; case S_Q1: ok = take(pay, len, t.q1); break; case S_R1: ok = take(pay, len, t.r1); break; case S_VR1: ok = take(pay, len, t.vr1); break;
        case S_FIN1: ok = len == sizeof(F); if (ok) memcpy((void *)&t.fin1, pay, len); break; case S_Q2: ok = take(pay, len, t.q2); break; case S_R2: ok = take(pay, len, t.r2); break;
        case S_VR2: ok = take(pay, len, t.vr2); break; case S_FIN2: ok = len == sizeof(F); if (ok) memcpy((void *)&t.fin2, pay, len); break; case S_WQ: ok = take(pay, len, t.wq); break;
        case S_WA: ok = take(pay, len, t.wa); break; case S_WSCAL: ok = take(pay, len, t.wscal); break; case S_REPLY: ok = take(pay, len, t.reply); break;
        case S_PATHS: ok = take(pay, len, t.paths); break; case S_WROOTS: ok = take(pay, len, t.wroots); break; case S_WHIRROOT: ok = len == 32; if (ok) memcpy(t.whir_root, pay, 32); break;
        case S_WQIDX: ok = take(pay, len, t.wqidx); break; case S_WQN: ok = take(pay, len, t.wqn); break; case S_WQREPLY: ok = take(pay, len, t.wqreply); break;
        case S_WQPATHS: ok = take(pay, len, t.wqpaths); break; case S_WFINAL: ok = take(pay, len, t.wfinal); break;
        case S_ITERS: ok = len == 4; if (ok) { uint32_t it; memcpy(&it, pay, 4); t.iters = (int)it; } break;
        default: break;
    }
    return ok && r.ok;
}
void header(Wr &w, uint32_t kind, uint32_t nsec) { w.raw("HBPF", 4); w.u32(1); w.u32(kind); w.u32(nsec); }
bool check_header(Rd &r, uint32_t kind, uint32_t &nsec) { char m[4]; if (!r.raw(m, 4) || memcmp(m, "HBPF", 4)) return false; if (r.u32() != 1 || r.u32() != kind) return false; nsec = r.u32(); return r.ok; }
// a well-formed proof has every required section exactly once and as many sections as its header announces (unknown tags count, and are skipped)
struct Sections {
    std::set<uint32_t> seen; uint32_t n = 0; bool dup = false;
    void note(uint32_t tag) { n++; if (!seen.insert(tag).second) dup = true; }
    bool complete(std::initializer_list<uint32_t> required, uint32_t announced) const {
        if (dup || n != announced) return false;
        for (uint32_t t : required) if (!seen.count(t)) return false;
        return true;
    }
};
}  // namespace
std::vector<uint8_t> hobbit_host_serialize_open(const hobbit_host_open_transcript &t) {
    Wr w; header(w, 1, 12);
    w.vec(T_COLS, t.cols); w.vec(T_ROWS, t.rows); w.vec(T_REPLY, t.reply); w.vec(T_PATHS, t.paths); w.vec(T_QPOLY, t.qpoly); w.vec(T_R, t.r); w.vec(T_VR, t.vr); w.vec(T_FIN, t.fin);
    w.vec(T_SCAL, t.scalars); w.sec(T_ROOTS, t.roots, 64);
    { auto s = ser_sp(t.sp_c); w.sec(T_SPC, s.data(), s.size()); } { auto s = ser_sp(t.sp_f); w.sec(T_SPF, s.data(), s.size()); }
    return w.b;
}
bool hobbit_host_deserialize_open(const uint8_t *buf, size_t n, hobbit_host_open_transcript &t) {
    uint32_t nsec = 0; Sections sec;
    Rd r(buf, n); if (!check_header(r, 1, nsec)) return false;
    uint32_t tag; const uint8_t *pay; size_t len; bool ok = true;
    while (ok && r.next(tag, pay, len)) switch (sec.note(tag), tag) {
        case T_COLS: ok = take(pay, len, t.cols); break; case T_ROWS: ok = take(pay, len, t.rows); break; case T_REPLY: ok = take(pay, len, t.reply); break;
        case T_PATHS: ok = take(pay, len, t.paths); break; case T_QPOLY: ok = take(pay, len, t.qpoly); break; case T_R: ok = take(pay, len, t.r); break; case T_VR: ok = take(pay, len, t.vr); break;
        case T_FIN: ok = take(pay, len, t.fin); break; case T_SCAL: ok = take(pay, len, t.scalars); break; case T_ROOTS: ok = len == 64; if (ok) memcpy(t.roots, pay, 64); break;
        case T_SPC: ok = de_sp(pay, len, t.sp_c); break; case T_SPF: ok = de_sp(pay, len, t.sp_f); break;
        default: break;
    }
    if (ok && r.ok) { t.queries = (int)t.cols.size(); t.rounds = (int)t.r.size(); }
    return ok && r.ok && t.cols.size() == t.rows.size() &&
           sec.complete({T_COLS, T_ROWS, T_REPLY, T_PATHS, T_QPOLY, T_R, T_VR, T_FIN, T_SCAL, T_ROOTS, T_SPC, T_SPF}, nsec);
}
std::vector<uint8_t> hobbit_host_serialize_rs_open(const hobbit_host_elastic_transcript &t) {
    Wr w; header(w, 2, 13);
    const size_t nr = (size_t)t.rounds;
    { const int32_t nc = t.ncols; w.sec(T_NCOLS, &nc, 4); }
    w.vec(T_COLS, t.cols); w.vec(T_ROWS, t.rows);
    w.sec(T_REPLY, t.reply.data(), (size_t)t.queries * (size_t)t.reply_len * sizeof(F)); w.vec(T_PATHS, t.paths);
    w.sec(T_QPOLY, t.qpoly.data(), 3 * nr * sizeof(F)); w.sec(T_R, t.r.data(), nr * sizeof(F)); w.vec(T_VR, t.vr); w.vec(T_FIN, t.fin);
    w.sec(T_RV0, &t.rv0, sizeof(F)); w.sec(T_CFROOT, t.cf_root, 32); w.vec(T_RX, t.rx);
    { auto s = ser_sp(t.sp_f); w.sec(T_SPF, s.data(), s.size()); }
    return w.b;
}
bool hobbit_host_deserialize_rs_open(const uint8_t *buf, size_t n, hobbit_host_elastic_transcript &t) {
    uint32_t nsec = 0; Sections sec;
    Rd r(buf, n); if (!check_header(r, 2, nsec)) return false;
    uint32_t tag; const uint8_t *pay; size_t len; bool ok = true;
    while (ok && r.next(tag, pay, len)) switch (sec.note(tag), tag) {
        case T_NCOLS: ok = len == 4; if (ok) { int32_t nc; memcpy(&nc, pay, 4); t.ncols = nc; } break;
        case T_COLS: ok = take(pay, len, t.cols); break; case T_ROWS: ok = take(pay, len, t.rows); break; case T_REPLY: ok = take(pay, len, t.reply); break;
        case T_PATHS: ok = take(pay, len, t.paths); break; case T_QPOLY: ok = take(pay, len, t.qpoly); break; case T_R: ok = take(pay, len, t.r); break; case T_VR: ok = take(pay, len, t.vr); break;
        case T_FIN: ok = take(pay, len, t.fin); break; case T_RV0: ok = len == sizeof(F); if (ok) memcpy((void *)&t.rv0, pay, len); break;
        case T_CFROOT: ok = len == 32; if (ok) memcpy(t.cf_root, pay, 32); break; case T_RX: ok = take(pay, len, t.rx); break; case T_SPF: ok = de_sp(pay, len, t.sp_f); break;
        default: break;
    }
    if (ok && r.ok) { t.queries = (int)t.cols.size(); t.rounds = (int)t.r.size(); t.reply_len = t.queries ? (int)(t.reply.size() / (size_t)t.queries) : 0; }
    return ok && r.ok && t.cols.size() == t.rows.size() &&
           sec.complete({T_NCOLS, T_COLS, T_ROWS, T_REPLY, T_PATHS, T_QPOLY, T_R, T_VR, T_FIN, T_RV0, T_CFROOT, T_RX, T_SPF}, nsec);
}

// ---- driver (src/Our_PC.cpp:757-826, option 4, commit phase) ---------------------------------------
void test_PC(size_t N, int option, int K) {
    if (option != 4 && option != 1) { printf("Error: options 2 and 3 are the Orion / Brakedown comparison baselines, not built on the device path\n"); exit(-1); }
    vector<F> poly = generate_randomness((int)N);
    _hash comm; vector<vector<_hash>> MT_hashes;
    if (option == 1) { linear_time = false; tensor_row_size = 128; }                     // (:764-766: RS x RS)
    else {
        linear_time = true;
        tensor_row_size = (int)(N / (K * 1ULL << (11)));
        printf("%d\n", tensor_row_size);
        expander_init_store(tensor_row_size);
    }
    vector<vector<vector<F>>> _tensor;
    auto start = std::chrono::steady_clock::now();
    commit_standard(poly, comm, MT_hashes, _tensor, K);
    auto end = std::chrono::steady_clock::now();
    double commit_time = std::chrono::duration_cast<std::chrono::duration<double>>(end - start).count();
    std::cout << "Commit time: " << commit_time << " seconds" << std::endl;
    printf("root ");
    for (int i = 0; i < 32; i++) printf("%02x", MT_hashes.back()[0].arr[i]);
    printf("\n");
    double vt = 0.0, ps = 0.0;
    start = std::chrono::steady_clock::now();
    open_standard(poly, generate_randomness((int)log2((double)poly.size())), MT_hashes, _tensor, K, vt, ps);
    end = std::chrono::steady_clock::now();
    double total = commit_time + std::chrono::duration_cast<std::chrono::duration<double>>(end - start).count();
    std::cout << "Total time: " << total << " seconds" << std::endl;
    printf("%lf,%lf\n", ps, vt);
}

// ---- extern "C" hooks so tests can drive the C++ mirror through ctypes -------------------------------
extern "C" {
int hobbit_host_test_pc_root(size_t N, int K, uint8_t *root_out) {
    srandom(1);
    vector<F> poly = generate_randomness((int)N);
    linear_time = true; tensor_row_size = (int)(N / (K * 1ULL << 11));
    expander_init_store(tensor_row_size);
    _hash comm; vector<vector<_hash>> MT; vector<vector<vector<F>>> T;
    commit_standard(poly, comm, MT, T, K);
    memcpy(root_out, MT.back()[0].arr, 32);
    return (int)MT.size();
}
// hobbit_host_upload_graphs: wipe the device's graphs, re-upload them from _C / D, commit again -- the root must not change
int hobbit_host_graph_reupload_check(size_t N, int K) {
    srandom(1);
    vector<F> poly = generate_randomness((int)N);
    linear_time = true; tensor_row_size = (int)(N / (K * 1ULL << 11));
    expander_init_store(tensor_row_size);
    _hash comm; vector<vector<_hash>> MT, MT2; vector<vector<vector<F>>> T;
    commit_standard(poly, comm, MT, T, K);
    hobbit_host_upload_graphs(tensor_row_size);
    commit_standard(poly, comm, MT2, T, K);
    return memcmp(MT.back()[0].arr, MT2.back()[0].arr, 32) == 0 ? 1 : 0;
}
// commit + open through the C++ mirror on test_PC's inputs; returns the transcript pieces a test compares with the oracle
int hobbit_host_test_pc_open(size_t N, int K, unsigned seed, uint64_t *qpoly, uint64_t *r, uint32_t *cols_rows, uint8_t *sp_roots /* C_f, C_c, whir_c, whir_f */, int *checks5, double *ps_out) {
    srandom(1);
    vector<F> poly = generate_randomness((int)N);
    linear_time = true; tensor_row_size = (int)(N / (K * 1ULL << 11));
    expander_init_store(tensor_row_size);
    _hash comm; vector<vector<_hash>> MT; vector<vector<vector<F>>> T;
    commit_standard(poly, comm, MT, T, K);
    vector<F> x = generate_randomness((int)log2((double)N));
    srandom(seed);
    double vt = 0, ps = 0;
    open_standard(poly, x, MT, T, K, vt, ps);
    hobbit_host_open_transcript &t = hobbit_host_last_open();
    memcpy(qpoly, t.qpoly.data(), 16 * t.qpoly.size()); memcpy(r, t.r.data(), 16 * t.r.size());
    for (int i = 0; i < t.queries; i++) { cols_rows[2 * i] = t.cols[i]; cols_rows[2 * i + 1] = t.rows[i]; }
    memcpy(sp_roots, t.roots, 64); memcpy(sp_roots + 64, t.sp_c.whir_root, 32); memcpy(sp_roots + 96, t.sp_f.whir_root, 32);
    checks5[0] = t.checks[0]; checks5[1] = t.checks[1]; checks5[2] = t.checks[2]; checks5[3] = t.sp_c.wchecks[0] & t.sp_c.wchecks[1]; checks5[4] = t.sp_f.wchecks[0] & t.sp_f.wchecks[1];
    *ps_out = ps;
    return t.rounds;
}
// test_PC(N, 1, K) through the mirror (linear_time == false, tensor_row_size = 128): root, then the opening's transcript pieces
int hobbit_host_test_pc_rs_open(size_t N, int K, unsigned seed, uint8_t *root_out, uint64_t *qpoly, uint64_t *r, uint32_t *cols_rows, uint8_t *roots2 /* C_f, whir_f */, int *checks3, double *ps_out) {
    srandom(1);
    vector<F> poly = generate_randomness((int)N);
    linear_time = false; tensor_row_size = 128;
    _hash comm; vector<vector<_hash>> MT; vector<vector<vector<F>>> T;
    commit_standard(poly, comm, MT, T, K);
    memcpy(root_out, MT.back()[0].arr, 32);
    vector<F> x = generate_randomness((int)log2((double)N));
    srandom(seed);
    double vt = 0, ps = 0;
    open_standard(poly, x, MT, T, K, vt, ps);
    hobbit_host_elastic_transcript &t = hobbit_host_last_elastic_open();
    memcpy(qpoly, t.qpoly.data(), 48 * (size_t)t.rounds); memcpy(r, t.r.data(), 16 * (size_t)t.rounds);
    for (int i = 0; i < t.queries; i++) { cols_rows[2 * i] = t.cols[i]; cols_rows[2 * i + 1] = t.rows[i]; }
    memcpy(roots2, t.cf_root, 32); memcpy(roots2 + 32, t.sp_f.whir_root, 32);
    checks3[0] = t.checks[0]; checks3[1] = t.checks[1]; checks3[2] = t.sp_f.iters ? (t.sp_f.wchecks[0] & t.sp_f.wchecks[1]) : 1;
    *ps_out = ps;
    linear_time = true;
    return t.rounds;
}
// synthetic read_trace / read_memory for the tests (L, R, O: n F; S: n int; addr, value, access: F)
static struct { const F *L, *R, *O; const int *S; const F *addr, *value, *access; } g_syn;
static void syn_read_trace(stream_descriptor &fd, vector<F> &bL, vector<F> &bR, vector<F> &bO, vector<int> &bS) {
    const size_t B = bL.size(), at = fd.pos * B; fd.pos++;
    for (size_t j = 0; j < B; j++) { bL[j] = g_syn.L[at + j]; bR[j] = g_syn.R[at + j]; bO[j] = g_syn.O[at + j]; bS[j] = g_syn.S[at + j]; }
}
static void syn_read_memory(stream_descriptor &fd, vector<F> &ba, vector<F> &bv, vector<F> &bc) {
    const size_t B = ba.size(), at = fd.pos * B; fd.pos++;
    for (size_t j = 0; j < B; j++) { ba[j] = g_syn.addr[at + j]; bv[j] = g_syn.value[at + j]; bc[j] = g_syn.access[at + j]; }
}
// prove_gate_consistency[_lookups] through the mirror over a caller-supplied trace (L, R, O: n F; S: n int): the challenges R and the final
// folded values come back for the comparison with the oracle
int hobbit_host_gate_stream(size_t n, size_t B, int lookups, unsigned seed, const uint64_t *L, const uint64_t *R, const uint64_t *O, const int *S, const uint64_t *r,
                            const uint64_t *lr, uint64_t *R_out, uint64_t *fin_out, uint64_t *q2_out, int *checks, double *ps_out) {
    g_syn.L = (const F *)L; g_syn.R = (const F *)R; g_syn.O = (const F *)O; g_syn.S = S;
    hobbit_read_trace_hook = syn_read_trace;
    BUFFER_SPACE = B;
    stream_descriptor tr; tr.name = "transcript_stream"; tr.size = n; reset_stream(tr);
    const int logB = (int)log2((double)B);
    vector<F> rv(logB); memcpy((void *)rv.data(), r, 16 * (size_t)logB);
    if (lookups) { has_lookups = true; lookup_rand.assign(4, F(0)); memcpy((void *)lookup_rand.data(), lr, 32); }
    srandom(seed);
    double vt = 0, ps = 0;
    if (lookups) prove_gate_consistency_lookups(tr, rv, vt, ps); else prove_gate_consistency(tr, rv, vt, ps);
    has_lookups = false;
    hobbit_host_gate_transcript &t = hobbit_host_last_gate();
    memcpy(R_out, t.R.data(), 16 * t.R.size()); memcpy(fin_out, t.fin.data(), 16 * t.fin.size()); memcpy(q2_out, t.q2.data(), 16 * t.q2.size());
    for (int i = 0; i < 5; i++) checks[i] = t.checks[i];
    *ps_out = ps;
    return (int)t.fin.size();
}
// round trip of the wire format on the transcripts of the last open_standard (kind 1) / last RS x RS opening (kind 2): returns the proof size
// in bytes, or a negative number naming what failed
long hobbit_host_wire_roundtrip(int kind) {
    std::vector<uint8_t> a, b;
    if (kind == 1) {
        a = hobbit_host_serialize_open(hobbit_host_last_open());
        hobbit_host_open_transcript t;
        if (!hobbit_host_deserialize_open(a.data(), a.size(), t)) return -1;
        b = hobbit_host_serialize_open(t);
        if (hobbit_host_deserialize_open(a.data(), a.size() - 7, t)) return -3;          // a truncated proof must be rejected
    } else {
        a = hobbit_host_serialize_rs_open(hobbit_host_last_elastic_open());
        hobbit_host_elastic_transcript t;
        if (!hobbit_host_deserialize_rs_open(a.data(), a.size(), t)) return -1;
        b = hobbit_host_serialize_rs_open(t);
        if (hobbit_host_deserialize_rs_open(a.data(), a.size() - 7, t)) return -3;
    }
    if (a != b) return -2;
    std::vector<uint8_t> c = a; c[0] ^= 1;
    hobbit_host_open_transcript t1; hobbit_host_elastic_transcript t2;
    if (kind == 1 ? hobbit_host_deserialize_open(c.data(), c.size(), t1) : hobbit_host_deserialize_rs_open(c.data(), c.size(), t2)) return -4;     // wrong magic
    auto parses = [&](const std::vector<uint8_t> &v) { return kind == 1 ? hobbit_host_deserialize_open(v.data(), v.size(), t1) : hobbit_host_deserialize_rs_open(v.data(), v.size(), t2); };
    // header: "HBPF" | version | kind | section count (16 bytes); a section: u32 tag | u64 bytes | payload
    uint64_t l0; memcpy(&l0, a.data() + 16 + 4, 8);
    const size_t first_end = 16 + 12 + (size_t)l0;
    auto set_count = [](std::vector<uint8_t> &v, int delta) { uint32_t n; memcpy(&n, v.data() + 12, 4); n = (uint32_t)((int)n + delta); memcpy(v.data() + 12, &n, 4); };
    {   std::vector<uint8_t> d = a; d.insert(d.end(), a.begin() + 16, a.begin() + first_end); set_count(d, +1);      // a section twice, announced
        if (parses(d)) return -5; }
    {   std::vector<uint8_t> d(a.begin(), a.begin() + 16); d.insert(d.end(), a.begin() + first_end, a.end()); set_count(d, -1);   // a required section missing, count consistent
        if (parses(d)) return -6; }
    {   std::vector<uint8_t> d = a; set_count(d, +1);                                                                       // header announces more sections than follow
        if (parses(d)) return -7; }
    return (long)a.size();
}
int hobbit_host_sumcheck2(const uint64_t *v1, const uint64_t *v2, size_t n, const uint64_t *prev, uint64_t *qpoly, uint64_t *r, uint64_t *vr, uint64_t *fin) {
    vector<F> a(n), b(n); memcpy((void *)a.data(), v1, 16 * n); memcpy((void *)b.data(), v2, 16 * n);
    F p; memcpy((void *)&p, prev, 16);
    double vt = 0, ps = 0;
    proof P = generate_2product_sumcheck_proof(a, b, p, vt, ps);
    for (size_t i = 0; i < P.q_poly.size(); i++) { memcpy(qpoly + 6 * i, &P.q_poly[i].a, 16); memcpy(qpoly + 6 * i + 2, &P.q_poly[i].b, 16); memcpy(qpoly + 6 * i + 4, &P.q_poly[i].c, 16); }
    memcpy(r, P.randomness[0].data(), 16 * P.randomness[0].size());
    memcpy(vr, P.vr.data(), 32); memcpy(fin, &P.final_rand, 16);
    return (int)P.q_poly.size();
}
int hobbit_host_elastic_root(size_t N, size_t B, int option, uint8_t *root_out) {
    srandom(1);
    BUFFER_SPACE = B;
    _hash comm; vector<vector<_hash>> MT;
    stream_descriptor fd; fd.name = "test"; fd.size = N;
    if (option == 1) { linear_time = false; tensor_row_size = (int)(B >> 11); }
    else { linear_time = true; tensor_row_size = (int)(B >> 14); expander_init_store(tensor_row_size); }
    commit(fd, comm, MT);
    memcpy(root_out, MT.back()[0].arr, 32);
    return (int)MT.size();
}
// test_Elastic_PC(N, 1) through the mirror from a fresh generator state; hands back root, queries, replies and the four transcripts
int hobbit_host_elastic_open(size_t N, size_t B, uint8_t *root_out, uint32_t *cols_rows, uint64_t *reply, uint64_t *qpoly, uint64_t *r, int *checks, double *ps_out) {
    srandom(1);
    BUFFER_SPACE = B; linear_time = false; tensor_row_size = (int)(B >> 11);
    _hash comm; vector<vector<_hash>> MT;
    stream_descriptor fd; fd.name = "test"; fd.size = N;
    commit(fd, comm, MT);
    memcpy(root_out, MT.back()[0].arr, 32);
    double vt = 0, ps = 0;
    open(fd, generate_randomness((int)log2((double)N)), MT, vt, ps);
    hobbit_host_elastic_transcript &t = hobbit_host_last_elastic_open();
    for (int i = 0; i < t.queries; i++) { cols_rows[2 * i] = t.cols[i]; cols_rows[2 * i + 1] = t.rows[i]; }
    memcpy(reply, t.reply.data(), 16 * (size_t)t.queries * t.reply_len);
    memcpy(qpoly, t.qpoly.data(), 16 * 3 * (size_t)t.rounds); memcpy(r, t.r.data(), 16 * (size_t)t.rounds);
    checks[0] = t.checks[0]; checks[1] = t.checks[1]; checks[2] = t.sp_f.wchecks[0] & t.sp_f.wchecks[1];
    *ps_out = ps;
    return t.rounds;
}
// test_Elastic_PC(N, 2) through the mirror from a fresh generator state (graphs, commit, x, open under linear_time); hands back root, queries,
// replies (the reference's row order), the four transcripts of recursive_prover_Spielman_stream, C_c's root and the proof size
int hobbit_host_elastic_open2(size_t N, size_t B, uint8_t *root_out, uint32_t *cols_rows, uint64_t *reply, uint64_t *qpoly, uint64_t *r, int *checks, double *ps_out,
                              int *nrem_out, uint8_t *cc_root_out) {
    srandom(1);
    BUFFER_SPACE = B; linear_time = true; tensor_row_size = (int)(B >> 14);
    expander_init_store(tensor_row_size);
    _hash comm; vector<vector<_hash>> MT;
    stream_descriptor fd; fd.name = "test"; fd.size = N;
    commit(fd, comm, MT);
    memcpy(root_out, MT.back()[0].arr, 32);
    double vt = 0, ps = 0;
    open(fd, generate_randomness((int)log2((double)N)), MT, vt, ps);
    hobbit_host_elastic_transcript &t = hobbit_host_last_elastic_open();
    for (int i = 0; i < t.queries; i++) { cols_rows[2 * i] = t.cols[i]; cols_rows[2 * i + 1] = t.rows[i]; }
    memcpy(reply, t.reply.data(), 16 * (size_t)t.queries * t.reply_len);
    memcpy(qpoly, t.qpoly.data(), 16 * 3 * (size_t)t.rounds); memcpy(r, t.r.data(), 16 * (size_t)t.rounds);
    checks[0] = t.checks[0]; checks[1] = t.sp_c.iters ? (t.sp_c.wchecks[0] & t.sp_c.wchecks[1]) : 1; checks[2] = t.sp_f.iters ? (t.sp_f.wchecks[0] & t.sp_f.wchecks[1]) : 1;
    *ps_out = ps; *nrem_out = t.nrem; memcpy(cc_root_out, t.cc_root, 32);
    return t.rounds;
}
// one call through each of the remaining reference-named wrappers; results for the test to compare with the oracle
int hobbit_host_mirror_check(const uint64_t *tree_in /* 4 x 64 F */, const uint64_t *b3 /* 3 tables: 256 | 64 F each */, const uint64_t *b3a /* 2 F */, uint64_t *out /* >= 64 F */,
                             uint8_t *roots /* 2 x 32 B */, double *ps_out) {
    double vt = 0, ps = 0; F *o = (F *)out; int n = 0;
    {   // prove_multiplication_tree_new
        vector<vector<F>> in(4, vector<F>(64)); for (int j = 0; j < 4; j++) memcpy((void *)in[j].data(), tree_in + 2 * 64 * j, 64 * sizeof(F));
        srandom(77);
        mul_tree_proof P = prove_multiplication_tree_new(in, F(17, 5), vector<F>(), vt, ps);
        o[n++] = P.out_eval; o[n++] = P.final_eval; for (auto &v : P.output) o[n++] = v; o[n++] = P.final_r[0]; o[n++] = P.final_r.back(); o[n++] = F((long long)P.proofs.size());
    }
    {   // batch_3product_sumcheck
        vector<vector<F>> A(2), Bv(2), C(2); size_t lens[2] = {256, 64}, off = 0;
        for (int j = 0; j < 2; j++) { A[j].assign((const F *)b3 + off, (const F *)b3 + off + lens[j]); Bv[j].assign((const F *)b3 + 320 + off, (const F *)b3 + 320 + off + lens[j]);
                                      C[j].assign((const F *)b3 + 640 + off, (const F *)b3 + 640 + off + lens[j]); off += lens[j]; }
        vector<F> a((const F *)b3a, (const F *)b3a + 2);
        struct proof P = batch_3product_sumcheck(A, Bv, C, a, vt, ps);
        o[n++] = P.c_poly[0].a; o[n++] = P.c_poly.back().d; o[n++] = P.randomness[0].back(); for (auto &v : P.vr) o[n++] = v;
    }
    {   // _compute_tensorcode (RS x RS, trs = 4) and shockwave_commit / shockwave_prove, globals C_f / C_c
        vector<F> msg((const F *)tree_in, (const F *)tree_in + 256);
        linear_time = false; tensor_row_size = 4; BUFFER_SPACE = 256;
        vector<vector<F>> rows(8, vector<F>(128)); vector<F *> ptr(8); for (int i = 0; i < 8; i++) ptr[i] = rows[i].data();
        _compute_tensorcode(msg.data(), ptr.data(), 256);
        o[n++] = rows[0][0]; o[n++] = rows[7][127]; o[n++] = rows[3][64];
        vector<F> big(1 << 13); for (size_t i = 0; i < big.size(); i++) big[i] = ((const F *)tree_in)[i % 256] * F((long long)(i / 256 + 1));
        C_f = shockwave_commit(big, 32);
        memcpy(roots, C_f->MT.back()[0].arr, 32); o[n++] = C_f->encoded_matrix[31][511]; o[n++] = C_f->matrix[5][7];
        C_c = shockwave_commit(msg, 8); memcpy(roots + 32, C_c->MT.back()[0].arr, 32); delete C_c; C_c = nullptr;
        vector<F> x((const F *)b3a, (const F *)b3a + 2); x.resize(13, F(3)); for (int i = 2; i < 13; i++) x[i] = ((const F *)b3)[i];
        srandom(78);
        shockwave_prove(C_f, x, vt, ps); C_f = nullptr;
        hobbit_host_shockwave_transcript &t = hobbit_host_last_shockwave();
        o[n++] = t.q1[0]; o[n++] = t.fin1; o[n++] = t.fin2; o[n++] = F((long long)t.iters); o[n++] = F((long long)(t.wchecks[0] + 2 * t.wchecks[1]));
    }
    *ps_out = ps;
    return n;
}
// prove_multiplication_tree_stream_shallow through the mirror on the default stream (returns the number of products)
// commit_layers through the mirror: roots of the layer commitments (zero where a layer fits one buffer)
int hobbit_host_commit_layers(size_t fd_size, size_t B, int batches, int layer_id, int distance, uint8_t *roots) {
    BUFFER_SPACE = B;
    stream_descriptor fd; fd.name = "test"; fd.size = fd_size;
    vector<stream_descriptor> fc; vector<vector<vector<_hash>>> MT;
    commit_layers(fd, fc, MT, batches, layer_id, distance);
    for (size_t i = 0; i < fc.size(); i++) { if (!MT[i].empty()) memcpy(roots + 32 * i, MT[i].back()[0].arr, 32); else memset(roots + 32 * i, 0, 32); }
    return (int)fc.size();
}
int hobbit_host_mul_tree_stream(size_t B, int vectors, size_t size, int distance, const uint64_t *prev_x, int nx, uint64_t *out, double *ps_out, int naive) {
    BUFFER_SPACE = B;
    stream_descriptor fd; fd.name = "test"; fd.size = size * (size_t)vectors;
    double vt = 0, ps = 0;
    vector<F> px((const F *)prev_x, (const F *)prev_x + nx);
    srandom(11);
    vector<F> o = prove_multiplication_tree_stream_shallow(fd, vectors, (int)size, F(32), distance, px, naive != 0, vt, ps);
    memcpy(out, o.data(), o.size() * sizeof(F)); *ps_out = ps;
    return (int)o.size();
}
void hobbit_host_close(void) { hobbit_host_shutdown(); }
}
