// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY.
//
// extern "C" entry points over the REAL reference implementation (compiled from /root/reference
// by oracle/Makefile `make ref` into oracle/_ref/libhobbit_ref.so).  Nothing here restates an
// algorithm: every function only marshals plain buffers into the std::vector / F** arguments the
// reference functions take and copies their results out.  Used (a) to generate the committed
// golden vectors under tests/golden/ (oracle/gen_golden.py), (b) to cross-check the C
// restatement in oracle/hobbit_oracle.c while /root/reference is present, and (c) as the
// "reference" CPU baseline timed by bench.py.
//
// F buffers are uint64_t[2] = {real, img}, exactly virgo::fieldElement (src/fieldElement.hpp:96-97).
// Hashes are uint8_t[32] = struct _hash (src/Blake3_hash.h:3-5).
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "config_pc.hpp"
#include "mimc.h"
#include "utils.hpp"
#include "merkle_tree.h"
#include "sumcheck.h"
#include "PC_utils.h"
#include "Our_PC.hpp"
#include "Elastic_PC.hpp"
#include "Virgo.h"
#include "linear_code_encode.h"

extern bool linear_time;                     // src/Our_PC.cpp:21
extern int tensor_row_size;                  // src/main.cpp:31
extern shockwave_data *C_f, *C_c;            // src/PC_utils.cpp:6-7
void _aggregate(vector<F> &poly, vector<F> beta1, vector<F> random_points, vector<F> &aggregated_vector,
                vector<vector<F>> &aggregated_tensor, bool linear_time, int K);   // src/Our_PC.cpp:258
void _compute_aggregation_reply(vector<vector<size_t>> &I, vector<vector<F>> &reply,
                                vector<vector<vector<F>>> &_tensor, int K);        // src/Our_PC.cpp:291

void compute2p_error_terms(vector<F> &buff1, vector<F> &buff2, vector<F> &fold_buff1, vector<F> &fold_buff2, F &K1, F &K2);                 // src/sumcheck.cpp:374
void compute4p_error_terms(vector<F> &buff1, vector<F> &buff2, vector<F> &buff3, vector<int> &buff4, vector<F> &fold_buff1, vector<F> &fold_buff2,
                           vector<F> &fold_buff3, vector<F> &fold_buff4, F &K1, F &K2, F &K3, F &K4);                                       // :381
void compute3p_error_terms(vector<F> &buff1, vector<int> &buff2, vector<F> &fold_buff1, vector<F> &fold_buff2, vector<F> &fold_buff3,
                           vector<F> &beta, F &K1, F &K2, F &K3);                                                                            // :408
void batch_prod(vector<vector<F>> &fold_buff1, vector<vector<F>> &fold_buff2, vector<vector<F>> &fold_buff3, vector<vector<F>> &buff1,
                vector<vector<F>> &buff2, vector<vector<F>> &buff3, int batches, vector<F> &R, F &Kf, vector<F> &K_partial,
                vector<vector<F>> &remaining_betas, vector<F> a, int i, double &vt, double &ps);                                            // :1093
struct proof batch_3product_sumcheck(vector<vector<F>> &arr1, vector<vector<F>> &arr2, vector<vector<F>> &arr3, vector<F> a, double &vt, double &ps);   // :275

// src/Elastic_PC.cpp:315, 316, 487: the query list is a file-scope global there; aggregate / compute_aggregation_reply are not in the header
extern vector<vector<size_t>> I;
extern vector<vector<F>> aux_commit;                                                                                                         // src/Elastic_PC.cpp:315
extern shockwave_data *C_f, *C_c;                                                                                                            // src/Virgo.cpp globals, as src/Elastic_PC.cpp:11 declares them
extern int aggregation_queries;                                                                                                              // src/Elastic_PC.cpp:10
void aggregate(stream_descriptor fd, vector<F> beta1, vector<F> random_points, vector<vector<_hash>> &MT_hashes, vector<F> &aggregated_vector,
               vector<vector<F>> &aggregated_tensor);                                                                                        // src/Elastic_PC.cpp:316
void compute_aggregation_reply(stream_descriptor fd, vector<vector<size_t>> &I, vector<vector<F>> &reply);                                  // src/Elastic_PC.cpp:487

void generate_3product_sumcheck_beta_stream_batch_optimized(stream_descriptor fd, vector<vector<F>> r, int batches, int distance, int layer_id, vector<F> old_claims,
                                                            vector<F> &new_claims, vector<vector<F>> &new_r, double &vt, double &ps);      // src/sumcheck.cpp:1150
void generate_claims_opt(stream_descriptor fd, vector<F> r, vector<F> &claims, int batches, int layer_id, int distance);                      // src/sumcheck.cpp:1014

void commit_layers(stream_descriptor fd, vector<stream_descriptor> &fd_com, vector<vector<vector<_hash>>> &MT_hashes, int batches, int layer_id, int distance);   // src/sumcheck.cpp:983

static_assert(sizeof(F) == 16, "fieldElement must be 16 bytes");
static_assert(sizeof(_hash) == 32, "_hash must be 32 bytes");

static inline F ldF(const uint64_t *p) { F f; f.real = p[0]; f.img = p[1]; return f; }
static inline void stF(uint64_t *p, const F &f) { p[0] = f.real; p[1] = f.img; }
static vector<F> vecF(const uint64_t *p, size_t n) { vector<F> v(n); if (n) memcpy((void *)v.data(), p, 16 * n); return v; }

// state kept between calls (one commitment at a time)
static vector<vector<_hash>> g_MT;
static vector<vector<vector<F>>> g_tensor;

extern bool has_lookups;            // src/main.cpp:70
extern vector<F> lookup_rand;       // src/main.cpp:67
// the reference's own drivers (C++ linkage; src/Our_PC.cpp:757, src/Elastic_PC.cpp:736)
void test_PC(size_t N, int option, int K);
void test_Elastic_PC(size_t N, int option);

extern "C" {

void ref_init(void) { init_hash(); }
// glibc: rand() and random() share one generator whose default seed is 1 -> fresh-process state.
void ref_rng_reset(void) { srandom(1); }
void ref_rng_seed(unsigned s) { srandom(s); }
long ref_libc_rand(void) { return rand(); }
long ref_libc_random(void) { return random(); }

// ---- field (src/fieldElement.cpp:34-96, 206-209) --------------------------------------------
void ref_f_add(const uint64_t *a, const uint64_t *b, uint64_t *o, size_t n) { for (size_t i = 0; i < n; i++) stF(o + 2 * i, ldF(a + 2 * i) + ldF(b + 2 * i)); }
void ref_f_sub(const uint64_t *a, const uint64_t *b, uint64_t *o, size_t n) { for (size_t i = 0; i < n; i++) stF(o + 2 * i, ldF(a + 2 * i) - ldF(b + 2 * i)); }
void ref_f_mul(const uint64_t *a, const uint64_t *b, uint64_t *o, size_t n) { for (size_t i = 0; i < n; i++) stF(o + 2 * i, ldF(a + 2 * i) * ldF(b + 2 * i)); }
void ref_f_neg(const uint64_t *a, uint64_t *o, size_t n) { for (size_t i = 0; i < n; i++) stF(o + 2 * i, -ldF(a + 2 * i)); }
void ref_f_inv(const uint64_t *a, uint64_t *o, size_t n) { for (size_t i = 0; i < n; i++) stF(o + 2 * i, ldF(a + 2 * i).inv()); }
void ref_root_of_unity(int logn, uint64_t *o) { stF(o, getRootOfUnity(logn)); }

// ---- mimc (src/mimc.cpp:95-107) ---------------------------------------------------------------
void ref_mimc(const uint64_t *x, const uint64_t *k, uint64_t *o, size_t n) { for (size_t i = 0; i < n; i++) stF(o + 2 * i, mimc_hash(ldF(x + 2 * i), ldF(k + 2 * i))); }

// ---- BLAKE3 / Merkle (src/Blake3_hash.cpp:5-10, src/merkle_tree.cpp:62-87,193-221,255-287,308-324)
void ref_blake3_64(const uint8_t *in, uint8_t *out, size_t n) { for (size_t i = 0; i < n; i++) blake3_hash((uint8_t *)in + 64 * i, out + 32 * i); }
void ref_hash_md(const uint64_t *xyzw, const uint8_t *prev, uint8_t *out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        _hash p; memcpy(p.arr, prev + 32 * i, 32);
        _hash r = merkle_tree::hash_double_field_element_merkle_damgard_blake(ldF(xyzw + 8 * i), ldF(xyzw + 8 * i + 2), ldF(xyzw + 8 * i + 4), ldF(xyzw + 8 * i + 6), p);
        memcpy(out + 32 * i, r.arr, 32);
    }
}
static size_t flatten_levels(const vector<vector<_hash>> &h, uint8_t *out) {
    size_t off = 0;
    for (auto &lv : h) { memcpy(out + 32 * off, lv.data(), 32 * lv.size()); off += lv.size(); }
    return off;
}
// leaves N/4 then levels; out must hold (2*(N/4)-1)*32 bytes. returns number of hashes written.
size_t ref_mt_commit_blake(const uint64_t *leafs, int N, uint8_t *out) {
    vector<F> v = vecF(leafs, N);
    vector<vector<_hash>> h;
    merkle_tree::merkle_tree_prover::MT_commit_Blake(v.data(), h, N);
    return flatten_levels(h, out);
}
size_t ref_create_tree_blake(const uint8_t *level0, int n, uint8_t *out) {
    vector<vector<_hash>> h((int)log2(n) + 1);
    h[0].resize(n); memcpy(h[0].data(), level0, 32 * (size_t)n);
    merkle_tree::merkle_tree_prover::create_tree_blake(n, h, 32, true);
    return flatten_levels(h, out);
}

// ---- expander graphs + encode (src/expanders.h:20-47,78-92; src/linear_code_encode.h:62-119) ---
long long ref_expander_init_store(long long n) { return expander_init_store(n); }
// kind 0 = _C[dep], 1 = D[dep]; returns L, fills R,degree
long long ref_graph_dims(int dep, int kind, long long *R, int *degree) {
    graph &g = kind ? D[dep] : _C[dep];
    *R = g.R; *degree = g.degree; return g.L;
}
void ref_graph_edges(int dep, int kind, long long *nbr, uint64_t *w) {
    graph &g = kind ? D[dep] : _C[dep];
    for (long long i = 0; i < g.L; i++)
        for (int j = 0; j < g.degree; j++) { nbr[i * g.degree + j] = g.neighbor[i][j]; stF(w + 2 * (i * g.degree + j), g.weight[i][j]); }
}
// overwrite the stored weights of one level (used to test full-range F_{p^2} weights)
void ref_graph_set_weights(int dep, int kind, const uint64_t *w) {
    graph &g = kind ? D[dep] : _C[dep];
    for (auto &rw : g.r_weight) rw.clear();
    for (auto &rn : g.r_neighbor) rn.clear();
    for (long long i = 0; i < g.L; i++)
        for (int j = 0; j < g.degree; j++) {
            g.weight[i][j] = ldF(w + 2 * (i * g.degree + j));
            g.r_neighbor[g.neighbor[i][j]].push_back(i);
            g.r_weight[g.neighbor[i][j]].push_back(g.weight[i][j]);
        }
}
// dst must hold 2n F (zero-initialised by us, as the callers' vector<F>(2n, 0) is)
int ref_encode_monolithic(const uint64_t *src, uint64_t *dst, long long n) {
    vector<F> s = vecF(src, n), d(2 * n, F(0));
    int len = encode_monolithic(s.data(), d.data(), n);
    memcpy(dst, d.data(), 32 * n);
    return len;
}
// the reference allocates its global scratch on first use only (linear_code_encode.h:64-72):
// re-arm it when a larger n follows a smaller one.
void ref_encode_reset_scratch(void) { __encode_initialized = false; }

// ---- FFT / eq-table / evaluation (src/utils.cpp:605-673, 467-527, 251-296, 789-802, 873-883) ---
void ref_fft_raw(uint64_t *arr, int logn, int inverse) { _fft((F *)arr, logn, inverse != 0); }
void ref_fft_vec(uint64_t *arr, int logn, int inverse) {
    vector<F> v = vecF(arr, (size_t)1 << logn); fft(v, logn, inverse != 0); memcpy(arr, v.data(), 16 * v.size());
}
void ref_precompute_beta(const uint64_t *r, int k, uint64_t *out) {
    vector<F> B; precompute_beta(vecF(r, k), B); memcpy(out, B.data(), 16 * B.size());
}
void ref_evaluate_vector(const uint64_t *v, size_t n, const uint64_t *r, int k, uint64_t *out) { stF(out, evaluate_vector(vecF(v, n), vecF(r, k))); }
void ref_generate_randomness(int n, uint64_t *out) { vector<F> x = generate_randomness(n); memcpy(out, x.data(), 16 * (size_t)n); }

// ---- tensor code (src/PC_utils.cpp:66-123) ------------------------------------------------------
// out: row-major (2*trs) x (2*M/trs)
void ref_compute_tensorcode(const uint64_t *msg, size_t M, int trs, int lin, uint64_t *out) {
    tensor_row_size = trs; linear_time = lin != 0;
    vector<F> m = vecF(msg, M); vector<vector<F>> t;
    compute_tensorcode(m, t);
    size_t cols = t[0].size();
    for (size_t i = 0; i < t.size(); i++) memcpy(out + 2 * i * cols, t[i].data(), 16 * cols);
}

// ---- Our_PC commit (src/Our_PC.cpp:146-171) -----------------------------------------------------
// levels_out: (2*M-1)*32 bytes (level 0 = M leaves, ... root). tensor kept for ref_tensor_get.
size_t ref_commit_standard(const uint64_t *poly, size_t N, int K, int trs, int lin, uint8_t *levels_out) {
    tensor_row_size = trs; linear_time = lin != 0;
    vector<F> p = vecF(poly, N);
    _hash comm; g_MT.clear(); g_tensor.clear();
    commit_standard(p, comm, g_MT, g_tensor, K);
    return flatten_levels(g_MT, levels_out);
}
void ref_tensor_get(int chunk, const uint32_t *rows, const uint32_t *cols, size_t nq, uint64_t *out) {
    for (size_t q = 0; q < nq; q++) stF(out + 2 * q, g_tensor[chunk][rows[q]][cols[q]]);
}
void ref_tensor_row(int chunk, int row, uint64_t *out) { memcpy(out, g_tensor[chunk][row].data(), 16 * g_tensor[chunk][row].size()); }
// path for query (col=c0,row=c1) (src/merkle_tree.cpp:308-324); returns depth
int ref_open_tree_blake(size_t c0, size_t c1, int columns, uint8_t *path_out) {
    vector<size_t> c = {c0, c1};
    vector<_hash> p = merkle_tree::merkle_tree_prover::open_tree_blake(g_MT, c, columns);
    memcpy(path_out, p.data(), 32 * p.size());
    return (int)p.size();
}
void ref_compute_aggregation_reply(const uint64_t *I, size_t nq, int K, uint64_t *reply) {
    extern int aggregation_queries; aggregation_queries = (int)nq;
    vector<vector<size_t>> II(nq); for (size_t q = 0; q < nq; q++) { II[q] = {(size_t)I[2 * q], (size_t)I[2 * q + 1]}; }
    vector<vector<F>> r; _compute_aggregation_reply(II, r, g_tensor, K);
    for (size_t q = 0; q < nq; q++) memcpy(reply + 2 * q * K, r[q].data(), 16 * (size_t)K);
}
void ref_release_commit(void) { g_MT.clear(); g_MT.shrink_to_fit(); g_tensor.clear(); g_tensor.shrink_to_fit(); }


// test_PC(N,4,K)'s commitment from a fresh-process generator state, run entirely inside the library (no numpy copy of a
// 4 GiB polynomial): srandom(1); poly = generate_randomness(N); expander_init_store(N/(K*2^11)); commit_standard
// (src/Our_PC.cpp:757-816).  Leaves g_MT / g_tensor alive for ref_open_tree_blake / ref_tensor_get.
size_t ref_test_pc_commit(size_t N, int K, uint8_t *levels_out) {
    srandom(1);
    vector<F> poly = generate_randomness((int)N);
    linear_time = true; tensor_row_size = (int)(N / ((size_t)K << 11));
    __encode_initialized = false;
    expander_init_store(tensor_row_size);
    _hash comm; g_MT.clear(); g_tensor.clear();
    commit_standard(poly, comm, g_MT, g_tensor, K);
    return flatten_levels(g_MT, levels_out);
}

// test_PC(N,4,K) from the same fresh generator state, commit AND open (src/Our_PC.cpp:757-826): commit_standard, x = generate_randomness(log2 N),
// open_standard.  The opening reaches SHA3 (my_hhash, from the prebuilt lib/libXKCP.a that is never linked) inside its first shockwave_prove, so this
// call does NOT return: the process dies on the unresolved symbol.  Used only under the transcript recorder with HOBBIT_REC_STREAM
// (oracle/gen_open_transcript.py): what the reference's own P1..P4 and the first shockwave_prove's sumchecks hashed before that point is the fixture.
void ref_test_pc_open(size_t N, int K) {
    srandom(1);
    vector<F> poly = generate_randomness((int)N);
    linear_time = true; tensor_row_size = (int)(N / ((size_t)K << 11));
    __encode_initialized = false;
    expander_init_store(tensor_row_size);
    _hash comm; g_MT.clear(); g_tensor.clear();
    commit_standard(poly, comm, g_MT, g_tensor, K);
    vector<F> x = generate_randomness((int)log2((double)N));
    double vt = 0.0, ps = 0.0;
    open_standard(poly, x, g_MT, g_tensor, K, vt, ps);
}

// The reference's OWN drivers from a fresh generator state, for the transcript recorder (oracle/gen_open_transcript.py).  Neither returns: every
// opening reaches SHA3 somewhere (first shockwave_prove / recursive_prover_RS / verify_claim_opt_blake) and the process dies on the unresolved symbol.
//   test_PC(N, option, K)        src/Our_PC.cpp:757-826   (option 1: RS x RS, tensor_row_size = 128; option 4: RS x expander)
//   test_Elastic_PC(N, option)   src/Elastic_PC.cpp:736-784  with BUFFER_SPACE = B   (option 1: RS x RS; option 2: RS x expander)
void ref_run_test_pc(size_t N, int option, int K) {
    srandom(1);
    if (option == 1) linear_time = false;
    __encode_initialized = false;
    test_PC(N, option, K);
}
void ref_run_test_elastic(size_t N, size_t B, int option) {
    srandom(1);
    BUFFER_SPACE = B;
    __encode_initialized = false;
    test_Elastic_PC(N, option);
}

// aggregation axpy only (src/Our_PC.cpp:258-272): aggr[j] = sum_i beta[i]*poly[i*M+j].
// The reference function continues into shockwave_commit; we call the real function and free
// what it allocates.
void ref_aggregate(const uint64_t *poly, size_t N, const uint64_t *beta, int K, int trs, int lin, uint64_t *aggr_out) {
    tensor_row_size = trs; linear_time = lin != 0;
    vector<F> p = vecF(poly, N), b = vecF(beta, K), rv(K, F(1)), aggr; vector<vector<F>> at;
    _aggregate(p, b, rv, aggr, at, linear_time, K);
    memcpy(aggr_out, aggr.data(), 16 * aggr.size());
    delete C_f; C_f = nullptr;
    if (linear_time) { delete C_c; C_c = nullptr; }
}


// _aggregate as above, also handing back the roots of the two inner commitments it makes (src/Our_PC.cpp:274-287):
// roots[0..32) = C_f (shockwave_commit(aggr, 32)), roots[32..64) = C_c (shockwave_commit(parity half of the aggregate's tensor code, 32)).
void ref_aggregate_roots(const uint64_t *poly, size_t N, const uint64_t *beta, int K, int trs, uint64_t *aggr_out, uint8_t *roots) {
    tensor_row_size = trs; linear_time = true;
    vector<F> p = vecF(poly, N), b = vecF(beta, K), rv(K, F(1)), aggr; vector<vector<F>> at;
    _aggregate(p, b, rv, aggr, at, true, K);
    memcpy(aggr_out, aggr.data(), 16 * aggr.size());
    memcpy(roots, C_f->MT.back()[0].arr, 32); memcpy(roots + 32, C_c->MT.back()[0].arr, 32);
    delete C_f; C_f = nullptr; delete C_c; C_c = nullptr;
}

// ---- sumchecks (src/sumcheck.cpp:2391-2460, 1974-2058) ------------------------------------------
// qpoly: rounds*3 F (a,b,c); r: rounds F; vr: 2 F; fin: 1 F
void ref_sumcheck2(const uint64_t *v1, const uint64_t *v2, size_t n, const uint64_t *prev_r, uint64_t *qpoly, uint64_t *r, uint64_t *vr, uint64_t *fin) {
    vector<F> a = vecF(v1, n), b = vecF(v2, n); double vt = 0, ps = 0;
    proof P = generate_2product_sumcheck_proof(a, b, ldF(prev_r), vt, ps);
    for (size_t i = 0; i < P.q_poly.size(); i++) { stF(qpoly + 6 * i, P.q_poly[i].a); stF(qpoly + 6 * i + 2, P.q_poly[i].b); stF(qpoly + 6 * i + 4, P.q_poly[i].c); }
    for (size_t i = 0; i < P.randomness[0].size(); i++) stF(r + 2 * i, P.randomness[0][i]);
    stF(vr, P.vr[0]); stF(vr + 2, P.vr[1]); stF(fin, P.final_rand);
}
// cpoly: rounds*4 F (a,b,c,d); vr: 3 F. Inputs are destroyed in place by the reference; we copy.
void ref_sumcheck3(const uint64_t *v1, const uint64_t *v2, const uint64_t *v3, size_t n, const uint64_t *prev_r, uint64_t *cpoly, uint64_t *r, uint64_t *vr, uint64_t *fin) {
    vector<F> a = vecF(v1, n), b = vecF(v2, n), c = vecF(v3, n); double vt = 0, ps = 0;
    proof P = _generate_3product_sumcheck_proof(a, b, c, ldF(prev_r), vt, ps);
    for (size_t i = 0; i < P.c_poly.size(); i++) { stF(cpoly + 8 * i, P.c_poly[i].a); stF(cpoly + 8 * i + 2, P.c_poly[i].b); stF(cpoly + 8 * i + 4, P.c_poly[i].c); stF(cpoly + 8 * i + 6, P.c_poly[i].d); }
    for (size_t i = 0; i < P.randomness[0].size(); i++) stF(r + 2 * i, P.randomness[0][i]);
    stF(vr, P.vr[0]); stF(vr + 2, P.vr[1]); stF(vr + 4, P.vr[2]); stF(fin, P.final_rand);
}


// prove_gate_consistency_standard (src/sumcheck.cpp:434-501): the in-memory degree-4 gate sumcheck (a0 = a1 = a2 = 1, a3 = -1, transcript
// seeded with F(213), claimed sum 0).  It returns nothing; its inputs are folded in place, so element 0 of each table afterwards is a
// function of every round's challenge, i.e. of every round polynomial.  out: arr_L[0], arr_R[0], arr_O[0], add_gate[0].
void ref_gate_standard(const uint64_t *L, const uint64_t *R, const uint64_t *O, const uint64_t *add, size_t n, const uint64_t *r, int k, uint64_t *out) {
    vector<F> l = vecF(L, n), rr = vecF(R, n), o = vecF(O, n), a = vecF(add, n), rv = vecF(r, k);
    double vt = 0, ps = 0;
    prove_gate_consistency_standard(l, rr, o, a, rv, vt, ps);
    stF(out, l[0]); stF(out + 2, rr[0]); stF(out + 4, o[0]); stF(out + 6, a[0]);
}

// ---- code-membership / FFT-as-sumcheck helpers (src/sumcheck.cpp:2888-2929, 2975-3027, 3223-3235;
//      src/utils.cpp:694-775) -------------------------------------------------------------------------
long long ref_evaluate_parity_matrix(const uint64_t *beta, size_t size_a, long long n, uint64_t *A) {
    vector<F> a(size_a, F(0)), b = vecF(beta, size_a);
    int lvl = 0;
    long long len = evaluate_parity_matrix(a, b, 0, (int)n, 0, lvl);
    memcpy(A, a.data(), 16 * size_a);
    return len;
}
void ref_phi_g_init(const uint64_t *rx, int n, const uint64_t *scale, int is_ifft, uint64_t *phi_g) {
    vector<F> r = vecF(rx, n), g((size_t)1 << n, F(0));
    phiGInit(g, r.begin(), ldF(scale), n, is_ifft != 0);
    memcpy(phi_g, g.data(), 16 * g.size());
}
void ref_prepare_matrix(const uint64_t *M, size_t rows, size_t cols, const uint64_t *r, int k, uint64_t *out) {
    vector<vector<F>> m(rows);
    for (size_t i = 0; i < rows; i++) m[i] = vecF(M + 2 * i * cols, cols);
    vector<F> v = prepare_matrix(m, vecF(r, k));
    memcpy(out, v.data(), 16 * rows);
}
static void dump_proof2(const proof &P, uint64_t *qpoly, uint64_t *r, uint64_t *vr, uint64_t *fin) {
    for (size_t i = 0; i < P.q_poly.size(); i++) { stF(qpoly + 6 * i, P.q_poly[i].a); stF(qpoly + 6 * i + 2, P.q_poly[i].b); stF(qpoly + 6 * i + 4, P.q_poly[i].c); }
    for (size_t i = 0; i < P.q_poly.size(); i++) stF(r + 2 * i, i < P.randomness[0].size() ? P.randomness[0][i] : F(0));
    stF(vr, P.vr[0]); stF(vr + 2, P.vr[1]); stF(fin, P.final_rand);
}
// prove_linear_code draws r1 = generate_randomness(log2 size) itself: we seed the libc generator so that
// the caller can reproduce r1 (returned in r1_out), then hand back the sumcheck transcript.
void ref_prove_linear_code(const uint64_t *codeword, size_t size, long long n, unsigned seed, uint64_t *r1_out, uint64_t *qpoly, uint64_t *r, uint64_t *vr, uint64_t *fin) {
    vector<F> cw = vecF(codeword, size); double vt = 0, ps = 0;
    srandom(seed);
    proof P = prove_linear_code(cw, (int)n, vt, ps);
    memcpy(r1_out, P.randomness[1].data(), 16 * P.randomness[1].size());
    dump_proof2(P, qpoly, r, vr, fin);
}
void ref_prove_fft(const uint64_t *m, size_t s, const uint64_t *rr, const uint64_t *prev_sum, uint64_t *qpoly, uint64_t *r, uint64_t *vr, uint64_t *fin) {
    vector<F> mm = vecF(m, s); int k = (int)log2((double)(2 * s)); double vt = 0, ps = 0;
    proof P = prove_fft(mm, vecF(rr, k), ldF(prev_sum), vt, ps);
    P.randomness[0].push_back(F(0));   // prove_fft pops the last challenge; keep the dump shape fixed
    dump_proof2(P, qpoly, r, vr, fin);
}
void ref_prove_fft_matrix(const uint64_t *M, size_t rows, size_t cols, const uint64_t *rr, const uint64_t *prev_sum, uint64_t *qpoly, uint64_t *r, uint64_t *vr, uint64_t *fin) {
    vector<vector<F>> m(rows);
    for (size_t i = 0; i < rows; i++) m[i] = vecF(M + 2 * i * cols, cols);
    int k = (int)log2((double)(2 * cols)) + (int)log2((double)rows); double vt = 0, ps = 0;
    proof P = prove_fft_matrix(m, vecF(rr, k), ldF(prev_sum), vt, ps);
    dump_proof2(P, qpoly, r, vr, fin);
}

// ---- inner PCS commitments (src/Virgo.cpp:104-178) ------------------------------------------------------------
size_t ref_shockwave_commit(const uint64_t *poly, size_t N, int k, uint64_t *enc_out, uint8_t *levels_out) {
    vector<F> p = vecF(poly, N);
    shockwave_data *d = shockwave_commit(p, k);
    size_t W = 2 * N / k;
    for (int i = 0; i < k; i++) memcpy(enc_out + 2 * (size_t)i * W, d->encoded_matrix[i], 16 * W);
    size_t cnt = flatten_levels(d->MT, levels_out);
    delete d;
    return cnt;
}
void ref_change_form(uint64_t *poly, int logn) { vector<F> p = vecF(poly, (size_t)1 << logn); change_form(p, logn, 0, 0); memcpy(poly, p.data(), 16 * p.size()); }
size_t ref_whir_commit(const uint64_t *poly, size_t N, uint64_t *com_out, uint8_t *levels_out) {
    vector<F> p = vecF(poly, N); Whir_data W;
    whir_commit(p, W);
    memcpy(com_out, W.poly_com.data(), 16 * W.poly_com.size());
    return flatten_levels(W.MT, levels_out);
}

// ---- batch_3product_sumcheck (src/sumcheck.cpp:275-372) and prove_multiplication_tree_new (:35-257) ----
int ref_batch_3product_sumcheck(uint64_t *t1, uint64_t *t2, uint64_t *t3, const size_t *lens, int batches, const uint64_t *a, uint64_t *cpoly, uint64_t *r_out, uint64_t *vr) {
    vector<vector<F>> A1(batches), A2(batches), A3(batches); size_t o = 0;
    for (int j = 0; j < batches; j++) { A1[j] = vecF(t1 + 2 * o, lens[j]); A2[j] = vecF(t2 + 2 * o, lens[j]); A3[j] = vecF(t3 + 2 * o, lens[j]); o += lens[j]; }
    double vt = 0, ps = 0;
    proof P = batch_3product_sumcheck(A1, A2, A3, vecF(a, batches), vt, ps);
    for (size_t i = 0; i < P.c_poly.size(); i++) { stF(cpoly + 8 * i, P.c_poly[i].a); stF(cpoly + 8 * i + 2, P.c_poly[i].b); stF(cpoly + 8 * i + 4, P.c_poly[i].c); stF(cpoly + 8 * i + 6, P.c_poly[i].d); }
    for (size_t i = 0; i < P.randomness[0].size(); i++) stF(r_out + 2 * i, P.randomness[0][i]);
    memcpy(vr, P.vr.data(), 16 * P.vr.size());
    return (int)P.c_poly.size();
}
int ref_mul_tree(const uint64_t *input, size_t vectors, size_t size, const uint64_t *previous_r, const uint64_t *prev_x, uint64_t *cpoly, uint64_t *r_out, uint64_t *vr,
                 uint64_t *fin, uint64_t *final_r, uint64_t *out_eval, uint64_t *final_eval) {
    vector<vector<F>> in(vectors);
    for (size_t j = 0; j < vectors; j++) in[j] = vecF(input + 2 * j * size, size);
    vector<F> px; if (prev_x) px = vecF(prev_x, (size_t)log2((double)vectors));
    double vt = 0, ps = 0;
    mul_tree_proof MP = prove_multiplication_tree_new(in, ldF(previous_r), px, vt, ps);
    size_t qo = 0, ro = 0;
    for (size_t l = 0; l < MP.proofs.size(); l++) {
        proof &P = MP.proofs[l];
        for (size_t i = 0; i < P.c_poly.size(); i++) { stF(cpoly + 2 * (qo + 4 * i), P.c_poly[i].a); stF(cpoly + 2 * (qo + 4 * i + 1), P.c_poly[i].b); stF(cpoly + 2 * (qo + 4 * i + 2), P.c_poly[i].c); stF(cpoly + 2 * (qo + 4 * i + 3), P.c_poly[i].d); }
        for (size_t i = 0; i < P.randomness[0].size(); i++) stF(r_out + 2 * (ro + i), P.randomness[0][i]);
        stF(vr + 6 * l, P.vr[0]); stF(vr + 6 * l + 2, P.vr[1]); stF(vr + 6 * l + 4, P.vr[2]); stF(fin + 2 * l, P.final_rand);
        qo += 4 * P.c_poly.size(); ro += P.randomness[0].size();
    }
    memcpy(final_r, MP.final_r.data(), 16 * MP.final_r.size());
    stF(out_eval, MP.out_eval); stF(final_eval, MP.final_eval);
    return (int)MP.proofs.size();
}

// the reference's own globals has_lookups / lookup_rand (src/main.cpp:67,70): lookup gate maps of compute{3,4}p_error_terms
void ref_set_lookups(int on, const uint64_t *lr) {
    has_lookups = on != 0;
    if (on) { lookup_rand.assign(4, F(0)); lookup_rand[0] = ldF(lr); lookup_rand[1] = ldF(lr + 2); }
}
// ---- streaming-sumcheck error terms (src/sumcheck.cpp:374-432) and batch_prod (:1093-1136)
void ref_err2p(const uint64_t *b1, const uint64_t *b2, const uint64_t *f1, const uint64_t *f2, size_t n, uint64_t *K) {
    vector<F> B1 = vecF(b1, n), B2 = vecF(b2, n), F1 = vecF(f1, n), F2 = vecF(f2, n);
    F K1 = ldF(K), K2 = ldF(K + 2);
    compute2p_error_terms(B1, B2, F1, F2, K1, K2);
    stF(K, K1); stF(K + 2, K2);
}
void ref_err3p(const uint64_t *b1, const int32_t *b2, const uint64_t *f1, const uint64_t *f2, const uint64_t *f3, const uint64_t *beta, size_t n, uint64_t *K) {
    vector<F> B1 = vecF(b1, n), F1 = vecF(f1, n), F2 = vecF(f2, n), F3 = vecF(f3, n), Bt = vecF(beta, n);
    vector<int> B2(b2, b2 + n);
    F K1 = ldF(K), K2 = ldF(K + 2), K3 = ldF(K + 4);
    compute3p_error_terms(B1, B2, F1, F2, F3, Bt, K1, K2, K3);
    stF(K, K1); stF(K + 2, K2); stF(K + 4, K3);
}
void ref_err4p(const uint64_t *b1, const uint64_t *b2, const uint64_t *b3, const int32_t *b4, const uint64_t *f1, const uint64_t *f2, const uint64_t *f3,
               const uint64_t *f4, size_t n, uint64_t *K) {
    vector<F> B1 = vecF(b1, n), B2 = vecF(b2, n), B3 = vecF(b3, n), F1 = vecF(f1, n), F2 = vecF(f2, n), F3 = vecF(f3, n), F4 = vecF(f4, n);
    vector<int> B4(b4, b4 + n);
    F K1 = ldF(K), K2 = ldF(K + 2), K3 = ldF(K + 4), K4 = ldF(K + 6);
    compute4p_error_terms(B1, B2, B3, B4, F1, F2, F3, F4, K1, K2, K3, K4);
    stF(K, K1); stF(K + 2, K2); stF(K + 4, K3); stF(K + 6, K4);
}
// one batch_prod step with `batches` table triples of n elements each (flat [batches][n]); R has one entry on input.
// outputs: new challenge, Kf, K_partial[batches], folded tables in place.
void ref_batch_prod(uint64_t *f1, uint64_t *f2, uint64_t *f3, const uint64_t *b1, const uint64_t *b2, const uint64_t *b3, int batches, size_t n,
                    const uint64_t *r_last, const uint64_t *a, const uint64_t *rem_beta, uint64_t *Kf_io, uint64_t *Kp_io, uint64_t *rand_out) {
    auto split = [&](const uint64_t *p) { vector<vector<F>> v(batches); for (int j = 0; j < batches; j++) v[j] = vecF(p + 2 * (size_t)j * n, n); return v; };
    vector<vector<F>> F1 = split(f1), F2 = split(f2), F3 = split(f3), B1 = split(b1), B2 = split(b2), B3 = split(b3);
    vector<F> R = {ldF(r_last)}, Kp = vecF(Kp_io, batches), A = vecF(a, batches);
    vector<vector<F>> rb(batches); for (int j = 0; j < batches; j++) rb[j] = {ldF(rem_beta + 2 * j)};
    F Kf = ldF(Kf_io); double vt = 0, ps = 0;
    batch_prod(F1, F2, F3, B1, B2, B3, batches, R, Kf, Kp, rb, A, 0, vt, ps);
    stF(Kf_io, Kf); stF(rand_out, R.back());
    memcpy(Kp_io, Kp.data(), 16 * (size_t)batches);
    for (int j = 0; j < batches; j++) { memcpy(f1 + 2 * (size_t)j * n, F1[j].data(), 16 * n); memcpy(f2 + 2 * (size_t)j * n, F2[j].data(), 16 * n); memcpy(f3 + 2 * (size_t)j * n, F3[j].data(), 16 * n); }
}

// ---- Elastic_PC streaming commit (src/Elastic_PC.cpp:174-285) on the synthetic "test" stream ----
// opt 1: RSxRS (trs = B/2^11); opt 2: RS x expander (trs = B/2^14, graphs drawn here).
size_t ref_elastic_commit(size_t N, size_t B, int opt, uint8_t *levels_out) {
    BUFFER_SPACE = B;
    if (opt == 1) { linear_time = false; tensor_row_size = (int)(B >> 11); }
    else { linear_time = true; tensor_row_size = (int)(B >> 14); expander_init_store(tensor_row_size); }
    stream_descriptor fd; fd.name = "test"; fd.size = N; fd.pos = 0;
    _hash comm; vector<vector<_hash>> MT;
    commit(fd, comm, MT);
    return flatten_levels(MT, levels_out);
}
void ref_read_stream_pc(size_t N, size_t B, size_t chunk_idx, uint64_t *out) {
    stream_descriptor fd; fd.name = "test"; fd.size = N; fd.pos = 0;
    vector<F> buf(B);
    for (size_t i = 0; i <= chunk_idx; i++) read_stream_PC(fd, buf.data(), (int)B);
    memcpy(out, buf.data(), 16 * B);
}


// ---- Elastic_PC open, RS x RS (opt 1): the two stream passes that do not reach SHA3 (src/Elastic_PC.cpp:316-347, 487-533, 59-111) ----
// read_stream's default branch (src/witness_stream.cpp:2348-2352) serves the "test" descriptor: v[i] = F(i % 1024 + 1), every chunk alike.
void ref_read_stream(size_t B, uint64_t *out) {
    stream_descriptor fd; fd.name = "test"; fd.size = B; fd.pos = 0;
    vector<F> buf(B); read_stream(fd, buf, (int)B);
    memcpy(out, buf.data(), 16 * B);
}
// aggregate() with linear_time = false: aggr[j] = sum_i beta[i] * chunk_i[j], then C_f = shockwave_commit(aggr, 32) (root handed back).
void ref_elastic_aggregate(size_t N, size_t B, const uint64_t *beta, uint64_t *aggr_out, uint8_t *cf_root) {
    BUFFER_SPACE = B; linear_time = false; tensor_row_size = (int)(B >> 11);
    stream_descriptor fd; fd.name = "test"; fd.size = N; fd.pos = 0;
    size_t K = N / B;
    vector<F> b = vecF(beta, K), rv(K, F(1)), aggr; vector<vector<F>> at; vector<vector<_hash>> MT;
    aggregate(fd, b, rv, MT, aggr, at);
    memcpy(aggr_out, aggr.data(), 16 * aggr.size());
    memcpy(cf_root, C_f->MT.back()[0].arr, 32);
    delete C_f; C_f = nullptr;
}
// compute_aggregation_reply() with linear_time = false (update_reply): Iq = nq x (col, row); reply: nq x (N/B) (every chunk of the
// "test" stream is non-zero, so no chunk is skipped).
void ref_elastic_reply(size_t N, size_t B, const uint64_t *Iq, size_t nq, uint64_t *reply) {
    BUFFER_SPACE = B; linear_time = false; tensor_row_size = (int)(B >> 11); aggregation_queries = (int)nq;
    stream_descriptor fd; fd.name = "test"; fd.size = N; fd.pos = 0;
    vector<vector<size_t>> II(nq); for (size_t q = 0; q < nq; q++) II[q] = {(size_t)Iq[2 * q], (size_t)Iq[2 * q + 1]};
    vector<vector<F>> r;
    compute_aggregation_reply(fd, II, r);
    size_t K = N / B;
    for (size_t q = 0; q < nq; q++) memcpy(reply + 2 * q * K, r[q].data(), 16 * r[q].size());
}

// ---- Elastic_PC open, RS x expander (opt 2): aggregate()'s linear_time branch (src/Elastic_PC.cpp:348-413) and compute_aggregation_reply ->
// update_reply_spielman (:431-485).  Neither reaches SHA3.  The caller has run ref_expander_init_store(B >> 14) (test_Elastic_PC option 2, :767).
// Iq = nq x (col, row) as open() draws them (:650-655).  Returns the number of "remaining" columns (those with a queried parity row);
// aux: their expander codewords, nr x 2trs; tensor_out (nullable): aggregated_tensor, trs x 2B/trs.
size_t ref_elastic_aggregate2(size_t N, size_t B, const uint64_t *beta, const uint64_t *Iq, size_t nq, uint64_t *aggr_out, uint8_t *cf_root, uint8_t *cc_root,
                              uint64_t *aux_out, uint64_t *tensor_out) {
    BUFFER_SPACE = B; linear_time = true; tensor_row_size = (int)(B >> 14); aggregation_queries = (int)nq;
    I.assign(nq, vector<size_t>()); for (size_t q = 0; q < nq; q++) I[q] = {(size_t)Iq[2 * q], (size_t)Iq[2 * q + 1]};
    stream_descriptor fd; fd.name = "test"; fd.size = N; fd.pos = 0;
    size_t K = N / B;
    vector<F> b = vecF(beta, K), rv(K, F(1)), aggr; vector<vector<F>> at; vector<vector<_hash>> MT;
    aggregate(fd, b, rv, MT, aggr, at);
    memcpy(aggr_out, aggr.data(), 16 * aggr.size());
    memcpy(cf_root, C_f->MT.back()[0].arr, 32);
    memcpy(cc_root, C_c->MT.back()[0].arr, 32);
    size_t nr = aux_commit.size();
    if (aux_out) for (size_t i = 0; i < nr; i++) memcpy(aux_out + 2 * i * aux_commit[i].size(), aux_commit[i].data(), 16 * aux_commit[i].size());
    if (tensor_out) for (size_t i = 0; i < at.size(); i++) memcpy(tensor_out + 2 * i * at[i].size(), at[i].data(), 16 * at[i].size());
    delete C_f; C_f = nullptr; delete C_c; C_c = nullptr;
    return nr;
}
void ref_elastic_reply2(size_t N, size_t B, const uint64_t *Iq, size_t nq, uint64_t *reply) {
    BUFFER_SPACE = B; linear_time = true; tensor_row_size = (int)(B >> 14); aggregation_queries = (int)nq;
    stream_descriptor fd; fd.name = "test"; fd.size = N; fd.pos = 0;
    vector<vector<size_t>> II(nq); for (size_t q = 0; q < nq; q++) II[q] = {(size_t)Iq[2 * q], (size_t)Iq[2 * q + 1]};
    vector<vector<F>> r;
    compute_aggregation_reply(fd, II, r);
    size_t K = N / B;
    for (size_t q = 0; q < nq; q++) memcpy(reply + 2 * q * K, r[q].data(), 16 * r[q].size());
}


// ---- streaming multiplication-tree prover on the "test" stream (read_stream's default branch) -------------------------------
void ref_read_mul_tree_layer(size_t fd_size, size_t size, int layer, uint64_t *out) {                 // src/witness_stream.cpp:2413-2456
    stream_descriptor fd; fd.name = "test"; fd.size = fd_size; fd.pos = 0;
    vector<F> v(size); read_mul_tree_layer(fd, v, (int)size, layer);
    memcpy(out, v.data(), 16 * size);
}
void ref_read_mul_tree_data(size_t fd_size, size_t size, int layer, int distance, int batches, uint64_t *out) {   // :2458-2510
    stream_descriptor fd; fd.name = "test"; fd.size = fd_size; fd.pos = 0;
    vector<vector<F>> V(batches); for (int i = 0; i < batches; i++) V[i].resize(size >> (i * distance));
    read_mul_tree_data(fd, V, (int)size, layer, distance);
    size_t o = 0; for (int i = 0; i < batches; i++) { memcpy(out + 2 * o, V[i].data(), 16 * V[i].size()); o += V[i].size(); }
}
void ref_generate_claims_opt(size_t fd_size, size_t B, const uint64_t *r, int rlen, int batches, int layer_id, int distance, uint64_t *claims) {   // src/sumcheck.cpp:1014-1054
    BUFFER_SPACE = B;
    stream_descriptor fd; fd.name = "test"; fd.size = fd_size; fd.pos = 0;
    vector<F> c; generate_claims_opt(fd, vecF(r, rlen), c, batches, layer_id, distance);
    memcpy(claims, c.data(), 16 * c.size());
}
// generate_3product_sumcheck_beta_stream_batch_optimized (src/sumcheck.cpp:1150-1393): draws a, b and the pad challenge from libc
// (seed first); hands back the new claims and new_r rows (row i: 1 + (log2 B - i*distance) + log2(size/2B) entries at stride ld).
// Exits the process if one of its own checks ("Error in sumcheck 1/2") fails.
void ref_sumcheck3_stream_batch(size_t fd_size, size_t B, const uint64_t *r, int rlen, int batches, int distance, int layer_id, const uint64_t *old_claims, int n_old,
                                uint64_t *new_claims, uint64_t *new_r, int ld) {
    BUFFER_SPACE = B;
    stream_descriptor fd; fd.name = "test"; fd.size = fd_size; fd.pos = 0;
    vector<vector<F>> rr(batches), nr; for (int i = 0; i < batches; i++) rr[i] = vecF(r + 2 * (size_t)i * rlen, rlen);
    vector<F> nc(batches, F(0)); double vt = 0, ps = 0;
    generate_3product_sumcheck_beta_stream_batch_optimized(fd, rr, batches, distance, layer_id, vecF(old_claims, n_old), nc, nr, vt, ps);
    memcpy(new_claims, nc.data(), 16 * nc.size());
    for (size_t i = 0; i < nr.size(); i++) memcpy(new_r + 2 * i * ld, nr[i].data(), 16 * nr[i].size());
}
// prove_multiplication_tree_stream_shallow (src/sumcheck.cpp:1746-1915), naive = true (no commit_layers / open_layers: those end in
// SHA3): returns P1.output (the `vectors` products).  Everything after the in-memory tree is self-checked inside (exit(-1)).
int ref_mul_tree_stream_shallow(size_t fd_size, size_t B, int vectors, size_t size, const uint64_t *previous_r, int distance, const uint64_t *prev_x, int nx, uint64_t *out) {
    BUFFER_SPACE = B;
    stream_descriptor fd; fd.name = "test"; fd.size = fd_size; fd.pos = 0;
    double vt = 0, ps = 0;
    vector<F> o = prove_multiplication_tree_stream_shallow(fd, vectors, (int)size, ldF(previous_r), distance, vecF(prev_x, nx), true, vt, ps);
    memcpy(out, o.data(), 16 * o.size());
    return (int)o.size();
}

// commit_layers (src/sumcheck.cpp:983-1003): Elastic_PC commitments (RS x RS) to the "PC_layer" streams of the batched multiplication-tree
// prover; roots_out: (batches - 1) x 32 B (zero where the reference skips a layer that fits one buffer), sizes_out / layers_out likewise
int ref_commit_layers(size_t fd_size, size_t B, int batches, int layer_id, int distance, uint8_t *roots_out, uint64_t *sizes_out, uint64_t *layers_out) {
    BUFFER_SPACE = B;
    stream_descriptor fd; fd.name = "test"; fd.size = fd_size; fd.pos = 0;
    vector<stream_descriptor> fc; vector<vector<vector<_hash>>> MT;
    commit_layers(fd, fc, MT, batches, layer_id, distance);
    for (size_t i = 0; i < fc.size(); i++) {
        sizes_out[i] = fc[i].size; layers_out[i] = fc[i].layer;
        if (!MT[i].empty()) memcpy(roots_out + 32 * i, MT[i].back()[0].arr, 32); else memset(roots_out + 32 * i, 0, 32);
    }
    return (int)fc.size();
}

// whole-driver timing hook for bench.py's cpu_baseline ("reference" kind): commit only.
double ref_time_commit_standard(size_t N, int K) {
    srandom(1);
    vector<F> poly = generate_randomness((int)N);
    linear_time = true; tensor_row_size = (int)(N / ((size_t)K << 11));
    __encode_initialized = false;
    expander_init_store(tensor_row_size);
    vector<vector<_hash>> MT; vector<vector<vector<F>>> T; _hash comm;
    struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
    commit_standard(poly, comm, MT, T, K);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
}

}  // extern "C"
