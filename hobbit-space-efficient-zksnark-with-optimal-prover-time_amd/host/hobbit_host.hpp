// hobbit_host.hpp -- C++ host mirror of the reference's hot-path interface.
//
// The reference is one C++ executable whose "plugin API" for this path is a set of free functions
// and globals.  This header re-declares exactly those names with the reference's
// signatures so that a reference-shaped driver (src/main.cpp, test_PC) compiles against it
// unchanged; every body in hobbit_host.cpp is a thin call into the C ABI (include/hobbit_hip.h).
// Host-only pieces stay on the host exactly as in the reference: the libc rand()/random() draws
// (graphs, generate_randomness, queries) and the printf/exit self-check convention.
//
// Each declaration cites the reference declaration it mirrors.
#pragma once
#include <cstdint>
#include <cstddef>
#include <string>
#include <vector>
#include "hobbit_hip.h"

using std::vector;

namespace virgo {
// src/fieldElement.hpp:20-108 -- same 16-byte layout {real, img}, same operators; the arithmetic is
// the library's (canonical results identical to the reference's, see hobbit_field.hpp)
class fieldElement {
public:
    fieldElement() : real(0), img(0) {}
#ifdef HOBBIT_HOST_REFERENCE_BUILD
    // The reference declares these two (src/fieldElement.hpp:24,44): a user-provided copy constructor makes the class non-trivial for the
    // purposes of calls, so an F passed or returned by value travels through a hidden pointer instead of two registers.  The mirror must
    // agree with the reference's objects on that when both sit in one process (found the hard way: prove_multiplication_tree_stream_shallow
    // takes `F previous_r` by value).
    fieldElement(const fieldElement &b) : real(b.real), img(b.img) {}
    fieldElement &operator=(const fieldElement &b) { real = b.real; img = b.img; return *this; }
#endif
    fieldElement(long long x);
    fieldElement(long long x, long long y);
    fieldElement operator+(const fieldElement &o) const;
    fieldElement operator-(const fieldElement &o) const;
    fieldElement operator-() const;
    fieldElement operator*(const fieldElement &o) const;
    bool operator==(const fieldElement &o) const { return real == o.real && img == o.img; }
    bool operator!=(const fieldElement &o) const { return !(*this == o); }
    fieldElement &operator+=(const fieldElement &o) { *this = *this + o; return *this; }
    fieldElement &operator-=(const fieldElement &o) { *this = *this - o; return *this; }
    fieldElement &operator*=(const fieldElement &o) { *this = *this * o; return *this; }
    fieldElement inv() const;
    static fieldElement zero() { return fieldElement(0); }
    static fieldElement one() { return fieldElement(1); }
    unsigned long long real, img;
};
}  // namespace virgo
#define F virgo::fieldElement          /* src/config_pc.hpp:10-12 */
#define F_ONE virgo::fieldElement::one()
#define F_ZERO virgo::fieldElement::zero()

struct _hash { uint8_t arr[32]; };     /* src/Blake3_hash.h:3-5 */

/* globals that are part of the reference ABI (src/main.cpp:31,38; src/Our_PC.cpp:21; src/Elastic_PC.cpp:10,14) */
extern int tensor_row_size;
extern size_t BUFFER_SPACE;
extern bool linear_time;
extern int aggregation_queries;

/* src/polynomial.h:18-70 (coefficients highest degree first) */
class linear_poly { public: F a, b; linear_poly() {} linear_poly(const F &aa, const F &bb) : a(aa), b(bb) {} F eval(const F &x) const { return a * x + b; } };
class quadratic_poly { public: F a, b, c; quadratic_poly() {} quadratic_poly(const F &aa, const F &bb, const F &cc) : a(aa), b(bb), c(cc) {}
    F eval(const F &x) const { return ((a * x) + b) * x + c; } };
class cubic_poly { public: F a, b, c, d; cubic_poly() {} cubic_poly(const F &aa, const F &bb, const F &cc, const F &dd) : a(aa), b(bb), c(cc), d(dd) {}
    F eval(const F &x) const { return (((a * x) + b) * x + c) * x + d; } };

/* src/sumcheck.h:21-43 (the members the hot-path functions fill) */
#ifndef HOBBIT_HOST_REFERENCE_BUILD
struct proof {
    int type = 0;
    vector<vector<F>> randomness;
    vector<quadratic_poly> q_poly;
    vector<cubic_poly> c_poly;
    vector<F> vr;
    F final_rand;
};
#else
/* in front of the reference's own objects a `proof` (and the mul_tree_proof that holds a vector of them) is returned BY VALUE into storage
 * the reference's code laid out and will destroy: every member of src/sumcheck.h:21-43, in its order */
struct proof {
    int type = 0;
    vector<vector<F>> randomness;
    vector<quadratic_poly> q_poly;
    vector<cubic_poly> c_poly;
    vector<F> output;
    vector<F> vr;
    vector<F> gr;
    vector<F> liu_sum;
    vector<vector<F>> sig;
    vector<vector<F>> final_claims_v;
    F divident, divisor, quotient, remainder;
    vector<vector<vector<F>>> w_hashes;
    vector<F> r, individual_sums;
    vector<vector<F>> Partial_sums;
    vector<quadratic_poly> q_poly1;
    vector<quadratic_poly> q_poly2;
    F final_rand;
    F final_sum;
    int K = 0;
};
#endif

/* src/expanders.h:7-16 */
class graph {
public:
    int degree = 0;
    vector<vector<long long>> neighbor, r_neighbor;
    vector<vector<F>> weight, r_weight;
    long long L = 0, R = 0;
};
extern graph _C[100], D[100];                                   /* src/expander.cpp:2 */

/* device context used by every call below; created on first use on HOBBIT_DEVICE (default 0).
 * Creation failure (no GPU / library) prints the reason and exits(-1), the reference's own error
 * convention -- there is no CPU fallback. */
hobbit_ctx *hobbit_host_ctx();
void hobbit_host_shutdown();

/* src/mimc.h:5-6 */
void init_hash();
F mimc_hash(F input, F k);
/* src/utils.hpp:13,18,23,39,57 */
void precompute_beta(vector<F> r, vector<F> &B);
vector<F> generate_randomness(int size);
F evaluate_vector(vector<F> v, vector<F> r);
void fft(vector<F> &arr, int logn, bool flag);
void _fft(F *arr, int logn, bool flag);
/* src/expanders.h:78 ; src/linear_code_encode.h:62 */
long long expander_init_store(long long n, int dep = 0);
void hobbit_host_upload_graphs(long long n);                    /* device upload of the graphs already in _C / D (drawn by another copy of expander_init_store) */
int encode_monolithic(const F *src, F *dst, long long n, int dep = 0);
/* src/Blake3_hash.h:9 */
void blake3_hash(uint8_t *src, uint8_t *dst);
/* src/merkle_tree.h:25,33,36,38 */
namespace merkle_tree {
_hash hash_double_field_element_merkle_damgard_blake(virgo::fieldElement x, virgo::fieldElement y, virgo::fieldElement z, virgo::fieldElement w, _hash &prev_hash);
namespace merkle_tree_prover {
void MT_commit_Blake(F *leafs, vector<vector<_hash>> &hashes, int N);
void create_tree_blake(int ele_num, vector<vector<_hash>> &hashes, const int element_size = 256 / 8, bool alloc_required = false);
vector<_hash> open_tree_blake(vector<vector<_hash>> &MT_hashes, vector<size_t> c, int collumns);
}  // namespace merkle_tree_prover
}  // namespace merkle_tree
/* src/PC_utils.h:6 */
void compute_tensorcode(vector<F> &message, vector<vector<F>> &tensor);
/* src/Our_PC.hpp:11-15 */
void commit_standard(vector<F> &poly, _hash &comm, vector<vector<_hash>> &MT_hashes, vector<vector<vector<F>>> &_tensor, int K);
void test_PC(size_t N, int option, int K);
void open_standard(vector<F> &poly, vector<F> x, vector<vector<_hash>> &Commitment_MT, vector<vector<vector<F>>> &_tensor, int K, double &vt, double &ps);
/* The reference's open_standard returns nothing but vt / ps (it never serialises its proof); the messages the device prover
 * produced for the last open_standard call are kept here.  Layouts as hobbit_open_out / hobbit_shockwave_out (include/hobbit_hip.h). */
struct hobbit_host_shockwave_transcript {
    vector<uint32_t> I; vector<F> q1, r1, vr1, q2, r2, vr2, wq, wa, wscal, reply, wqreply, wfinal; F fin1, fin2;
    vector<uint8_t> wroots, paths, wqpaths; uint8_t whir_root[32]; vector<int32_t> wqidx, wqn; int wchecks[2] = {0, 0}, iters = 0;
};
struct hobbit_host_open_transcript {
    int queries = 0, rounds = 0;
    vector<uint32_t> cols, rows; vector<F> reply, qpoly, r, vr, fin, scalars; vector<uint8_t> paths; uint8_t roots[64];
    int checks[3] = {0, 0, 0};
    hobbit_host_shockwave_transcript sp_c, sp_f;
};
hobbit_host_open_transcript &hobbit_host_last_open();
/* src/Our_PC.cpp:258-272, 291-305 (file-local helpers of open_standard) */
void _aggregate_axpy(vector<F> &poly, vector<F> beta1, vector<F> &aggregated_vector, int K);
void _compute_aggregation_reply(vector<vector<size_t>> &I, vector<vector<F>> &reply, vector<vector<vector<F>>> &_tensor, int K);
/* src/sumcheck.h:70,78 */
struct proof generate_2product_sumcheck_proof(vector<F> &_v1, vector<F> &_v2, F previous_r, double &vt, double &ps);
struct proof _generate_3product_sumcheck_proof(vector<F> &v1, vector<F> &v2, vector<F> &v3, F previous_r, double &vt, double &ps);

/* src/sumcheck.h:89 (src/sumcheck.cpp:434-501).  The reference folds its four inputs in place and returns nothing; here element 0 of
 * arr_L, arr_R, arr_O, add_gate is set to the fully folded value (all a caller can use) and the rest is left as passed in. */
void prove_gate_consistency_standard(vector<F> &arr_L, vector<F> &arr_R, vector<F> &arr_O, vector<F> &add_gate, vector<F> r, double &vt, double &ps);
/* src/sumcheck.h:67,74,78,81 ; src/utils.hpp:24,41 */
int evaluate_parity_matrix(vector<F> &A, vector<F> &beta1, int Offset, int n, int dep, int &lvl);
proof prove_linear_code(vector<F> &codeword, int n, double &vt, double &ps);
struct proof prove_fft(vector<F> &m, vector<F> r, F previous_sum, double &vt, double &ps);
struct proof prove_fft_matrix(vector<vector<F>> M, vector<F> r, F previous_sum, double &vt, double &ps);
void phiGInit(vector<F> &phi_g, const vector<F>::const_iterator &rx, const F &scale, int n, bool isIFFT);
vector<F> prepare_matrix(vector<vector<F>> M, vector<F> r);
vector<vector<F>> transpose(vector<vector<F>> M);

/* src/sumcheck.h:8-20 */
struct mul_tree_proof {
    F initial_randomness;
    size_t size = 0;
    F in1, in2;
    F out_eval;
    vector<struct proof> proofs;
    vector<F> output;
    vector<F> final_r;
    vector<F> global_randomness, individual_randomness;
    vector<F> partial_eval;
    F final_eval;
};
/* src/sumcheck.h:92, src/sumcheck.cpp:275 (defined there without a header declaration) */
mul_tree_proof prove_multiplication_tree_new(vector<vector<F>> &input, F previous_r, vector<F> prev_x, double &vt, double &ps);
struct proof batch_3product_sumcheck(vector<vector<F>> &arr1, vector<vector<F>> &arr2, vector<vector<F>> &arr3, vector<F> a, double &vt, double &ps);
/* src/PC_utils.h:9 : tensor = 2*tensor_row_size row pointers of 2*size/tensor_row_size F each */
void _compute_tensorcode(F *message, F **tensor, int size);
/* src/Virgo.h:27-47, 95 ; src/Virgo.cpp:120, 435.  The encoded matrix and the column tree stay on the device (d_*); matrix /
 * encoded_matrix are materialised on the host as the reference has them (k rows). */
struct shockwave_data {
    int k = 0;
    int N = 0;
    F **encoded_matrix = nullptr, **matrix = nullptr;
    vector<vector<_hash>> MT;
    void *d_matrix = nullptr, *d_enc = nullptr, *d_levels = nullptr;      /* not in the reference: device residents of the same data */
    ~shockwave_data();
};
shockwave_data *shockwave_commit(vector<F> &poly, int k);
void shockwave_prove(shockwave_data *data, vector<F> x, double &vt, double &ps);
struct hobbit_host_shockwave_transcript;
hobbit_host_shockwave_transcript &hobbit_host_last_shockwave();           /* the messages of the last shockwave_prove (the reference returns none) */
/* src/PC_utils.cpp:6-7 ; src/linear_code_encode.cpp:3-4 ; src/sumcheck.cpp:27,29 : globals of the reference ABI.  C_f / C_c are set by
 * this mirror's shockwave users as the reference's _aggregate / aggregate do; scratch / __encode_initialized exist for source
 * compatibility only (the device encode has no host scratch); routine_time accumulates the time spent re-generating streams, sc_vt is
 * never written on the prover side. */
extern shockwave_data *C_f, *C_c;
extern F *scratch[2][100];
extern bool __encode_initialized;
extern double routine_time, sc_vt;

/* src/witness_stream.h:5-16 (the fields the PCS reads), src/Elastic_PC.hpp:12-16 */
struct stream_descriptor {
    int idx = 0, offset = 0, stage = 0;
    bool finished = false;
    size_t pos = 0, pos_j = 0;
    size_t data_size = 0, row_size = 0, col_size = 0, size = 0, layer = 0, tree_pos = 0;
    std::string name;
    size_t input_pos = 0, input_pos_j = 0, input_data_size = 0, input_size = 0;
    std::string input_name;
};
void read_stream_PC(stream_descriptor &fd, F *v, int size);      /* synthetic default stream only (src/witness_stream.cpp:2405-2411) */
void commit(stream_descriptor fd, _hash &comm, vector<vector<_hash>> &MT_hashes);
void init_commitment(bool mod);
void read_stream(stream_descriptor &fd, vector<F> &v, int size);  /* default branch only (src/witness_stream.cpp:2348-2352) */
void open(stream_descriptor fd, vector<F> x, vector<vector<_hash>> &Commitment_MT, double &vt, double &ps);   /* !linear_time (RS x RS) */
/* src/main.cpp:985-1087: the non-streaming prover.  Everything it proves runs on the device (both commit_standard / open_standard modes,
 * prove_multiplication_tree_new, prove_gate_consistency_standard); what it READS comes from the witness generator
 * (read_trace, src/witness_stream.cpp:1701; read_memory), which is not part of this library: a reference build assigns its own two
 * functions to the hooks, a test assigns synthetic readers.  circuit_size: src/main.cpp global (set by init_stream, :1100). */
extern size_t circuit_size;
extern void (*hobbit_read_trace_hook)(stream_descriptor &fd, vector<F> &buff_L, vector<F> &buff_R, vector<F> &buff_O, vector<int> &buff_S);
extern void (*hobbit_read_memory_hook)(stream_descriptor &fd, vector<F> &buff_addr, vector<F> &buff_value, vector<F> &buff_access);
void reset_stream(stream_descriptor &fd);                      /* src/witness_stream.cpp:228-234 */
/* src/sumcheck.h:90-91, src/sumcheck.cpp:796 / :503: the streaming gate-consistency provers.  The trace is read through
 * hobbit_read_trace_hook (BUFFER_SPACE gates per call); has_lookups / lookup_rand: src/main.cpp:70 / :67 (prove_gate_consistency_lookups
 * reads lookup_rand[0..1] and needs has_lookups set, as compute{3,4}p_error_terms do in the reference). */
extern bool has_lookups;
extern vector<F> lookup_rand;
extern int tensor_code;                                        /* src/Elastic_PC.cpp:14 (1 = the tensor code; the only value the device path implements) */
void prove_gate_consistency(stream_descriptor tr, vector<F> r, double &vt, double &ps);
void prove_gate_consistency_lookups(stream_descriptor tr, vector<F> r, double &vt, double &ps);
/* the messages of the last streaming gate prover (the reference returns none): layouts as hobbit_gate_stream_out / hobbit_gate_lkp_stream_out */
struct hobbit_host_gate_transcript { vector<F> R, a, poly, gr, fin, Peval, b, q2, r2, vr2; F fin2; int checks[5] = {0, 0, 0, 0, 0}; bool lookups = false; };
hobbit_host_gate_transcript &hobbit_host_last_gate();
void test_Elastic_PC(size_t N, int option);                     /* src/Elastic_PC.cpp:736-771: options 1 (commit + open) and 2 (commit; its open is undefined in the reference) */
/* the messages of the last Elastic open (the reference returns only vt / ps); layouts as hobbit_elastic_open_out */
struct hobbit_host_elastic_transcript {
    int queries = 0, rounds = 0, reply_len = 0, ncols = 0;
    vector<uint32_t> cols, rows; vector<F> reply, qpoly, r, vr, fin, rx; vector<uint8_t> paths; uint8_t cf_root[32]; F rv0;
    int checks[2] = {0, 0};
    hobbit_host_shockwave_transcript sp_f;
    /* Elastic_PC::open under linear_time (option 2; recursive_prover_Spielman_stream): remaining columns, padded size of aux_commit, C_c's root,
     * s[0] / s2 / y1, and shockwave_prove(C_c, .); qpoly / r then hold P1, P2, P3, P5 and checks[0] is prove_fft_matrix's test */
    bool lin = false; int nrem = 0; size_t np = 0; uint8_t cc_root[32]; vector<F> scal;
    hobbit_host_shockwave_transcript sp_c;
};
hobbit_host_elastic_transcript &hobbit_host_last_elastic_open();
/* On-wire form of an opening proof.  The reference never serialises a proof (struct proof / mul_tree_proof stay in
 * memory, src/sumcheck.h:8-43), so this layout is ours; little-endian throughout:
 *   header  "HBPF" | u32 version = 1 | u32 kind (1 = open_standard, RS x expander; 2 = RS x RS opening: open_standard with !linear_time or
 *           Elastic_PC::open) | u32 number of sections
 *   section u32 tag | u64 payload bytes | payload   (field elements as 16-byte {real, img}; hashes as 32 bytes; indices as u32 / i32)
 * The nested shockwave_prove transcripts are sections whose payload is itself a sequence of sections.  Unknown tags are skipped by the
 * reader, a truncated or malformed buffer is rejected.  A proof carries messages only -- no checks, no timings. */
std::vector<uint8_t> hobbit_host_serialize_open(const hobbit_host_open_transcript &t);
bool hobbit_host_deserialize_open(const uint8_t *buf, size_t n, hobbit_host_open_transcript &t);
std::vector<uint8_t> hobbit_host_serialize_rs_open(const hobbit_host_elastic_transcript &t);
bool hobbit_host_deserialize_rs_open(const uint8_t *buf, size_t n, hobbit_host_elastic_transcript &t);
void test_Elastic_PC_commit(size_t N, int option);              /* commit phase of test_Elastic_PC (src/Elastic_PC.cpp:736-771) */
/* src/sumcheck.h:84, src/sumcheck.cpp:1150 : the streaming multiplication-tree prover over read_stream (default stream only) */
vector<F> prove_multiplication_tree_stream_shallow(stream_descriptor fd, int vectors, int size, F previous_r, int distance, vector<F> prev_x, bool naive, double &vt, double &ps);
/* src/sumcheck.cpp:983-1011: Elastic_PC commitments to / openings of the "PC_layer" streams of the batched path */
void commit_layers(stream_descriptor fd, vector<stream_descriptor> &fd_com, vector<vector<vector<_hash>>> &MT_hashes, int batches, int layer_id, int distance);
void open_layers(vector<stream_descriptor> &fd_com, vector<vector<vector<_hash>>> &MT_hashes, double &vt, double &ps);
void generate_3product_sumcheck_beta_stream_batch_optimized(stream_descriptor fd, vector<vector<F>> r, int batches, int distance, int layer_id, vector<F> old_claims,
                                                            vector<F> &new_claims, vector<vector<F>> &new_r, double &vt, double &ps);

/* Not in the reference: the 16 GiB `_tensor` of a 2^28 commit stays on the device.
 * commit_standard leaves `_tensor[i]` empty unless HOBBIT_MATERIALIZE_TENSOR=1 (or the tensor is
 * under 256 MiB); the device-resident commitment of the last commit_standard is reachable here
 * and is what _compute_aggregation_reply / open_tree_blake read when `_tensor[i]` is empty. */
hobbit_commitment *hobbit_host_last_commitment();
void hobbit_host_materialize_tensor(vector<vector<vector<F>>> &_tensor);
