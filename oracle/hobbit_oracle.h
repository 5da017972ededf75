/* oracle/hobbit_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference's Our_PC / sumcheck hot path (see hobbit_oracle.c for
 * the per-function reference citations).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product path (libhobbit_hip.so) never does.
 *
 * Parity status: PINNED -- every function below is checked bit-exactly against the real reference
 * (oracle/_ref/libhobbit_ref.so, built from /root/reference by oracle/Makefile) in
 * tests/test_oracle_vs_ref.py (runs where /root/reference is present) and against the golden
 * vectors that library produced (tests/golden/, tests/test_oracle_golden.py; runs everywhere).
 *
 * F = uint64_t[2] {real, img} in F_{p^2}, p = 2^61-1, i^2 = -1; hashes are uint8_t[32].
 */
#ifndef HOBBIT_ORACLE_H
#define HOBBIT_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t re, im; } oF;

/* field */
void orc_f_add(const oF *a, const oF *b, oF *o, size_t n);
void orc_f_sub(const oF *a, const oF *b, oF *o, size_t n);
void orc_f_mul(const oF *a, const oF *b, oF *o, size_t n);
void orc_f_neg(const oF *a, oF *o, size_t n);
void orc_f_inv(const oF *a, oF *o, size_t n);
void orc_root_of_unity(int logn, oF *o);

/* transcript hash */
void orc_mimc(const oF *x, const oF *k, oF *o, size_t n);

/* BLAKE3 64 B -> 32 B, Merkle-Damgard leaf, trees */
void orc_blake3_64(const uint8_t *in, uint8_t *out, size_t n);
void orc_hash_md(const oF *xyzw, const uint8_t *prev, uint8_t *out, size_t n);
size_t orc_mt_commit_blake(const oF *leafs, size_t N, uint8_t *levels_out);
size_t orc_create_tree_blake(const uint8_t *level0, size_t n, uint8_t *levels_out);
/* levels = flat [level0 (n) | level1 (n/2) | ... | root]; returns depth = log2(n) */
int orc_open_tree_blake(const uint8_t *levels, size_t n_leaves, size_t col, size_t row, size_t columns, uint8_t *path_out);

/* libc RNG (shared glibc generator; default seed 1) */
void orc_rng_reset(void);
void orc_generate_randomness(int n, oF *out);

/* expander graphs: drawn with libc rand()/random() in the reference's call order */
long long orc_expander_init_store(long long n);
long long orc_graph_dims(int dep, int kind, long long *R, int *degree);
void orc_graph_edges(int dep, int kind, long long *nbr, oF *w);
void orc_graph_set_weights(int dep, int kind, const oF *w);
int orc_encode_monolithic(const oF *src, oF *dst, long long n);

/* FFT / eq table / evaluation */
void orc_fft(oF *arr, int logn, int inverse);          /* fresh twiddles (fft(vector&), utils.cpp:467) */
void orc_fft_cached(oF *arr, int logn, int inverse);   /* _fft with the length-keyed twiddle cache (utils.cpp:605) */
void orc_fft_cache_reset(void);
void orc_precompute_beta(const oF *r, int k, oF *out);
void orc_evaluate_vector(const oF *v, size_t n, const oF *r, int k, oF *out);

/* tensor code + Our_PC commit */
void orc_compute_tensorcode(const oF *msg, size_t M, int trs, int lin, oF *out);
size_t orc_commit_standard(const oF *poly, size_t N, int K, int trs, int lin, uint8_t *levels_out, oF *tensor_out);
void orc_aggregate(const oF *poly, size_t N, const oF *beta, int K, oF *aggr_out);

/* sumchecks */
void orc_sumcheck2(const oF *v1, const oF *v2, size_t n, const oF *prev_r, oF *qpoly, oF *r, oF *vr, oF *fin);
void orc_sumcheck3(const oF *v1, const oF *v2, const oF *v3, size_t n, const oF *prev_r, oF *cpoly, oF *r, oF *vr, oF *fin);

/* degree-4 gate-consistency sumcheck (src/sumcheck.cpp:875-929); tables add, beta, L, R, O, mul folded in place */
void orc_gate_claim(oF *const t[6], size_t n, const oF *a, oF *out);
void orc_gate_sumcheck(oF *t0, oF *t1, oF *t2, oF *t3, oF *t4, oF *t5, size_t n, const oF *a, oF *rand_io, oF *sum_io, oF *poly, oF *r, oF *fin, int *check);

/* code-membership / FFT-as-sumcheck helpers */
long long orc_evaluate_parity_matrix(const oF *beta, size_t size_a, long long n, oF *A);
void orc_phi_g_init(const oF *rx, int n, const oF *scale, int is_ifft, oF *phi_g);
void orc_prepare_matrix(const oF *M, size_t rows, size_t cols, const oF *r, int k, oF *out);
void orc_prove_linear_code(const oF *codeword, size_t size, long long n, const oF *r1, oF *qpoly, oF *r, oF *vr, oF *fin);
void orc_prove_fft(const oF *m, size_t s, const oF *rr, oF *qpoly, oF *r, oF *vr, oF *fin);
void orc_prove_fft_matrix(const oF *M, size_t rows, size_t cols, const oF *rr, oF *qpoly, oF *r, oF *vr, oF *fin);

/* inner PCS commitments of the opening */
size_t orc_shockwave_commit(const oF *poly, size_t N, int k, oF *enc_out, uint8_t *levels_out);
void orc_change_form(oF *poly, int logn);
size_t orc_whir_commit(const oF *poly, size_t N, oF *com_out, uint8_t *levels_out);

/* query material of WHIR's _verify_iteration (see hobbit_oracle.c); every pointer may be NULL */
typedef struct { int32_t *qidx; oF *qreply; uint8_t *qpaths; oF *final_pb; int32_t *nq; } orc_whir_queries;
int orc_whir_prove_ex(const oF *poly_in, size_t N, const oF *x, const oF *com, const uint8_t *com_levels, oF *qpoly, oF *a_out, uint8_t *fri_roots,
                      oF *scal, int *checks, orc_whir_queries *Q);
int orc_shockwave_prove_ex(const oF *matrix, const oF *enc, const uint8_t *levels, size_t N, int k, const oF *x, int xlen, uint32_t *I_out, oF *q1, oF *r1o, oF *vr1,
                           oF *fin1, oF *q2, oF *r2o, oF *vr2, oF *fin2, oF *wq, oF *wa, uint8_t *wroots, oF *wscal, int *wchecks, uint8_t *whir_root,
                           oF *reply, uint8_t *paths, orc_whir_queries *Q);
int orc_whir_prove(const oF *poly_in, size_t N, const oF *x, oF *qpoly, oF *a_out, uint8_t *fri_roots, oF *scal, int *checks);
int orc_shockwave_prove(const oF *matrix, const oF *enc, size_t N, int k, const oF *x, int xlen, uint32_t *I_out, oF *q1, oF *r1o, oF *vr1, oF *fin1,
                        oF *q2, oF *r2o, oF *vr2, oF *fin2, oF *wq, oF *wa, uint8_t *wroots, oF *wscal, int *wchecks, uint8_t *whir_root);

/* batched cubic sumcheck and the multiplication-tree prover */
int orc_batch_3product_sumcheck(oF *t1, oF *t2, oF *t3, const size_t *lens, int batches, const oF *a, oF *cpoly, oF *r_out, oF *vr);
int orc_mul_tree(const oF *input, size_t vectors, size_t size, const oF *previous_r_in, const oF *prev_x, oF *cpoly, oF *r_out, oF *vr, oF *fin,
                 oF *final_r, oF *out_eval, oF *final_eval);

/* streaming-sumcheck error terms / folds (K arrays are accumulated into, as in the reference) */
void orc_err2p(const oF *b1, const oF *b2, const oF *f1, const oF *f2, size_t n, oF *K);
void orc_err3p(const oF *b1, const int32_t *b2, const oF *f1, const oF *f2, const oF *f3, const oF *beta, size_t n, oF *K);
void orc_err4p(const oF *b1, const oF *b2, const oF *b3, const int32_t *b4, const oF *f1, const oF *f2, const oF *f3, const oF *f4, size_t n, oF *K);
void orc_batch_prod_terms(const oF *b1, const oF *b2, const oF *b3, const oF *f1, const oF *f2, const oF *f3, size_t n, oF *K);
void orc_fold_axpy(oF *fold, const oF *buff, const oF *rnd, size_t n);
void orc_batch_prod(oF *f1, oF *f2, oF *f3, const oF *b1, const oF *b2, const oF *b3, int batches, size_t n, const oF *r_last, const oF *a,
                    const oF *rem_beta, oF *Kf, oF *Kp, oF *rand_out);

/* Our_PC open without the inner shockwave/WHIR PCS (see hobbit_oracle.c); returns total rounds */
int orc_open_core(const oF *poly, size_t N, int K, int trs, const oF *x, int queries, uint32_t *I_out, oF *reply_out, const oF *tensor,
                  oF *scalars_out, oF *qpoly, oF *r_out, oF *vr, oF *fin, int *checks, uint8_t *roots);

/* the same from a given aggregate vector (the multi-GPU open sums it from per-rank partials) */
int orc_open_core_aggr(const oF *aggr, size_t M, int K, int trs, int queries, uint32_t *I_out, oF *scalars_out, oF *qpoly, oF *r_out, oF *vr, oF *fin, int *checks,
                       uint8_t *roots);

/* Elastic_PC streaming commit on the synthetic "test" stream */
void orc_read_stream_pc(size_t B, oF *out);
size_t orc_elastic_commit(size_t N, size_t B, int opt, uint8_t *levels_out);

/* Elastic_PC open, RS x RS (option 1; option 2's update_reply_spielman is undefined behaviour in the reference, see hobbit_oracle.c) */
void orc_read_stream(size_t B, oF *out);
void orc_elastic_aggregate(size_t N, size_t B, const oF *beta, oF *aggr_out, uint8_t *cf_root);
size_t orc_elastic_reply(size_t N, size_t B, const uint64_t *Iq, size_t nq, oF *reply);
/* Our_PC open_standard with linear_time == false (test_PC option 1) up to shockwave_prove(C_f, rx); tensor: K x 2trs x cols or NULL */
int orc_open_standard_rs(const oF *poly, size_t N, int K, int trs, const oF *x, int queries, const uint8_t *commit_levels, const oF *tensor, uint32_t *I_out, oF *rv0_out,
                         oF *aggr_out, uint8_t *cf_root, oF *reply_out, uint8_t *paths_out, int *ncols_out, oF *qpoly, oF *r_out, oF *vr, oF *fin, int *checks, oF *rx_out);
int orc_elastic_open_rs(size_t N, size_t B, const oF *x, int queries, const uint8_t *commit_levels, uint32_t *I_out, oF *rv0_out, oF *aggr_out, uint8_t *cf_root,
                        oF *reply_out, uint8_t *paths_out, int *ncols_out, oF *qpoly, oF *r_out, oF *vr, oF *fin, int *checks, oF *rx_out);

/* streaming sumcheck drivers (see hobbit_oracle.c) */
size_t orc_elastic_aggregate2(size_t N, size_t B, const oF *beta, const uint64_t *Iq, size_t nq, oF *aggr_out, uint8_t *cf_root, uint8_t *cc_root, oF *aux_out, oF *tensor_out);
size_t orc_elastic_reply2(size_t N, size_t B, const uint64_t *Iq, size_t nq, oF *reply, int stale_parity_quirk);
int orc_elastic_open_spielman(size_t N, size_t B, const oF *x, int queries, const uint8_t *commit_levels, int stale_parity_quirk, uint32_t *I_out, oF *rv0_out,
                              oF *aggr_out, uint8_t *roots, oF *reply_out, uint8_t *paths_out, int *nr_out, oF *aux_out, oF *scal, oF *qpoly, oF *r_out,
                              oF *vr, oF *fin, int *checks, oF *rx_out);
void orc_stream_config(int kind, uint64_t seed);
void orc_read_mul_tree_layer(size_t size, int layer, oF *out);
void orc_read_mul_tree_data(size_t size, int layer, int distance, int batches, oF *out);
int orc_sumcheck3_stream_batch(size_t fd_size, size_t B, const oF *r, int rlen, int batches, int distance, int layer_id, const oF *old_claims, int n_old,
                               oF *new_claims, oF *new_r, int new_r_ld, oF *cpoly1, oF *r1, oF *vr1, oF *qpoly2, oF *r2, oF *vr2, oF *fin2, oF *R_out, int *checks);

void orc_generate_claims_opt(size_t fd_size, size_t B, const oF *r, int batches, int layer_id, int distance, oF *claims);
void orc_gate_consistency_stream(const oF *L, const oF *Rt, const oF *O, const int32_t *S, size_t n_chunks, size_t B, const oF *r, oF *R_out, oF *a_out, oF *poly, oF *gr,
                                 oF *fin6, oF *Peval, oF *b_out, oF *q2, oF *r2, oF *vr2, oF *fin2, int *checks);

void orc_set_lookups(int on, const oF *lr);      /* has_lookups / lookup_rand[0..1] for orc_err3p / orc_err4p */
/* prove_gate_consistency_lookups (src/sumcheck.cpp:503-795); lr = lookup_rand[0..1]; fin9 = add_L, add_R, L, R, O, lkp, lkp_O, mul, beta; checks[5] */
void orc_gate_consistency_lookups_stream(const oF *L, const oF *Rt, const oF *O, const int32_t *S, size_t n_chunks, size_t B, const oF *r, const oF *lr, oF *R_out, oF *a_out,
                                         oF *poly, oF *gr, oF *fin9, oF *Peval, oF *b_out, oF *q2, oF *r2, oF *vr2, oF *fin2, int *checks);

size_t orc_elastic_commit_pc_layer(size_t N, size_t B, int layer, uint8_t *levels_out);
void orc_elastic_group_digests(size_t B, int opt, size_t group, uint8_t *out);
size_t orc_elastic_commit_model(size_t N, size_t B, int opt, uint8_t *levels_out);

/* cpu_baseline helper: generate test_PC's inputs and time commit_standard (seconds) */
double orc_time_commit_standard(size_t N, int K);
/* the same commit on T threads (rows / columns / leaves dealt in ranges inside each chunk); identical levels */
size_t orc_commit_standard_mt(const oF *poly, size_t N, int K, int trs, int lin, int T, uint8_t *levels_out, oF *tensor_out);
double orc_time_commit_standard_mt(size_t N, int K, int T);

#ifdef __cplusplus
}
#endif
#endif
