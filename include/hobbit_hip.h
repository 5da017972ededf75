/* hobbit_hip.h -- C ABI of libhobbit_hip.so, the MI355X (gfx950) implementation of the HOBBIT
 * prover hot path: Our_PC commit/open building blocks and the in-memory sumchecks.
 *
 * This is the drop-in boundary: every entry point names the reference function (file:line under
 * the reference's src/) whose body it replaces.  The reference has no FFI of its own (one C++
 * executable); the C++ host mirror in <package>/host/hobbit_host.hpp keeps the reference's
 * signatures (commit_standard, generate_2product_sumcheck_proof, ...) and calls this ABI.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ or torch types.
 *   - field elements: uint64_t[2] = {real, img}, canonical, exactly virgo::fieldElement
 *     (src/fieldElement.hpp:96-97).  Hashes: uint8_t[32] = struct _hash (src/Blake3_hash.h:3-5).
 *   - pointers named d_* are DEVICE pointers (hipMalloc / hobbit_malloc / torch data_ptr on the
 *     context's device); pointers named h_* are HOST pointers.
 *   - every call returns 0 on success, a negative HOBBIT_E* code otherwise; hobbit_last_error()
 *     describes the most recent failure on the context.  Nothing falls back to the CPU: without
 *     a HIP device hobbit_ctx_create fails with HOBBIT_ENODEV.
 *   - work is enqueued on the context's stream; calls that return data to h_* pointers
 *     synchronise that stream, all others are asynchronous.
 *   - not re-entrant per context (like the reference: global scratch, global RNG); use one context
 *     per thread / per GPU.
 */
#ifndef HOBBIT_HIP_H
#define HOBBIT_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define HOBBIT_OK 0
#define HOBBIT_ENODEV (-1)   /* no usable HIP device */
#define HOBBIT_EINVAL (-2)   /* bad argument / unsupported shape */
#define HOBBIT_ENOMEM (-3)   /* device allocation failed */
#define HOBBIT_EHIP (-4)     /* HIP runtime error (see hobbit_last_error) */
#define HOBBIT_ESTATE (-5)   /* call order violated (e.g. encode before graph upload) */

typedef struct hobbit_ctx hobbit_ctx;
typedef struct hobbit_commitment hobbit_commitment;
typedef struct hobbit_elastic hobbit_elastic;
typedef struct hobbit_elastic_open hobbit_elastic_open;
typedef struct { uint64_t re, im; } hobbit_F;

/* ---- context, memory, timing ------------------------------------------------------------- */
int hobbit_ctx_create(int device, hobbit_ctx **out);
/* share an existing hipStream_t (e.g. torch's current stream); stream==NULL creates one */
int hobbit_ctx_create_on_stream(int device, void *hip_stream, hobbit_ctx **out);
void hobbit_ctx_destroy(hobbit_ctx *ctx);
const char *hobbit_last_error(const hobbit_ctx *ctx);
const char *hobbit_version(void);
int hobbit_sync(hobbit_ctx *ctx);
int hobbit_malloc(hobbit_ctx *ctx, size_t bytes, void **d_ptr);
int hobbit_free(hobbit_ctx *ctx, void *d_ptr);
int hobbit_memcpy_h2d(hobbit_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int hobbit_memcpy_d2h(hobbit_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
int hobbit_memset(hobbit_ctx *ctx, void *d_dst, int value, size_t bytes);
/* HIP-event timing on the context's stream (bench.py's timed region / roofline) */
int hobbit_timer_begin(hobbit_ctx *ctx);
int hobbit_timer_end_ms(hobbit_ctx *ctx, float *ms);
/* per-kernel profiling: when enabled every kernel launch is bracketed by HIP events on the
 * context's stream; hobbit_profile_get returns total ms and launch count by kernel name. */
int hobbit_profile_enable(hobbit_ctx *ctx, int on);   /* 0 off, 1 every launch, 2 only the commit's bulk kernels */
int hobbit_profile_reset(hobbit_ctx *ctx);
int hobbit_profile_get(hobbit_ctx *ctx, const char *kernel, double *total_ms, long long *launches);
int hobbit_profile_names(hobbit_ctx *ctx, char *buf, size_t buflen); /* ';'-separated */

/* ---- field / transcript (host-side helpers; src/fieldElement.cpp, src/mimc.cpp:95-107) ---- */
void hobbit_mimc(const hobbit_F *x, const hobbit_F *k, hobbit_F *out);           /* mimc_hash */
/* Transcript recorder (test support).  Every Fiat-Shamir hash of the path is mimc_hash (src/mimc.cpp:95-107) computed on the host; while
 * recording is on for the CALLING THREAD, each one this library computes on that thread -- inside any prover entry point and through
 * hobbit_mimc -- appends the record (x, k, result) = 6 x u64.  The sequence is the complete transcript of the provers that ran, in order;
 * tests compare its sha256 with the sequence a call-through recorder in front of the real reference's mimc_hash captured on the same streams
 * (oracle/ref_recorder.cpp, tests/golden/transcripts.json).  record(1) clears and starts, record(0) stops; read copies up to max_records.
 * Entry points that use a second host thread (hobbit_open_standard's shockwave_prove(C_c)) record that thread's hashes on that thread. */
void hobbit_transcript_record(int on);
size_t hobbit_transcript_count(void);
size_t hobbit_transcript_read(uint64_t *out, size_t max_records);
void hobbit_f_mul_host(const hobbit_F *a, const hobbit_F *b, hobbit_F *out, size_t n);
void hobbit_f_inv_host(const hobbit_F *a, hobbit_F *out, size_t n);
/* generate_randomness (src/utils.cpp:873-883), host side, on the process-wide libc generator: every 100 elements c = random(),
 * element = F(c) + F(rand()) */
void hobbit_generate_randomness(size_t n, hobbit_F *h_out);
/* element-wise device ops (test surface for the device field arithmetic): op 0 add, 1 sub, 2 mul */
int hobbit_f_binop(hobbit_ctx *ctx, int op, const hobbit_F *d_a, const hobbit_F *d_b, hobbit_F *d_out, size_t n);

/* ---- expander code (src/expanders.h:20-47,78-92; src/linear_code_encode.h:62-119) ---------- */
/* The host keeps drawing the graphs with libc rand()/random() in the reference's order
 * (expander_init_store); it uploads each level here.  kind 0 = _C[dep], 1 = D[dep].
 * nbr[i*degree+j] in [0,R), w[i*degree+j] any canonical F (32-bit real weights take a fast path). */
int hobbit_graph_reset(hobbit_ctx *ctx);
int hobbit_graph_upload(hobbit_ctx *ctx, int dep, int kind, long long L, long long R, int degree,
                        const long long *h_nbr, const hobbit_F *h_w);
/* call after all levels for code length n are uploaded; returns codeword length via *len */
int hobbit_graph_finalize(hobbit_ctx *ctx, long long n, long long *len);
/* encode_monolithic on `batch` messages; message b at d_src + b*ld_src (n F, contiguous), its
 * codeword written to d_dst + b*ld_dst (2n F: codeword then zeros).  d_src may alias d_dst. */
int hobbit_encode_batch(hobbit_ctx *ctx, const hobbit_F *d_src, hobbit_F *d_dst, long long n,
                        size_t batch, size_t ld_src, size_t ld_dst);

/* ---- FFT (src/utils.cpp:605-673 _fft; natural order in and out) ---------------------------- */
/* in place on `batch` rows of 2^logn F, row b at d_data + b*ld (logn <= 12; forward transforms up to 2^24 on
 * contiguous rows).  inverse!=0 scales by 1/len.
 * Twiddles are always those of the requested direction (the reference's length-keyed cache quirk,
 * SURVEY.md 1, is NOT reproduced here; the host mirror documents where it matters). */
int hobbit_fft_batch(hobbit_ctx *ctx, hobbit_F *d_data, int logn, size_t batch, size_t ld, int inverse);

/* ---- BLAKE3 / Merkle (src/Blake3_hash.cpp:5-10; src/merkle_tree.cpp:62-87,193-221,255-324) -- */
int hobbit_blake3_64(hobbit_ctx *ctx, const uint8_t *d_in, uint8_t *d_out, size_t n);      /* blake3_hash x n */
/* hash_double_field_element_merkle_damgard_blake x n: out[i] = H(H(xyzw[i]) | prev[i]); in place ok */
int hobbit_hash_md(hobbit_ctx *ctx, const hobbit_F *d_xyzw, const uint8_t *d_prev, uint8_t *d_out, size_t n);
/* MT_commit_Blake: leaves H(4 consecutive F) into d_levels[0..N/4), then the tree */
int hobbit_mt_commit_blake(hobbit_ctx *ctx, const hobbit_F *d_leafs, size_t N, uint8_t *d_levels);
/* create_tree_blake: d_levels holds level 0 (n hashes) and receives levels 1.. behind it
 * ((2n-1)*32 bytes in all).  left_left_quirk=1 reproduces the reference (parent = H(left|left),
 * src/merkle_tree.cpp:275-280); 0 builds the conventional H(left|right) tree. */
int hobbit_merkle_levels(hobbit_ctx *ctx, uint8_t *d_levels, size_t n, int left_left_quirk);
/* open_tree_blake: sibling path of leaf pos from flat levels; h_path receives log2(n) hashes */
int hobbit_merkle_path(hobbit_ctx *ctx, const uint8_t *d_levels, size_t n, size_t pos, uint8_t *h_path);
/* batched: h_paths receives nq x log2(n) hashes */
int hobbit_merkle_paths(hobbit_ctx *ctx, const uint8_t *d_levels, size_t n, const uint64_t *h_pos, size_t nq, uint8_t *h_paths);

/* ---- multilinear utilities (src/utils.cpp:251-296 precompute_beta, 789-802 evaluate_vector) -- */
int hobbit_eq_table(hobbit_ctx *ctx, const hobbit_F *h_r, int k, hobbit_F *d_out);
int hobbit_eval_vector(hobbit_ctx *ctx, const hobbit_F *d_v, size_t n, const hobbit_F *h_r, hobbit_F *h_out);

/* ---- tensor code (src/PC_utils.cpp:66-123 compute_tensorcode) ------------------------------ */
/* message M F (row-major trs x M/trs) -> tensor (2 trs) x (2M/trs), stored CODEWORD-MAJOR on the
 * device: element (row, col) at d_out[col*(2*trs) + row].  linear_time!=0: RS rows x expander
 * columns (graphs for n = trs must be finalized); 0: RS x RS.  Row length 2M/trs <= 32768. */
int hobbit_tensorcode(hobbit_ctx *ctx, const hobbit_F *d_msg, size_t M, int trs, int linear_time, hobbit_F *d_out);

/* ---- Our_PC commit (src/Our_PC.cpp:146-171 commit_standard) -------------------------------- */
/* poly: N F on the device.  The commitment object owns the device-resident tensor (K chunks,
 * codeword-major) and Merkle levels.  The call returns once the work is queued on the context's stream (stream semantics, as every
 * device-side entry point here): the host-returning accessors below and hobbit_sync() wait for it; an opening queued on the same
 * context simply runs behind it.  The raw device pointers (levels_dev / tensor_dev) are to be used on this context's stream or after
 * hobbit_sync(). */
int hobbit_commit_standard(hobbit_ctx *ctx, const hobbit_F *d_poly, size_t N, int K, int trs, int linear_time,
                           hobbit_commitment **out);
/* The same for a polynomial that still lives in (pageable) HOST memory -- what the reference's commit_standard(vector<F> &poly, ...) is handed
 * (src/Our_PC.cpp:146).  h_poly is streamed into d_poly (N F of device memory, which the caller keeps for the opening) chunk group by chunk
 * group through two pinned 64 MiB staging buffers owned by the context, and the row FFT of group g starts as soon as group g has landed while
 * group g + 1 is on its way: the upload costs max(PCIe + host copy, device work) instead of their sum.  Same commitment, bit for bit. */
int hobbit_commit_standard_host(hobbit_ctx *ctx, const hobbit_F *h_poly, hobbit_F *d_poly, size_t N, int K, int trs, int linear_time, hobbit_commitment **out);
void hobbit_commitment_free(hobbit_commitment *c);
size_t hobbit_commitment_num_leaves(const hobbit_commitment *c);
const uint8_t *hobbit_commitment_levels_dev(const hobbit_commitment *c);   /* flat levels, device */
/* K x cols x 2trs, device.  (An RS x expander commitment does not write the rows past the codeword's end -- they are zero by construction and
 * every accessor answers them as zeros; this call writes them before it returns the raw pointer, once per commitment.) */
const hobbit_F *hobbit_commitment_tensor_dev(const hobbit_commitment *c);
int hobbit_commitment_levels(hobbit_ctx *ctx, const hobbit_commitment *c, uint8_t *h_levels); /* (2M-1)*32 B */
int hobbit_commitment_root(hobbit_ctx *ctx, const hobbit_commitment *c, uint8_t *h_root);
/* _tensor[chunk][row][0..cols) in the reference's row-major order (lazy materialisation) */
int hobbit_commitment_tensor_row(hobbit_ctx *ctx, const hobbit_commitment *c, int chunk, int row, hobbit_F *h_out);
/* _compute_aggregation_reply (src/Our_PC.cpp:291-305): reply[q*K + i] = _tensor[i][rows[q]][cols[q]] */
int hobbit_commitment_gather(hobbit_ctx *ctx, const hobbit_commitment *c, const uint32_t *h_rows, const uint32_t *h_cols,
                             size_t nq, hobbit_F *h_reply);
/* open_tree_blake for query (col,row): pos = (row/4)*cols + col (src/merkle_tree.cpp:308-324) */
int hobbit_commitment_path(hobbit_ctx *ctx, const hobbit_commitment *c, size_t col, size_t row, uint8_t *h_path);
int hobbit_commitment_paths(hobbit_ctx *ctx, const hobbit_commitment *c, const uint32_t *h_cols, const uint32_t *h_rows, size_t nq,
                            uint8_t *h_paths);

/* ---- Elastic_PC streaming commit (src/Elastic_PC.cpp:174-285 commit) ------------------------- */
/* The stream stays with the host (read_stream_PC); each B-element chunk is pushed as a device
 * buffer.  Every 4th chunk the 4 stored tensor codes are hashed into the 4B running leaves;
 * finish builds the tree: d_levels receives (8B-1)*32 bytes (4B leaves ... root).
 * gcc_arg_order=1 reproduces the reference as built by GCC (Elastic_PC.cpp:238-239 evaluates its
 * call arguments right to left: the first two hash inputs are taken at position+1); 0 takes all
 * four at the same position.  trs = B/2^11 with linear_time=0 (opt 1), B/2^14 with 1 (opt 2). */
int hobbit_elastic_begin(hobbit_ctx *ctx, size_t B, int trs, int linear_time, int gcc_arg_order, hobbit_elastic **out);
int hobbit_elastic_push(hobbit_ctx *ctx, hobbit_elastic *e, const hobbit_F *d_chunk);
int hobbit_elastic_finish(hobbit_ctx *ctx, hobbit_elastic *e, uint8_t *d_levels);
/* multi-GPU form (groups of 4 consecutive chunks per GPU): as hobbit_elastic_push, but the 4th chunk of a group writes
 * the group's 4B inner digests H(c0, c1, c2, t3) (32 B each, the reference's leaf order) to d_digests instead of chaining them into
 * the running leaves; the rank that owns a leaf range chains the digests of all groups (hobbit_chain_digests) and builds its subtree */
int hobbit_elastic_push_inner(hobbit_ctx *ctx, hobbit_elastic *e, const hobbit_F *d_chunk, uint8_t *d_digests);
void hobbit_elastic_free(hobbit_elastic *e);

/* ---- inner PCS commitments of the opening (src/Virgo.cpp:104-178) ------------------------------ */
/* shockwave_commit: poly as k rows of N/k, rows RS-encoded to 2N/k (d_enc: k x 2N/k row-major), column digests
 * (MT_commit_Blake over the k entries of a column) and the tree over them (d_levels: (2*(2N/k)-1) hashes) */
int hobbit_shockwave_commit(hobbit_ctx *ctx, const hobbit_F *d_poly, size_t N, int k, hobbit_F *d_enc, uint8_t *d_levels);
/* change_form (src/Virgo.cpp:104-118), in place on 2^logn elements */
int hobbit_change_form(hobbit_ctx *ctx, hobbit_F *d_poly, int logn);
/* whir_commit: change_form, zero-pad x2, FFT, 16-way regroup (d_com: 2N F), MT_commit_Blake (d_levels: N - 1 hashes... (2N/4)*2-1) */
int hobbit_whir_commit(hobbit_ctx *ctx, const hobbit_F *d_poly, size_t N, hobbit_F *d_com, uint8_t *d_levels);

/* _whir_prove (src/Virgo.cpp:519-686), prover side.  The verifier emulation inside it (fold of the replies,
 * verify_claim_opt_blake = SHA3 accounting) draws nothing and changes no prover state; it is not run -- but the query
 * material _verify_iteration assembles for it (:245-275) IS produced.  libc draws on the host in the reference's order.
 * d_poly: N F (preserved); h_x: log2 N; d_com / d_com_levels: the outputs of hobbit_whir_commit(d_poly) (NULL: round-1
 * queries are drawn but not answered).  All pointers of hobbit_whir_out are host buffers:
 *   qpoly: 3 F per fold round (4 per iteration); a: the fold challenges; fri_roots: 32 B per iteration;
 *   scal: {eval, final sum}; checks[2]: the reference's exit(-1) checks (round sums; final verification), 1 = holds;
 *   iters: number of iterations T.
 *   Query rounds t = 1..T (nullable outputs): qn[t-1] indices each (99, then 100/(3t-2) - 1 ...; the last round uses
 *   100/(1+3T) - 1), answered against the previous layer (t = 1: the commitment, size 2N; t >= 2: FRI layer t-1, size 2N >> (t-1)):
 *   qidx: the indices r; qreply: 16 F per index (layer[r + j*size/16], j < 16); qpaths: open_tree_blake(tree, {r,0}, 0),
 *   log2(size/4) x 32 B per index, back to back; final_pb: final_poly | final_beta (N >> 4T F each, :641-646). */
typedef struct {
    hobbit_F *qpoly, *a; uint8_t *fri_roots; hobbit_F *scal; int *checks; int *iters;
    int32_t *qidx; hobbit_F *qreply; uint8_t *qpaths; hobbit_F *final_pb; int32_t *qn;
} hobbit_whir_out;
int hobbit_whir_prove(hobbit_ctx *ctx, const hobbit_F *d_poly, size_t N, const hobbit_F *d_com, const uint8_t *d_com_levels, const hobbit_F *h_x, hobbit_whir_out *out);
/* shockwave_prove (src/Virgo.cpp:435-517), prover side: row aggregation, whir_commit of the aggregate, 240 libc queries with
 * their replies (column I[i] of the encoded matrix, k F) and paths (open_tree_blake(MT, {I[i],0}, 0), log2(2N/k) x 32 B),
 * P1 = 2-product sumcheck against the query indicator, P2 = prove_fft, then _whir_prove.  d_matrix: k x N/k, d_enc: k x 2N/k,
 * d_levels: the column tree (all as produced by hobbit_shockwave_commit; d_levels NULL: no paths).  All outputs are host
 * buffers (sizes as hobbit_sumcheck2 / hobbit_whir_out; reply, paths and the wq* query fields may be NULL). */
typedef struct {
    uint32_t *I; hobbit_F *q1, *r1, *vr1, *fin1, *q2, *r2, *vr2, *fin2, *wq, *wa; uint8_t *wroots; hobbit_F *wscal; int *wchecks; uint8_t *whir_root; int *iters;
    hobbit_F *reply; uint8_t *paths;
    int32_t *wqidx; hobbit_F *wqreply; uint8_t *wqpaths; hobbit_F *wfinal; int32_t *wqn;
} hobbit_shockwave_out;
int hobbit_shockwave_prove(hobbit_ctx *ctx, const hobbit_F *d_matrix, const hobbit_F *d_enc, const uint8_t *d_levels, size_t N, int k, const hobbit_F *h_x, int xlen,
                           hobbit_shockwave_out *out);

/* ---- multi-GPU commit building blocks (chunk-sharded commit) ------------------ */
/* tensor codes of `nchunks` consecutive messages of M F each (chunk i at d_msg + i*M), outputs
 * codeword-major per chunk at d_out + i*4M */
int hobbit_tensorcode_chunks(hobbit_ctx *ctx, const hobbit_F *d_msg, size_t M, int nchunks, int trs, int linear_time, hobbit_F *d_out);
/* reply[q*nchunks + i] = chunk i of the shard at (row_q, col_q): _compute_aggregation_reply (src/Our_PC.cpp:291-305) on a rank's tensor shard
 * (as written by hobbit_tensorcode_chunks) */
int hobbit_tensor_gather(hobbit_ctx *ctx, const hobbit_F *d_tensor, size_t M, int nchunks, int trs, const uint32_t *h_rows, const uint32_t *h_cols, size_t nq,
                         hobbit_F *h_reply);
/* Multi-GPU open (no reference counterpart: the reference is one process): the per-rank partial aggregates of _aggregate (src/Our_PC.cpp:258-272)
 * are summed across ranks by a 64-bit INTEGER all-reduce -- canonical field components are < 2^61, so eight of them fit -- and brought back into
 * the field here.  n_words 64-bit words at d_words; fold == 0: w += bias (callers shift by -2^60 before the reduction so that a signed sum
 * cannot overflow); fold != 0: w = (w + bias) mod 2^61-1, canonical. */
int hobbit_u64_bias_fold(hobbit_ctx *ctx, void *d_words, size_t n_words, uint64_t bias, int fold);
/* inner leaf digests H(t[4j..4j+3][c]) (src/merkle_tree.cpp:70-75) of nchunks tensors, in leaf
 * order: d_out[(i*M + j*cols + c)*32] */
int hobbit_inner_digests(hobbit_ctx *ctx, const hobbit_F *d_tensor, size_t M, int nchunks, int trs, uint8_t *d_out);
/* Merkle-Damgard chain (src/merkle_tree.cpp:76-86) over K digests per leaf, in place on m leaves:
 * leaf[p] = H(dig_i[p] | leaf[p]) for i = 0..K-1, dig_i at d_digests + i*stride_bytes */
int hobbit_chain_digests(hobbit_ctx *ctx, const uint8_t *d_digests, size_t stride_bytes, int K, size_t m, uint8_t *d_leaves);
/* Multi-GPU commit by chain relay (no counterpart in the single-process reference, whose chain is the loop at
 * src/Our_PC.cpp:162-166): the Merkle-Damgard leaf chain over the `nchunks` chunks of a tensor shard ([chunk][col][2 trs], what
 * hobbit_tensorcode_chunks writes) for the leaf slots [slot_begin, slot_begin + slot_count), slot = col * (trs/2) + j -- the order the shard
 * is read in.  d_state_in: the running state handed over by the rank that holds the preceding chunks (slot_count x 32 B in slot order;
 * NULL = the zero state in front of chunk 0).  d_state_out (slot order, for the next rank) and / or d_leaves (the M-leaf array in the
 * reference's leaf order j * cols + col: on the rank that holds the last chunks) receive the result. */
int hobbit_leaf_chain_relay(hobbit_ctx *ctx, const hobbit_F *d_tensor, size_t M, int nchunks, int trs, int linear_time, size_t slot_begin, size_t slot_count,
                            const uint8_t *d_state_in, uint8_t *d_state_out, uint8_t *d_leaves);
/* Memory / lookup fingerprints of the wiring-consistency streams (src/witness_stream.cpp:2196, 2290-2305, and the same map in
 * prove_circuit_standard, src/main.cpp:1039-1044): d_out[i] = d_addr[i] + 1 + a * d_value[i] + b * d_freq[i]  (d_freq NULL: without the last
 * term).  The three columns come from the witness generator (read_memory); the map itself and everything after it run on the device. */
int hobbit_fingerprint_map(hobbit_ctx *ctx, const hobbit_F *d_addr, const hobbit_F *d_value, const hobbit_F *d_freq, const hobbit_F *h_a, const hobbit_F *h_b, hobbit_F *d_out, size_t n);

/* Verifier side of open_tree_blake (the reference's verify_claim_opt_blake, src/merkle_tree.cpp:326-362, never compares
 * anything): recompute the root from a leaf hash, its position and its `depth` siblings.  quirk_left_left = 1: the reference's tree
 * (parent = H(L | L), src/merkle_tree.cpp:275-280); 0: an ordinary H(L | R) tree.  Returns 1 / 0.  Host only, no context. */
int hobbit_verify_path_host(const uint8_t *leaf, uint64_t pos, const uint8_t *path, int depth, const uint8_t *root, int quirk_left_left);
/* SURVEY.md 8(b) names these three exports; they are thin forms of the calls above and below.
 * hobbit_leaf_chain: the Merkle-Damgard leaf chain (src/Our_PC.cpp:162-166) over `nchunks` chunks of a tensor ([chunk][col][2 trs]) on
 *   top of the M leaves ALREADY in d_leaves (leaf order; zero them for a fresh commitment) -- in-out, as the reference's loop.
 * hobbit_axpy_aggregate: acc[j] += coeff * chunk[j] (_aggregate's inner loop, src/Our_PC.cpp:267-272; Elastic_PC.cpp:330-333).
 * hobbit_stream_fold: kind 2 / 3 / 4 = compute{2,3,4}p_error_terms, 13 = one batch of batch_prod (src/sumcheck.cpp:374-432, 1093-1136);
 *   d_tables in the order of the typed entry points (hobbit_compute2p_error_terms ... hobbit_batch_prod); h_K is accumulated into. */
int hobbit_leaf_chain(hobbit_ctx *ctx, const hobbit_F *d_tensor, size_t M, int nchunks, int trs, int linear_time, uint8_t *d_leaves);
int hobbit_axpy_aggregate(hobbit_ctx *ctx, const hobbit_F *d_chunk, const hobbit_F *h_coeff, hobbit_F *d_acc, size_t n);
int hobbit_stream_fold(hobbit_ctx *ctx, int kind, const hobbit_F *const *d_tables, const int32_t *d_gate, size_t n, hobbit_F *h_K);
/* host-side blake3_hash x n (tree tops of a handful of nodes) */
void hobbit_blake3_64_host(const uint8_t *h_in, uint8_t *h_out, size_t n);

/* ---- Our_PC open building blocks ----------------------------------------------------------- */
/* _aggregate axpy (src/Our_PC.cpp:258-272): d_aggr[j] = sum_i beta[i] * poly[i*M + j], M = N/K */
int hobbit_aggregate(hobbit_ctx *ctx, const hobbit_F *d_poly, size_t N, const hobbit_F *h_beta, int K, hobbit_F *d_aggr);

/* Our_PC open WITHOUT the inner shockwave/WHIR PCS: open_standard (src/Our_PC.cpp:604-661) and
 * recursive_prover_Spielman (src/PC_utils.cpp:271-385) minus the two shockwave_prove calls (shockwave_commit of
 * the aggregate and of its parity half IS included).
 * Draws libc rand()/random() on the host in the reference's order (r_v[0]; 2 x queries; s; r1; s2; a).
 * All output pointers are host buffers supplied by the caller (cols/rows/reply/paths may be NULL):
 *   cols, rows : queries x u32;  reply : queries x K F;  paths : queries x log2(M) x 32 B
 *   qpoly / r  : the five sumcheck transcripts P1..P5 back to back,
 *                rounds = R1, 12, R1+12, R1+12, 12 with R1 = log2(2 trs)  (3 F per round / 1 F per round)
 *   vr : 5 x 2 F;  fin : 5 F;  scalars : r_v[0], s, s2, a, y1
 *   checks[3] : the reference's own consistency checks ("Error recursion 1", "Error recursion 2",
 *               prove_fft_matrix's claimed sum), 1 = holds */
typedef struct {
    uint32_t *cols, *rows; hobbit_F *reply; uint8_t *paths;
    hobbit_F *qpoly, *r, *vr, *fin, *scalars; int *checks;
    uint8_t *roots;      /* 64 B or NULL: roots of _aggregate's inner commitments C_f = shockwave_commit(aggr, 32) and
                            C_c = shockwave_commit(parity half, 32) (src/Our_PC.cpp:274-287); both are always computed */
    hobbit_shockwave_out *sp_c, *sp_f;   /* hobbit_open_standard only: transcripts of shockwave_prove(C_c, .) and shockwave_prove(C_f, .) */
} hobbit_open_out;
int hobbit_open_core(hobbit_ctx *ctx, const hobbit_F *d_poly, size_t N, const hobbit_commitment *c, const hobbit_F *h_x, int queries, hobbit_open_out *out);
/* The whole prover side of open_standard (src/Our_PC.cpp:604-661): hobbit_open_core followed by
 * shockwave_prove(C_c, P4.randomness minus its last entry) (src/PC_utils.cpp:368) and
 * shockwave_prove(C_f, P5.randomness minus its last entry) (:385), C_c over 32 x (trs*cols/32), C_f over 32 x (M/32).
 * out->sp_c / out->sp_f must be set (buffer sizes as hobbit_shockwave_prove).  What the reference runs after that
 * (src/Our_PC.cpp:663-690) is verifier-side accounting whose results are discarded; it is not built. */
int hobbit_open_standard(hobbit_ctx *ctx, const hobbit_F *d_poly, size_t N, const hobbit_commitment *c, const hobbit_F *h_x, int queries, hobbit_open_out *out);
/* The same from a given aggregate vector d_aggr (M F) = sum_i beta[i] * chunk_i (src/Our_PC.cpp:258-272) of a commitment with K chunks and
 * tensor_row_size trs -- the multi-GPU open: each rank aggregates its own chunks, the partials are summed, and everything
 * after that depends on the aggregate alone.  out->cols / rows receive the queries; out->reply and out->paths must be NULL (replies are
 * gathered from the ranks' tensor shards, paths from their subtrees).  The evaluation point is not needed: only its chunk variables enter
 * the opening, through the aggregate. */
int hobbit_open_from_aggregate(hobbit_ctx *ctx, const hobbit_F *d_aggr, size_t M, int K, int trs, int queries, hobbit_open_out *out);

/* ---- Elastic_PC open, RS x RS (src/Elastic_PC.cpp:625-726 open; test_Elastic_PC option 1) ------------------- */
/* The reference streams the data three times: commit, aggregate (:316-347) and compute_aggregation_reply (:487-533).  As in the
 * commit the stream stays with the host and every B-element chunk arrives as a device buffer, in stream order, once per pass.
 *   begin            : beta over the chunk variables (h_x[0 .. log2(N/B))), then the libc draws in the reference's order:
 *                      r_v[0] (generate_randomness(1)), `queries` x (rand() % cols, rand() % 2trs)            (:638-655)
 *   aggregate_push   : aggr += beta[i] * chunk_i                                                               (:327-334)
 *   aggregate_finish : C_f = shockwave_commit(aggr, 32)                                                        (:343-346)
 *   reply_push       : update_reply (:59-111): rows RS-encoded, each queried column RS-encoded, the queried entry appended to
 *                      that query's reply; an all-zero chunk appends nothing (:510-517)
 *   finish           : replies, Merkle paths (open_tree_blake(Commitment_MT, I[i], 2B/trs), :684-687; d_commit_levels = the
 *                      (8B-1) x 32 B tree hobbit_elastic_finish wrote, or NULL), then recursive_prover_RS
 *                      (src/PC_utils.cpp:396-512): P0, P2 = prove_fft_matrix, P3, P5 = prove_fft_matrix, shockwave_prove(C_f, r_x).
 * This entry point is option 1: trs = B/2^11 (4096-point row codes), linear_time = false.  Option 2: hobbit_elastic_open_begin_lin below.
 * The MT_commit_Blake over every reply (:677-680) and verify_claim_opt_blake are verifier-side accounting whose results are
 * discarded; they are not run.
 * All hobbit_elastic_open_out pointers are host buffers (NULL = not wanted, except qpoly/r/vr/fin/checks):
 *   cols, rows : queries x u32;  rv0 : 1 F;  reply : queries x reply_len F (reply_len = non-zero chunks <= N/B);
 *   paths : queries x log2(4B) x 32 B;  cf_root : 32 B;  ncols : distinct queried columns;
 *   qpoly / r : P0, P2, P3, P5 back to back, rounds log2(np2 * 2trs), log2(2trs), log2(2B), 12 (np2 = ncols rounded up to a
 *               power of two);  vr : 4 x 2 F;  fin : 4 F;
 *   checks[2] : prove_fft_matrix's exit(-1) sum checks for P2 and P5 (src/sumcheck.cpp:3016-3019), 1 = holds;
 *   rx : r_x = P5.randomness[0], 12 + log2(trs) F;  sp_f : transcript of shockwave_prove(C_f, r_x) (NULL: not run). */
typedef struct {
    uint32_t *cols, *rows; hobbit_F *rv0; hobbit_F *reply; int *reply_len; uint8_t *paths; uint8_t *cf_root; int *ncols;
    hobbit_F *qpoly, *r, *vr, *fin; int *checks; hobbit_F *rx;
    hobbit_shockwave_out *sp_f;
    /* read only by an opening begun with hobbit_elastic_open_begin_lin (below); option-1 callers may pass the shorter struct that ends at sp_f */
    uint8_t *cc_root; int *nrem; hobbit_F *scal; hobbit_shockwave_out *sp_c;
} hobbit_elastic_open_out;
int hobbit_elastic_open_begin(hobbit_ctx *ctx, size_t N, size_t B, int trs, const hobbit_F *h_x, int queries, hobbit_elastic_open **out);
/* Elastic_PC open, RS x expander (linear_time == true; test_Elastic_PC option 2, src/Elastic_PC.cpp:762-784: tensor_row_size = B / 2^14, 5900
 * queries, expander_init_store(tensor_row_size) -- the graphs of that size must have been uploaded to ctx).  Same three passes, same push / finish
 * / free entry points as above:
 *   begin_lin        : beta, r_v[0], the queries (:638-655); the distinct queried columns and, of those, the "remaining" ones that have a
 *                      queried parity row (row >= tensor_row_size; :357-391)
 *   aggregate_finish : C_f; aggregated_tensor (rows RS-encoded); aux_commit = expander codewords of the remaining columns; C_c =
 *                      shockwave_commit(pad(aux_commit), 32)                                                                      (:348-411)
 *   reply_push       : update_reply_spielman (:431-485) AS BUILT: a column WITHOUT a queried parity row is expander-encoded; for a column WITH one
 *                      the reference copies the un-encoded column over the first tensor_row_size entries of its 2*tensor_row_size buffer and
 *                      then reads the queried parity row from what the buffer still holds -- the parity of the last column that was encoded in
 *                      this chunk (zeros before any was).  The real reference returns these bytes deterministically
 *                      (oracle/check_elastic_open2_determinism.py; fixture tests/golden/elastic_open2.npz).
 *                      Reply row k belongs to the k-th query in (sorted column, then query order), as the reference appends them (:476-478).
 *   finish           : replies, paths (query order), recursive_prover_Spielman_stream (src/PC_utils.cpp:168-270): P1 = prove_linear_code over
 *                      sum_i s^i aux_commit[i], P2, P3 (aux_commit against repeated squares of s2 at positions 0 .. queries-1),
 *                      shockwave_prove(C_c, P3.r), y1, P5 = prove_fft_matrix (seeded with the last entry of an r one entry longer than it
 *                      uses), shockwave_prove(C_f, r_x).
 * Output: qpoly / r : P1, P2, P3, P5 back to back, rounds log2(2trs), log2(2B/trs), log2(np), log2(2B/trs) with np = nrem * 2trs rounded up
 *   to a power of two; vr 4 x 2 F; fin 4 F; checks[1]: prove_fft_matrix's exit(-1) sum check; rx: log2(B) F; scal = s[0], s2, y1 (required);
 *   sp_c, sp_f (required); cc_root 32 B; nrem; ncols; reply = queries x reply_len. */
int hobbit_elastic_open_begin_lin(hobbit_ctx *ctx, size_t N, size_t B, int trs, const hobbit_F *h_x, int queries, hobbit_elastic_open **out);
/* what the begin call derived from its query draws: distinct queried columns, remaining columns (0 for option 1) and np = the padded length of
 * aux_commit = the size of the polynomial C_c commits to (what sp_c's buffers must be sized for; 0 for option 1) */
void hobbit_elastic_open_dims(const hobbit_elastic_open *e, int *ncols, int *nrem, size_t *np);
int hobbit_elastic_open_aggregate_push(hobbit_ctx *ctx, hobbit_elastic_open *e, const hobbit_F *d_chunk);
int hobbit_elastic_open_aggregate_finish(hobbit_ctx *ctx, hobbit_elastic_open *e);
int hobbit_elastic_open_reply_push(hobbit_ctx *ctx, hobbit_elastic_open *e, const hobbit_F *d_chunk);
int hobbit_elastic_open_finish(hobbit_ctx *ctx, hobbit_elastic_open *e, const uint8_t *d_commit_levels, hobbit_elastic_open_out *out);
void hobbit_elastic_open_free(hobbit_elastic_open *e);

/* Our_PC open_standard with linear_time == false (src/Our_PC.cpp:604-692; test_PC option 1 and the circuit polynomial of
 * prove_circuit_standard, src/main.cpp:1043-1051,1082: RS x RS tensor, tensor_row_size = 128, 790 queries): r_v[0]; _aggregate (:258-276:
 * the aggregate and C_f, no C_c); the query draws; replies from the retained tensor; Merkle paths; recursive_prover_RS
 * (src/PC_utils.cpp:396-512) with shockwave_prove(C_f, r_x).  The commitment must come from hobbit_commit_standard(..., linear_time = 0).
 * Output as hobbit_elastic_open_out with reply = queries x K, paths = queries x log2(N/K) x 32 B, the P3 rounds log2(2 N/K) and the
 * P5 rounds / first part of rx log2(2 (N/K) / trs) long. */
int hobbit_open_standard_rs(hobbit_ctx *ctx, const hobbit_F *d_poly, size_t N, const hobbit_commitment *c, const hobbit_F *h_x, int queries, hobbit_elastic_open_out *out);

/* ---- sumchecks ----------------------------------------------------------------------------- */
/* generate_2product_sumcheck_proof (src/sumcheck.cpp:2391-2460).  Inputs preserved.
 * h_qpoly: rounds x 3 F (a,b,c highest degree first); h_r: rounds F; h_vr: 2 F; h_final: 1 F. */
int hobbit_sumcheck2(hobbit_ctx *ctx, const hobbit_F *d_v1, const hobbit_F *d_v2, size_t n, const hobbit_F *prev_r,
                     hobbit_F *h_qpoly, hobbit_F *h_r, hobbit_F *h_vr, hobbit_F *h_final);
/* Degree-4 gate-consistency sumcheck: the in-memory phase of prove_gate_consistency (src/sumcheck.cpp:875-929) over the six
 * folded tables fold_add, fold_beta, fold_L, fold_R, fold_O, fold_mul (n F each, device):
 *   sum_j  add*beta*(a0 L + a1 R) + a2*mul*beta*L*R + a3*beta*O,   h_a = a[0..3] (generate_randomness(4), :873).
 * Transcript as the reference: rand = mimc_hash(coefficient, rand) for poly.a..e, challenge = rand, adjacent-pair fold.
 * The reference folds the tables in place and then only reads element 0 of each; here the inputs are preserved and those six
 * values are returned in h_final (order as the arguments).  h_rand / h_sum: transcript state and claimed sum, in/out;
 * h_poly: rounds x 5 F (a..e); h_r: rounds F; *h_check = 1 iff every "Error in gate consistency 2" comparison held. */
int hobbit_gate_sumcheck(hobbit_ctx *ctx, const hobbit_F *d_add, const hobbit_F *d_beta, const hobbit_F *d_L, const hobbit_F *d_R, const hobbit_F *d_O,
                         const hobbit_F *d_mul, size_t n, const hobbit_F *h_a, hobbit_F *h_rand, hobbit_F *h_sum, hobbit_F *h_poly, hobbit_F *h_r,
                         hobbit_F *h_final, int *h_check);
/* _generate_3product_sumcheck_proof (src/sumcheck.cpp:1974-2058): pre-round-challenge fold order
 * kept.  The reference destroys its inputs in place; here d_v1..3 are preserved (the folded
 * values are returned in h_vr).  h_cpoly: rounds x 4 F; h_vr: 3 F. */
int hobbit_sumcheck3(hobbit_ctx *ctx, const hobbit_F *d_v1, const hobbit_F *d_v2, const hobbit_F *d_v3, size_t n,
                     const hobbit_F *prev_r, hobbit_F *h_cpoly, hobbit_F *h_r, hobbit_F *h_vr, hobbit_F *h_final);

/* ---- code-membership and FFT-as-sumcheck proofs (pieces of recursive_prover_Spielman) -------- */
/* evaluate_parity_matrix (src/sumcheck.cpp:2888-2929): d_A[0..size_a) = H^T d_beta over the uploaded
 * graphs of code length n (size_a >= codeword length; the reference passes codeword.size() = 2n) */
int hobbit_parity_matrix(hobbit_ctx *ctx, const hobbit_F *d_beta, size_t size_a, long long n, hobbit_F *d_A);
/* phiGInit (src/utils.cpp:694-755): the 2^n table of the FFT-as-multilinear-extension */
int hobbit_phi_g(hobbit_ctx *ctx, const hobbit_F *h_rx, int n, const hobbit_F *h_scale, int is_ifft, hobbit_F *d_out);
/* prepare_matrix(transpose(M), r) (src/utils.cpp:758-775): d_out[c] = evaluation over the row index of
 * column c of the row-major rows x cols matrix d_M at r[0..k) */
int hobbit_prepare_matrix_cols(hobbit_ctx *ctx, const hobbit_F *d_M, size_t rows, size_t cols, const hobbit_F *h_r, int k, hobbit_F *d_out);
/* prove_linear_code (src/sumcheck.cpp:3223-3235) with the challenge vector r1 supplied by the caller
 * (the reference draws it with generate_randomness; the host mirror does that); outputs as sumcheck2 */
int hobbit_prove_linear_code(hobbit_ctx *ctx, const hobbit_F *d_codeword, size_t size, long long n, const hobbit_F *h_r1, hobbit_F *h_qpoly,
                             hobbit_F *h_r, hobbit_F *h_vr, hobbit_F *h_final);
/* prove_fft (src/sumcheck.cpp:2975-2987): m of s elements is zero-padded to 2s; h_r has log2(2s) entries */
int hobbit_prove_fft(hobbit_ctx *ctx, const hobbit_F *d_m, size_t s, const hobbit_F *h_r, hobbit_F *h_qpoly, hobbit_F *h_rr, hobbit_F *h_vr,
                     hobbit_F *h_final);
/* prove_fft_matrix (src/sumcheck.cpp:2989-3027): M rows x cols row-major; h_r = [log2(2 cols) | log2 rows] */
int hobbit_prove_fft_matrix(hobbit_ctx *ctx, const hobbit_F *d_M, size_t rows, size_t cols, const hobbit_F *h_r, hobbit_F *h_qpoly, hobbit_F *h_rr,
                            hobbit_F *h_vr, hobbit_F *h_final);

/* batch_3product_sumcheck (src/sumcheck.cpp:275-372): `batches` table triples of power-of-two lengths h_lens[j],
 * concatenated in d_t1/d_t2/d_t3 (inputs preserved; the reference folds in place).  h_cpoly: rounds x 4 F
 * (rounds = log2 max length), h_r: rounds F, h_vr: batches x 3 F. */
int hobbit_batch_3product_sumcheck(hobbit_ctx *ctx, const hobbit_F *d_t1, const hobbit_F *d_t2, const hobbit_F *d_t3, const size_t *h_lens, int batches,
                                   const hobbit_F *h_a, hobbit_F *h_cpoly, hobbit_F *h_r, hobbit_F *h_vr);
/* prove_multiplication_tree_new (src/sumcheck.cpp:35-257) for power-of-two vectors x size (d_input row-major).
 * h_prev_x NULL: the challenge over the vector index is drawn with generate_randomness on the host (vectors > 1).
 * Layer transcripts (top layer first) back to back in h_cpoly (4 F per round) / h_r; h_vr 3 F and h_fin 1 F per
 * layer; h_final_r: log2(vectors*size) F.  *layers_out = number of sumcheck layers. */
int hobbit_mul_tree(hobbit_ctx *ctx, const hobbit_F *d_input, size_t vectors, size_t size, const hobbit_F *h_previous_r, const hobbit_F *h_prev_x,
                    hobbit_F *h_cpoly, hobbit_F *h_r, hobbit_F *h_vr, hobbit_F *h_fin, hobbit_F *h_final_r, hobbit_F *h_out_eval, hobbit_F *h_final_eval, int *layers_out);

/* ---- streaming (space-efficient) sumcheck building blocks: per-chunk error terms and folds ---- */
/* compute{2,3,4}p_error_terms (src/sumcheck.cpp:374-432, has_lookups == false): h_K is ACCUMULATED into, as the
 * reference's F& parameters are.  d_gate: int32 gate selectors (vector<int> buff_S). */
int hobbit_compute2p_error_terms(hobbit_ctx *ctx, const hobbit_F *d_b1, const hobbit_F *d_b2, const hobbit_F *d_f1, const hobbit_F *d_f2, size_t n, hobbit_F *h_K);
int hobbit_compute3p_error_terms(hobbit_ctx *ctx, const hobbit_F *d_b1, const int32_t *d_gate, const hobbit_F *d_f1, const hobbit_F *d_f2, const hobbit_F *d_f3,
                                 const hobbit_F *d_beta, size_t n, hobbit_F *h_K);
int hobbit_compute4p_error_terms(hobbit_ctx *ctx, const hobbit_F *d_b1, const hobbit_F *d_b2, const hobbit_F *d_b3, const int32_t *d_gate, const hobbit_F *d_f1,
                                 const hobbit_F *d_f2, const hobbit_F *d_f3, const hobbit_F *d_f4, size_t n, hobbit_F *h_K);
/* fold += rand * chunk (src/sumcheck.cpp:862-869, 1130-1135); the i32 form folds a gate selector (or 1 - selector) */
int hobbit_fold_axpy(hobbit_ctx *ctx, hobbit_F *d_fold, const hobbit_F *d_buff, const hobbit_F *h_rand, size_t n);
int hobbit_fold_axpy_i32(hobbit_ctx *ctx, hobbit_F *d_fold, const int32_t *d_sel, const hobbit_F *h_rand, int one_minus, size_t n);
/* batch_prod (src/sumcheck.cpp:1093-1136), one stream step over `batches` table triples of n elements (flat [batches][n]):
 * error terms on the device, transcript (mimc) on the host, the three fold tables updated in place */
int hobbit_batch_prod(hobbit_ctx *ctx, hobbit_F *d_f1, hobbit_F *d_f2, hobbit_F *d_f3, const hobbit_F *d_b1, const hobbit_F *d_b2, const hobbit_F *d_b3, int batches,
                      size_t n, const hobbit_F *h_r_last, const hobbit_F *h_a, const hobbit_F *h_rem_beta, hobbit_F *h_Kf, hobbit_F *h_Kp, hobbit_F *h_rand);

/* ---- streaming provers over a caller-supplied chunk source (BASELINE config 4: the MLP prover's math phases) ------------------- */
/* The reference re-generates its streams on demand (read_stream / read_trace, src/witness_stream.cpp) instead of holding them;
 * that machinery (and the Seval oracle behind it) stays with the host.  A source is called once per read, in stream order:
 * it must make the next n elements available on the device and hand back pointers that stay valid until its next call; n == 0
 * means reset_stream (src/witness_stream.cpp:228-234).  Its writes must be ordered after the work already queued on the
 * context's stream (hobbit_memcpy_h2d does that).  Return non-zero to abort. */
typedef int (*hobbit_chunk_source)(void *user, size_t n, const hobbit_F **d_chunk);
typedef int (*hobbit_trace_source)(void *user, size_t n, const hobbit_F **d_L, const hobbit_F **d_R, const hobbit_F **d_O, const int32_t **d_S);
/* read_mul_tree_layer (src/witness_stream.cpp:2413-2456; every stream but "wiring_consistency_check"): d_out[0..size) = products of
 * 2^layer consecutive stream elements (reads of 2*size elements; 1 <= layer).  read_mul_tree_data (:2458-2510): level 0 = `size`
 * products of 2^layer elements (reads of `size` elements), level i = products of 2^distance entries of level i-1; levels back to
 * back in d_out (size, size >> distance, ...). */
int hobbit_read_mul_tree_layer(hobbit_ctx *ctx, hobbit_chunk_source source, void *user, size_t size, int layer, hobbit_F *d_out);
int hobbit_read_mul_tree_data(hobbit_ctx *ctx, hobbit_chunk_source source, void *user, size_t size, int layer, int distance, int batches, hobbit_F *d_out);
/* generate_claims_opt (src/sumcheck.cpp:1014-1054): h_r (rlen F) is shared by the batches; h_claims: batches F */
int hobbit_generate_claims_opt(hobbit_ctx *ctx, hobbit_chunk_source source, void *user, size_t fd_size, size_t B, const hobbit_F *h_r, int rlen, int batches, int layer_id,
                               int distance, hobbit_F *h_claims);
/* generate_3product_sumcheck_beta_stream_batch_optimized (src/sumcheck.cpp:1150-1393): one pass of batch_prod over the stream
 * (error terms on the device, MiMC transcript on the host), batch_3product_sumcheck on the folded tables, the Partial_Evals pass,
 * the closing 2-product sumcheck over the chunk challenges.  libc draws in the reference's order (a, b, the pad challenge).
 * h_r: batches rows of rlen F.  All hobbit_stream3_out pointers are host buffers:
 *   new_claims: batches F; new_r: batches rows at stride new_r_ld (row i: 1 + (log2 B - i*distance) + log2(size/2B) entries);
 *   cpoly1 / r1 / vr1: batch_3product_sumcheck's transcript (log2 B rounds x 4 F; batches x 3 F);
 *   qpoly2 / r2 / vr2 / fin2: the 2-product sumcheck (log2(size/2B) rounds);  R: the permuted chunk challenges (size/2B F, nullable);
 *   checks[3]: K_partial == old_claims ("Error in sumcheck 0": the reference only prints), "Error in sumcheck 1", "Error in sumcheck 2". */
typedef struct {
    hobbit_F *new_claims, *new_r; int new_r_ld;
    hobbit_F *cpoly1, *r1, *vr1, *qpoly2, *r2, *vr2, *fin2, *R; int *checks;
} hobbit_stream3_out;
int hobbit_sumcheck3_stream_batch(hobbit_ctx *ctx, hobbit_chunk_source source, void *user, size_t fd_size, size_t B, const hobbit_F *h_r, int rlen, int batches, int distance,
                                  int layer_id, const hobbit_F *h_old_claims, int n_old, hobbit_stream3_out *out);
/* prove_multiplication_tree_stream_shallow (src/sumcheck.cpp:1746-1915) for every stream but "wiring_consistency_check", WITHOUT
 * commit_layers / open_layers (Elastic_PC commit / open of the "PC_layer" streams: the caller's business -- they run only when
 * layers > distance and naive == 0).  output: the `vectors` products; cpoly .. layers: the in-memory tree's transcript, as
 * hobbit_mul_tree; steps[0 .. *n_steps): the streaming sumchecks, last product layer first (buffers as hobbit_stream3_out, supplied by
 * the caller, max_steps of them); claims0: generate_claims_opt's output on the batched path (nullable); stream_layers: `layers`. */
typedef struct {
    hobbit_F *output, *cpoly, *r, *vr, *fin, *final_r, *out_eval, *final_eval; int *layers;
    hobbit_stream3_out *steps; int max_steps; int *n_steps; hobbit_F *claims0; int *stream_layers;
} hobbit_mul_stream_out;
int hobbit_mul_tree_stream_shallow(hobbit_ctx *ctx, hobbit_chunk_source source, void *user, size_t fd_size, size_t B, int vectors, size_t size, const hobbit_F *h_previous_r,
                                   int distance, const hobbit_F *h_prev_x, int naive, hobbit_mul_stream_out *out);
/* prove_gate_consistency (src/sumcheck.cpp:796-975): the chunk loop (compute{2,3,4}p_error_terms, transcript, six folds), the degree-4
 * sumcheck over the folded tables, the Peval pass and the closing 2-product sumcheck.  The trace source is read n_chunks times, reset,
 * and read n_chunks times again.  Host outputs: R (n_chunks F, R[0] = 1), a (4), poly (log2 B x 5 F), gr (log2 B), fin6 (the folded
 * add, beta, L, R, O, mul), Peval (6 x n_chunks), b (6), q2 / r2 / vr2 / fin2 (log2 n_chunks rounds), checks[3] ("Error in gate
 * consistency 1 / 2 / 3"; 1 = holds). */
typedef struct { hobbit_F *R, *a, *poly, *gr, *fin6, *Peval, *b, *q2, *r2, *vr2, *fin2; int *checks; } hobbit_gate_stream_out;
int hobbit_gate_consistency_stream(hobbit_ctx *ctx, hobbit_trace_source source, void *user, size_t n_chunks, size_t B, const hobbit_F *h_r, hobbit_gate_stream_out *out);

/* The reference's globals has_lookups / lookup_rand (src/main.cpp:67,70; set at :888,:910 for circuits with lookup gates): while on,
 * hobbit_compute3p_error_terms maps selectors 0 and 4 -> 1, 2 -> lookup_rand[0], 3 -> lookup_rand[1], anything else -> 0, and
 * hobbit_compute4p_error_terms uses [selector == 1] (src/sumcheck.cpp:388-394, 413-427).  h_lookup_rand: 2 elements (ignored when on == 0). */
int hobbit_set_lookups(hobbit_ctx *ctx, int on, const hobbit_F *h_lookup_rand);
/* prove_gate_consistency_lookups (src/sumcheck.cpp:503-795), the gate-consistency prover of circuits with lookup gates (main.cpp:915);
 * needs hobbit_set_lookups(ctx, 1, lookup_rand) first.  Trace selectors: 0 addition, 1 multiplication, 2 lookup (read_trace,
 * src/witness_stream.cpp:1724-1746).  Same stream protocol as hobbit_gate_consistency_stream.  Host outputs: R (n_chunks, R[0] = 1), a (5),
 * poly (log2 B x 5), gr (log2 B), fin9 (the folded add_L, add_R, L, R, O, lkp, lkp_O, mul, beta), Peval (8 x n_chunks), b (8),
 * q2 / r2 / vr2 / fin2, checks[5]: "Error in gate consistency 1 / 2 / 3", the per-chunk Kf_M self-check (:635), the Kf_lkp one (:650). */
typedef struct { hobbit_F *R, *a, *poly, *gr, *fin9, *Peval, *b, *q2, *r2, *vr2, *fin2; int *checks; } hobbit_gate_lkp_stream_out;
int hobbit_gate_consistency_lookups_stream(hobbit_ctx *ctx, hobbit_trace_source source, void *user, size_t n_chunks, size_t B, const hobbit_F *h_r,
                                           hobbit_gate_lkp_stream_out *out);

/* ---- synthetic inputs on the device (bench / tests) ---------------------------------------- */
/* splitmix64-derived full-range elements: element i = (sm(seed,2i+1) mod p, sm(seed,2i+2) mod p) */
int hobbit_fill_splitmix(hobbit_ctx *ctx, hobbit_F *d_out, size_t n, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
