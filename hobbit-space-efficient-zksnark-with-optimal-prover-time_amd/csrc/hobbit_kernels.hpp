// hobbit_kernels.hpp -- launcher prototypes (definitions in hobbit_kernels.hip)
#pragma once
#include "hobbit_ctx.hpp"

namespace hobbit {
int launch_f_binop(hobbit_ctx *ctx, int op, const F *a, const F *b, F *o, size_t n);
int launch_fill_splitmix(hobbit_ctx *ctx, F *o, size_t n, uint64_t seed);
int launch_u64_bias_fold(hobbit_ctx *ctx, uint64_t *w, size_t n, uint64_t bias, int fold);
int launch_fft_rows(hobbit_ctx *ctx, const F *src, size_t src_ld, uint32_t src_len, F *dst, size_t dst_ld, size_t dst_es, int logn,
                    const F *tw, F scale, int do_scale, uint32_t groups, uint32_t rows_per_group, size_t src_gs, size_t dst_gs);
int launch_fft4096(hobbit_ctx *ctx, const F *src, size_t src_ld, size_t src_es, uint32_t src_len, F *dst, size_t dst_ld, size_t dst_es, const F *tw1,
                   const F *tw2, const F *tw3, F w8, F w8_3, int w4_plus_i, F scale, int do_scale, uint32_t groups, uint32_t rows_per_group,
                   size_t src_gs, size_t dst_gs);
int launch_fft_combine(hobbit_ctx *ctx, int logr, const F *Y, F *dst, size_t dst_ld, const F *tw, uint32_t rows);
int launch_elastic_leaf(hobbit_ctx *ctx, const F *t0, const F *t1, const F *t2, const F *t3, uint32_t rows2, uint32_t cols, int shift, uint8_t *state);
int launch_elastic_inner(hobbit_ctx *ctx, const F *t0, const F *t1, const F *t2, const F *t3, uint32_t rows2, uint32_t cols, int shift, uint8_t *out);
int launch_elastic_finish(hobbit_ctx *ctx, const uint8_t *state, uint32_t rows2, uint32_t cols, uint8_t *leaves);
int launch_transpose(hobbit_ctx *ctx, const F *in, size_t in_gs, uint32_t rows, uint32_t cols, F *out, size_t out_gs, size_t ld_out, uint32_t groups);
int launch_encode(hobbit_ctx *ctx, const F *src, size_t ld_src, F *dst, size_t ld_dst, long long n, size_t batch, int write_msg);
int launch_blake3_64(hobbit_ctx *ctx, const uint8_t *in, uint8_t *out, size_t n);
int launch_hash_md(hobbit_ctx *ctx, const F *xyzw, const uint8_t *prev, uint8_t *out, size_t n);
int launch_merkle_levels(hobbit_ctx *ctx, uint8_t *levels, size_t n, int quirk);
int launch_fft_r8(hobbit_ctx *ctx, const F *src, size_t src_ld, uint32_t src_len, F *dst, size_t dst_ld, size_t dst_es, int logn, const F *tabs, int plus_i,
                  uint32_t groups, uint32_t rows_per_group, size_t src_gs, size_t dst_gs);
int launch_leaf_chain_relay(hobbit_ctx *ctx, const F *tensor, size_t chunk_stride, int K, uint32_t cols, uint32_t half_trs, size_t g_begin, size_t g_count,
                            const uint8_t *state_in, uint8_t *state_out, uint8_t *leaves, uint32_t zero_rows_from, int leaves_inout = 0);
int launch_leaf_chain(hobbit_ctx *ctx, const F *tensor, size_t chunk_stride, int K, uint32_t cols, uint32_t half_trs, uint8_t *leaves, uint32_t zero_rows_from);
int launch_inner_digests(hobbit_ctx *ctx, const F *tensor, size_t chunk_stride, int nchunks, uint32_t cols, uint32_t half_trs, uint8_t *out);
int launch_chain_digests(hobbit_ctx *ctx, const uint8_t *digests, size_t stride_bytes, int K, size_t m, uint8_t *leaves);
int launch_merkle_paths(hobbit_ctx *ctx, const uint8_t *levels, size_t n, const uint64_t *d_pos, size_t nq, int depth, uint8_t *d_paths);
int launch_eq_table(hobbit_ctx *ctx, CHP h_r, int k, F *d_out);
int launch_aggregate(hobbit_ctx *ctx, const F *poly, size_t M, int K, CHP h_beta, F *aggr);
int launch_gather(hobbit_ctx *ctx, const F *tensor, size_t chunk_stride, uint32_t rows2, int K, const uint32_t *d_rows, const uint32_t *d_cols,
                  size_t nq, F *d_reply, uint32_t rows_valid);
int launch_zero_rows(hobbit_ctx *ctx, F *tensor, size_t ncols, uint32_t rows2, uint32_t rows_valid);
int launch_tensor_row(hobbit_ctx *ctx, const F *chunk, uint32_t rows2, uint32_t cols, uint32_t row, F *d_out, uint32_t rows_valid);
int launch_eval_fold(hobbit_ctx *ctx, const F *v, F *o, size_t L, F r);
int launch_eval_fold2(hobbit_ctx *ctx, const F *v, F *o, size_t L, F r0, F r1);
int launch_eval_tail(hobbit_ctx *ctx, const F *v, F *o, size_t n, int levels, const F *r);
int launch_csr_gather(hobbit_ctx *ctx, const uint32_t *rowptr, const uint32_t *idx, const F *w, const F *x, F *y, size_t rows);
int launch_phi_step(hobbit_ctx *ctx, F *g, size_t half, int m, F rx, const F *pm, int last_only);
int launch_phi_head(hobbit_ctx *ctx, F *g, int n, int h, CHP h_rx, F scale, const F *pm);
int launch_fold_rows(hobbit_ctx *ctx, const F *in, F *out, size_t out_rows, size_t cols, F r);
int launch_transpose_ld(hobbit_ctx *ctx, const F *in, size_t in_gs, size_t in_ld, uint32_t rows, uint32_t cols, F *out, size_t out_gs, size_t ld_out,
                        uint32_t groups);
int launch_matvec_rows(hobbit_ctx *ctx, const F *Mx, size_t rows, size_t cols, const F *v, F *out);
int launch_vecmat(hobbit_ctx *ctx, const F *Mx, size_t rows, size_t cols, const F *v, F *out);
int launch_gather_strided(hobbit_ctx *ctx, const F *src, const uint64_t *d_idx, size_t nq, uint32_t m, size_t bmul, size_t stride, F *out);
int launch_dot(hobbit_ctx *ctx, const F *a, const F *b, size_t n, F *part, F *out);
int launch_change_form_tail(hobbit_ctx *ctx, F *data, size_t n, uint32_t T);
int launch_eq_pair_axpy(hobbit_ctx *ctx, CHP h_r1, CHP h_r2, int k, F a, F *d_half, F *d_out);
int launch_scatter(hobbit_ctx *ctx, const uint64_t *idx, const F *val, size_t n, F *out);
int launch_axpy(hobbit_ctx *ctx, F *y, const F *x, F a, size_t n);
int launch_err_terms(hobbit_ctx *ctx, int kind, const F *const *tables, const int32_t *gate, size_t n, MHP h_K);
int launch_axpy_i32(hobbit_ctx *ctx, F *y, const int32_t *sel, F a, int one_minus, size_t n);
int launch_sc3_poly(hobbit_ctx *ctx, const F *s1, const F *s2, const F *s3, size_t L, F *part, F *coef);
int launch_fold3(hobbit_ctx *ctx, const F *s1, const F *s2, const F *s3, F *d1, F *d2, F *d3, size_t L, F r);
int launch_mul_layer(hobbit_ctx *ctx, const F *x, size_t n_out, F *in1, F *in2, F *tr);
int launch_transpose_tw(hobbit_ctx *ctx, const F *in, size_t gs, uint32_t R, F *out, const F *tw, uint32_t half, const F *tw2, uint32_t groups);
int launch_fingerprint(hobbit_ctx *ctx, const F *addr, const F *value, const F *freq, F a, F b, F *out, size_t n);
int launch_fft_cols_r8(hobbit_ctx *ctx, const F *y, size_t gs, int logr, F *out, const F *tw2, const F *tabs, int plus_i, uint32_t batch);
int launch_fft_cols(hobbit_ctx *ctx, const F *y, size_t gs, int logr, F *out, const F *tw2, const F *twr, uint32_t batch);
int launch_build_tw2d(hobbit_ctx *ctx, const F *tw, uint32_t half, uint32_t R, F *out);
int launch_col_digest(hobbit_ctx *ctx, const F *enc, size_t W, int k, int quirk, uint8_t *out);
int launch_change_form_level(hobbit_ctx *ctx, const F *in, F *out, size_t n, size_t S);
int launch_whir_round(hobbit_ctx *ctx, F *poly, F *beta, size_t L, F a, F *part, F *coef);
int launch_eq_step_batched(hobbit_ctx *ctx, const F *old, F *nw, size_t m, size_t ld, const F *z, int v, int level, int reps);
int launch_eq_head_batched(hobbit_ctx *ctx, F *out, size_t ld, const F *z, int v, int h, int reps);
int launch_fill_F(hobbit_ctx *ctx, F *p, size_t stride, size_t n, F v);
int launch_sumcheck2(hobbit_ctx *ctx, const F *v1, const F *v2, size_t n, F prev_r, MHP h_qpoly, MHP h_r, MHP h_vr, MHP h_final);
int launch_sumcheck2_sparse(hobbit_ctx *ctx, const F *v1, const uint64_t *d_idx, const F *d_val, size_t m, size_t n, F prev_r, MHP h_qpoly, MHP h_r, MHP h_vr, MHP h_final);
int launch_gate_lkp_sumcheck(hobbit_ctx *ctx, const F *const tabs[9], size_t n, CHP h_a, MHP h_rand, MHP h_sum, MHP h_poly, MHP h_r, MHP h_final, int *h_check);
int launch_lkp_prepare(hobbit_ctx *ctx, const int32_t *S, const F *L, const F *R, const F *O, int32_t *s2, int32_t *s3, F *blo, size_t n);
int launch_lkp_sel_fold(hobbit_ctx *ctx, const int32_t *S, F rnd, F *aL, F *aR, F *lkp, F *mul, size_t n);
int launch_gate_sumcheck(hobbit_ctx *ctx, const F *const tabs[6], size_t n, CHP h_a, MHP h_rand, MHP h_sum, MHP h_poly, MHP h_r, MHP h_final, int *h_check);
int launch_sumcheck3(hobbit_ctx *ctx, const F *v1, const F *v2, const F *v3, size_t n, F prev_r, MHP h_cpoly, MHP h_r, MHP h_vr, MHP h_final);
int launch_any_nonzero(hobbit_ctx *ctx, const F *v, size_t n, int *d_flag);
int launch_gather_cols(hobbit_ctx *ctx, const F *T, size_t ld, uint32_t nrows, const uint32_t *d_cols, uint32_t ncols, F *G, size_t ldG);
int launch_spread_cols(hobbit_ctx *ctx, const uint32_t *d_cols, const F *d_vals, uint32_t ncols, uint32_t nrows, size_t ld, F *out);
int launch_seg_prod(hobbit_ctx *ctx, const F *in, uint32_t seg, size_t n_out, F *out);
int launch_deinterleave(hobbit_ctx *ctx, const F *v, size_t n, F *a, F *b);
int launch_dot_gen(hobbit_ctx *ctx, const F *a, const F *b, size_t sb, const F *c, size_t n, F *part, F *out);
int launch_dot_i32(hobbit_ctx *ctx, const F *a, const int32_t *sel, int one_minus, size_t n, F *part, F *out);
int launch_i32_to_F(hobbit_ctx *ctx, const int32_t *sel, int one_minus, size_t n, F *y);
int launch_zero(hobbit_ctx *ctx, void *p, size_t bytes);
int launch_copy(hobbit_ctx *ctx, void *d, const void *s, size_t bytes);
}  // namespace hobbit
