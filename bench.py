#!/usr/bin/env python3
"""bench.py -- Our_PC commit + open on MI355X (north-star size 2^28), one JSON line on rank 0.

A "step" (default --phase commit+open) is one Our_PC commit_standard (reference src/Our_PC.cpp:146-171: RS row FFTs, expander column
encode, BLAKE3 Merkle-Damgard leaf chain over the K chunks, Merkle tree) followed by the whole prover side of open_standard
(src/Our_PC.cpp:604-661 + recursive_prover_Spielman, src/PC_utils.cpp:271-385, with both shockwave_prove / WHIR proofs) of
test_PC(2^28, 4, 32) -- all on the GPU through the C ABI (libhobbit_hip.so), with the polynomial already resident in HBM when the
timed region starts.  --phase commit times the commit alone.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--logn 28] [--chunks 32] [--mode replicas|sharded]

N > 1: launched by torch.distributed.run, one rank per GPU.  Default mode for N > 1 is `sharded`: ONE commitment and ONE opening
of the same 2^logn polynomial, chunks sharded over the ranks (parallel.py: by default the chain relay -- contiguous chunks per rank, the leaf
chain's running states handed from rank to rank, the tree on the last rank; --exchange alltoall: one digest exchange, one all-gather of subtree roots, one
all-gather of partial aggregates) -- the north-star's 1/2/4/8 curve, "scaling": "strong".  --mode replicas commits and opens an
independent polynomial per rank (no data-path collective, "scaling": "weak").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s


def fft_butterflies(s):
    return (s // 2) * (s.bit_length() - 1)


def commit_op_counts(N, K, edges, nz=1.0):
    """per chunk mul = trs*FFT(4096) + cols*E(trs); add = 2*trs*FFT(4096) + cols*E(trs);
    BLAKE3 compressions per chunk and leaf: one for the 64 bytes of the four tensor entries unless all four rows lie past the codeword
    length (the constant H(0^64), not recomputed: share nz of the leaves has a non-zero row), one for the chain; + (M-1) for the tree.
    nz = 1 gives the reference's nominal 2M per chunk."""
    M = N // K
    trs = N // (K << 11)
    cols = 2 * M // trs
    mul = K * (trs * fft_butterflies(cols) + cols * edges)
    add = K * (2 * trs * fft_butterflies(cols) + cols * edges)
    comp = int(K * M * (1 + nz)) + (M - 1)
    return mul, add, comp


# which resource binds each bulk kernel: "valu" kernels report frac_valu (issue cycles the instruction stream needs at the
# sustained clock / measured time) as the primary fraction; the HBM fraction the contract asks for stays in `frac`
KERNEL_BOUND = {"k_leaf_chain": "valu", "k_fft4096": "valu", "k_encode_A": "valu", "k_encode_B": "valu", "k_encode": "valu", "k_transpose": "hbm",
                "k_enc_fat_A": "hbm", "k_enc_fat_D": "hbm", "k_encode_M": "valu", "k_enc_fat_C1": "hbm", "k_encode_M2": "valu",
                "k_inner_digests": "valu", "k_chain_digests": "valu", "k_leaf_chain_relay": "valu", "k_aggregate": "hbm"}
SUSTAINED_GHZ, PEAK_GHZ = 1.78, 2.4       # profiles/r01_microbench.txt: clock held under the VALU-heavy kernels; the chip's peak clock
TRAFFIC_PROFILE = "r03_hbm_traffic_commit_2e28.json"     # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (scripts/hbm_traffic.py)

ROOFLINE_NOTES = {
    "k_leaf_chain": "VALU-issue bound, not HBM bound: BLAKE3 compress = 2023 SIMD-cycles per wave-compression with v_alignbit/v_add3 at half rate "
                    "(profiles/r01_microbench.txt); 2^29+2^23 compressions = 9.4 ms at the 1.78 GHz the chip holds = the measured time",
    "k_fft4096": "bound by v_mad_u64_u32 issue (12 per F_{p^2} product, half rate)",
    "k_encode_A": "bound by gather latency (L2 edge records + LDS) and per-slice reduction overhead",
    "k_encode_B": "bound by seven small dependent SpMV steps (barriers) and gather latency",
    "k_transpose": "HBM bound",
    "k_leaf_chain_relay": "the leaf chain of the relay commit (one launch per block of leaves): VALU-issue bound like k_leaf_chain",
}


def measured_traffic(kernel, logn, K):
    """HBM bytes per launch of `kernel`.  PMC counters cannot be collected inside a timed run (rocprofv3 --pmc is its own pass and
    serialises kernels), so this REPLAYS the committed FETCH_SIZE/WRITE_SIZE passes of this same command (profiles/, gfx950-corrected
    by scripts/hbm_traffic.py); only valid for the default 2^28 / K=32 workload.  Returns (bytes or None, provenance string)."""
    if logn != 28 or K != 32:
        return None, "not collected for this workload"
    path = os.path.join(ROOT, "profiles", TRAFFIC_PROFILE)
    try:
        with open(path) as f:
            d = json.load(f)
        return d["kernels"][kernel]["hbm_bytes_per_launch"], "replayed from profiles/%s (separate rocprofv3 --pmc passes; code at %s)" % (TRAFFIC_PROFILE, d.get("commit", "?"))
    except Exception:
        return None, "profiles/%s not present" % TRAFFIC_PROFILE


def open_core_op_counts(N, K, edges):
    """analytic F-mul / F-add of the open core: aggregate N; tensor code of the M-element aggregate; [M'|C].s and
    beta^T[M'|C] (4M each); eq tables (2 x 4M + small); axpy 4M; 2-product sumchecks 6n mul + 10n add
    over n = 2trs, 4096, 4M, 4M, 4096; evaluate_vector 2M."""
    M = N // K
    trs = N // (K << 11)
    cols = 2 * M // trs
    big = 4 * M
    sc_n = 2 * trs + cols + 2 * big + cols
    mul = N + trs * fft_butterflies(cols) + cols * edges + 2 * big + 2 * big + big + 6 * sc_n + 2 * M + edges
    add = N + 2 * trs * fft_butterflies(cols) + cols * edges + 2 * big + 2 * big + big + 10 * sc_n + 2 * M + edges
    return mul, add


def whir_prove_op_counts(N):
    """F-mul / F-add of _whir_prove's prover side (src/Virgo.cpp:519-686) on an N-coefficient polynomial, counted from the
    restatement in oracle/hobbit_oracle.c: eq table + evaluation (2N), 4 fold rounds per iteration (6L mul + 10L add each),
    one FRI layer FFT per iteration, and 100 (then 100/log2(rate)) out-of-domain eq tables of N/16^iter entries used three times."""
    logn = N.bit_length() - 1
    mul = add = 2 * N
    it, reps = 0, 100
    while True:
        for i in range(4):
            L = N >> (4 * it + i + 1)
            mul += 6 * L; add += 10 * L
        it += 1
        cur, fsz = N >> (4 * it), (2 * N) >> it
        mul += fft_butterflies(fsz); add += 2 * fft_butterflies(fsz) + fft_butterflies(cur)       # FFT + change_form
        if logn - 4 * it <= 4:
            return mul + cur, add + cur
        mul += reps * 4 * cur; add += reps * 4 * cur
        reps = int(100.0 / ((fsz // cur).bit_length() - 1))


def shockwave_prove_op_counts(N, k=32):
    """shockwave_prove (src/Virgo.cpp:435-517) on a k x (N/k) matrix: row aggregation of the matrix and of its encoding (3N),
    whir_commit of the aggregate (change_form + 2w-point FFT), P1 (2-product sumcheck over 2w), prove_fft (phiGInit + sumcheck
    over 2w) and _whir_prove on w coefficients."""
    w = N // k; W = 2 * w
    mul = 3 * N + fft_butterflies(W) + 6 * W + W + 6 * W
    add = 3 * N + 2 * fft_butterflies(W) + fft_butterflies(w) + 10 * W + W + 10 * W
    wm, wa = whir_prove_op_counts(w)
    return mul + wm, add + wa


def open_op_counts(N, K, edges, full=True):
    mul, add = open_core_op_counts(N, K, edges)
    if full:
        M = N // K
        for n in (2 * M, M):                         # C_c over the parity half (trs*cols = 2M), C_f over the aggregate (M)
            m_, a_ = shockwave_prove_op_counts(n)
            mul += m_; add += a_
    return mul, add


def nonzero_group_fraction(trs, code_len):
    """share of the 4-row leaf groups with a non-zero row: rows >= the codeword length are zero in every chunk"""
    return min(1.0, ((code_len + 3) // 4) / (trs / 2.0)) if trs >= 4 else 1.0


def algorithmic_bytes(N, K, world=1, sharded=False, nz=1.0):
    """HBM-compulsory bytes per launch of each commit kernel (DESIGN.md 'Kernels'): what the
    algorithm must move given that the tensor is retained, not what the kernel happens to move."""
    M = N // K
    trs = N // (K << 11)
    f = 1.0 / world if sharded else 1.0                 # share of the chunks a rank encodes
    r0 = int(0.211 * trs) / trs if trs > 13 else 0.0    # |x_1| / |x_0|: output of the first expander level
    return {
        "k_fft4096": f * (16 * N + 32 * N),             # read the polynomial, write the row-major message half (2N elements)
        "k_transpose": f * (32 * N + 32 * N),           # row-major -> codeword-major
        "k_encode": f * (32 * N + 32 * N),              # single-pass encode: read message half, write parity half
        "k_encode_A": f * (32 * N + 32 * N * r0),       # read message half, write x_1 = C_0 x_0
        "k_encode_B": f * (32 * N * r0 + 32 * N * (1 - r0)),   # read x_1, write the rest of the parity half
        # deep codes (trs = 4096), round 3: C_0 by the persistent register-resident kernel (same bytes as pass A); the narrow middle steps read x_1
        # and write [x_2 .. z_1] (622 of the 4096 parity rows at trs = 4096); D_0 reads [x_1 .. z_1] back and writes z_0 and the zero tail
        "k_enc_fat_A": f * (32 * N + 32 * N * r0),
        "k_encode_M": f * (32 * N * r0 + 32 * N * (622.0 / 4096.0)),
        # the default splits the middle in two (HOBBIT_ENC_M2=2): C_1 alone in the fat form reads x_1 (864 rows) and writes x_2 (182 rows); the
        # remaining narrow steps read x_2 and write [x_3 .. z_1] (440 rows)
        "k_enc_fat_C1": f * (32 * N * r0 + 32 * N * (182.0 / 4096.0)),
        "k_encode_M2": f * (32 * N * (182.0 / 4096.0) + 32 * N * (440.0 / 4096.0)),
        # D_0 writes z_0 (1463 rows) and the zero tail (1147 rows) -- or, in a commitment (not in the sharded path's raw shards), z_0 and the three
        # rows up to the next leaf group only: the readers answer the rest as zeros (hobbit_commitment::rows_valid; HOBBIT_COMMIT_SKIP_TAIL=0 writes all)
        "k_enc_fat_D": f * (32 * N * (1486.0 / 4096.0) + 32 * N * ((1466.0 if (not sharded and os.environ.get("HOBBIT_COMMIT_SKIP_TAIL", "1") != "0") else 2610.0) / 4096.0)),
        "k_leaf_chain": 64 * N * nz + 32 * M,           # read the non-zero rows of the tensor once (rows past the codeword length are zero), write the leaves once
        "k_inner_digests": f * (64 * N + 32 * N),       # read the local tensor shard, write its 32-byte digests
        "k_chain_digests": (32 * M * K + 64 * M) * (1.0 / world if sharded else 1.0),
        "k_leaf_chain_relay": f * 64 * N * nz + 64 * M,  # relay commit: read the local shard's non-zero rows once, read + write the 32-byte running states (per step: all blocks)
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--logn", type=int, default=28)
    ap.add_argument("--chunks", type=int, default=32)
    ap.add_argument("--mode", choices=["replicas", "sharded"], default=None,
                    help="default: sharded when --gpus > 1 (ONE commitment + opening, chunks sharded over the GPUs with the digest exchange + "
                         "subtree-root all-gather of parallel.py: strong scaling, the north-star curve), replicas on one GPU; "
                         "replicas = one independent polynomial per GPU (weak)")
    ap.add_argument("--exchange", choices=["relay", "alltoall"], default="relay",
                    help="sharded mode: relay = contiguous chunks per GPU, the leaf chain's 32-byte running states handed from GPU to GPU in blocks "
                         "(M*32 B per hop); alltoall = strided chunks, inner digests exchanged all-to-all (K*M*32 B in total), per-GPU subtrees")
    ap.add_argument("--phase", choices=["commit", "commit+open", "commit+opencore"], default="commit+open",
                    help="commit+open adds the whole prover side of open_standard (recursive_prover_Spielman with both shockwave_prove/WHIR proofs); "
                         "commit+opencore stops before the two shockwave_prove calls")
    ap.add_argument("--queries", type=int, default=5900)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dropin", action="store_true", help="skip the reference-shaped driver leg (host/test_pc <logN> 4 <K>: host vector in, PCIe inclusive)")
    ap.add_argument("--cpu-logn", type=int, default=24, help="size of the CPU-baseline sample (2^24: about 17 s of single-thread reference time)")
    args = ap.parse_args()
    if args.mode is None:
        args.mode = "sharded" if args.gpus > 1 else "replicas"

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node == --gpus"
    torch.cuda.set_device(local_rank)

    from __graft_entry__ import load_package
    mod = load_package()
    hb = mod.Hobbit(local_rank)           # raises without the HIP library / a GPU

    N, K = 1 << args.logn, args.chunks
    trs = N // (K << 11)
    # inputs resident in HBM: full-range synthetic coefficients (a superset of the reference's
    # small `generate_randomness` values), one distinct polynomial per rank
    d_poly = hb.fill_splitmix(N, 1000 + rank)
    hb.rng_reset()
    code_len = hb.expander_init_store(trs)   # graphs drawn on the host with libc, reference order; returns the codeword length
    edges = sum(int(L) * int(d) for (L, R, d, nbr, w) in hb._graph_levels.values())
    mul, add, comp = commit_op_counts(N, K, edges, nonzero_group_fraction(trs, code_len))        # executed compressions (the roofline block uses the same share)
    comp_nominal = commit_op_counts(N, K, edges)[2]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    sharded = args.mode == "sharded"
    if sharded:
        relay = args.exchange == "relay"
        plan = mod.parallel.ShardPlan(N, K, trs, world, contiguous=relay)
        ops_ = mod.parallel.HipOps(hb, torch.device("cuda", local_rank))
        own = plan.chunks_of(rank)
        d_local = hb.alloc(16 * plan.M * len(own))            # this rank's chunks, contiguous; chunk i = splitmix(seed 2000+i)
        for li, i in enumerate(own):
            hb._chk(hb.lib.hobbit_fill_splitmix(hb.ctx, d_local.ptr + 16 * plan.M * li, plan.M, 2000 + i))
        hb.sync()
        last = {}

    do_open = args.phase != "commit"
    full_open = args.phase == "commit+open" or sharded            # the sharded open is always the full prover side
    import numpy as np
    x_open = np.stack([np.arange(1, args.logn + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15 % ((1 << 61) - 1)) % np.uint64((1 << 61) - 1),
                       np.arange(7, args.logn + 7, dtype=np.uint64) * np.uint64(1234567891011) % np.uint64((1 << 61) - 1)], axis=1)
    open_last = {}

    def step():
        if sharded:
            commit_fn = mod.parallel.sharded_commit_relay if relay else mod.parallel.sharded_commit
            last["res"] = commit_fn(ops_, dist, plan, rank, (d_local.ptr, len(own)))
            if do_open:
                ops_.set_local_chunks((d_local.ptr, len(own)))
                open_last["res"] = mod.parallel.sharded_open(ops_, dist, plan, rank, last["res"], x_open, args.queries)
            return                        # (the tensor shard stays with ops_ and is re-used by the next step)
        c = hb.commit_standard((d_poly, N), K, trs, 1, sync=not do_open)        # with an opening behind it: queued, not waited for
        if do_open:
            open_last["res"] = hb.open_core((d_poly, N), c, x_open, args.queries, full=full_open)
        c.free()                          # parks the 16.5 GiB of buffers for the next step

    # Setup, not warm-up: the first call allocates every workspace (16.5 GiB tensor, 12 GiB of scratch), and on this driver the
    # SECOND open of a process carries a one-time ~25 ms GPU-idle gap in front of its first kernel (seen in the rocprofv3 kernel
    # trace, independent of event brackets, pinned buffers or pauses).  Two untimed priming steps take both out of
    # the way whatever --warmup says.
    for _ in range(int(os.environ.get("HOBBIT_BENCH_PRIME", "2"))):
        step()
    for _ in range(args.warmup):
        step()
    # HIP-event brackets over the timed region on the bulk kernels only (mode 2: ~60 launches per step); bracketing every one of the
    # ~500 small launches of the open as well costs milliseconds of host time per step, so the full per-kernel table comes from one
    # extra, untimed, fully bracketed step after the timed region.
    hb.profile(0 if os.environ.get("HOBBIT_BENCH_NOPROF") else 2); hb.profile_reset()
    # The first kernels after the profiler reset / host bookkeeping above were measured 20-35 ms late on this box
    # (an idle-exit effect: the same step is on time when the GPU has just been busy), so keep the GPU busy with
    # ~0.1 s of untimed filler right up to the barrier that opens the timed region.
    _spin = hb.alloc(16 << 24)
    for _ in range(int(os.environ.get("HOBBIT_BENCH_SPIN", "2000"))):
        hb._chk(hb.lib.hobbit_fill_splitmix(hb.ctx, _spin.ptr, 1 << 24, 7))
    # The interpreter's cyclic garbage collector is switched off over the timed region, as the standard library's `timeit` does: with
    # torch imported a full (generation-2) collection walks ~10^6 objects and stops the host thread for 30-35 ms, i.e. the GPU idles
    # through most of a step; it fired in about one step of a hundred (the +34 ms outlier of BENCH_r01 and of this round's earlier runs).
    # Nothing in a step creates reference cycles.  HOBBIT_BENCH_GC=1 leaves the collector on (for the A/B).
    import gc
    gc_off = not os.environ.get("HOBBIT_BENCH_GC")
    if gc_off:
        if os.environ.get("HOBBIT_BENCH_GC_NOCOLLECT") is None:
            gc.collect()
        gc.disable()
    barrier()
    t0 = time.perf_counter()
    hb.timer_begin()
    step_ms = []
    for _ in range(args.steps):
        ts = time.perf_counter()
        step()
        step_ms.append(1e3 * (time.perf_counter() - ts))
    ev_ms = hb.timer_end_ms()
    barrier()
    wall = time.perf_counter() - t0
    if gc_off:
        gc.enable()
    prof = hb.profile_report()
    prof.pop("k_fill_splitmix", None)          # the untimed filler in front of the barrier
    hb.profile(1); hb.profile_reset()
    step()                                     # untimed: every launch bracketed, for the full per-kernel table
    hb.sync()
    prof_full = hb.profile_report()
    hb.profile(0)

    t = torch.tensor([wall], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall_max = float(t.item())

    # a commitment to report (root of rank 0's polynomial / of the sharded commitment)
    if sharded:
        root = bytes(last["res"]["root"]).hex()
    else:
        c = hb.commit_standard((d_poly, N), K, trs, 1)
        root = bytes(c.root()).hex()
        c.free()

    if rank == 0:
        ms_per_step = 1e3 * wall_max / args.steps
        omul, oadd = open_op_counts(N, K, edges, full_open) if do_open else (0, 0)
        if do_open:
            assert open_last["res"]["checks"].tolist() == [1, 1, 1], "open: the reference's consistency checks failed"
            if full_open:
                for sp in ("sp_c", "sp_f"):
                    assert open_last["res"][sp]["wchecks"].tolist() == [1, 1], "open: WHIR round / final checks failed in " + sp
        ops = (mul + add + omul + oadd) * (1 if sharded else world)
        value = ops / (wall_max / args.steps)
        nz = nonzero_group_fraction(trs, code_len)
        ab = algorithmic_bytes(N, K, world, sharded, nz)
        cand = {k: v for k, v in prof.items() if ab.get(k)} or {"k_leaf_chain": (1.0, 1)}
        dom = max(cand, key=lambda k: cand[k][0])
        # `ab` holds algorithmic bytes per STEP; a kernel may take several launches per step (the sharded path encodes chunk by chunk)
        launches_per_step = max(1.0, cand[dom][1] / float(args.steps))
        dom_ms = cand[dom][0] / cand[dom][1]
        # every bulk commit kernel against the HBM roof, per step (a name's launches from inside the open -- the aggregate's own tensor code: k_fft4096,
        # k_transpose, the encode passes on one chunk -- are in its time but not in its bytes, so these fractions err low by a few per cent)
        roofline_kernels = {k: {"ms_per_step": v[0] / args.steps, "algorithmic_GB_per_step": ab[k] / 1e9, "achieved_GBs": ab[k] / (v[0] / args.steps * 1e-3) / 1e9,
                                "frac": ab[k] / (v[0] / args.steps * 1e-3) / 1e9 / HBM_PEAK_GBS, "bound": KERNEL_BOUND.get(k, "hbm")} for k, v in sorted(cand.items()) if v[0] > 0}
        ab[dom] = ab[dom] / launches_per_step
        achieved = ab[dom] / (dom_ms * 1e-3) / 1e9
        out = {
            "metric": "Our_PC %s field-ops/s (F_p^2 mul+add), 2^%d-coefficient multilinear" % (args.phase if do_open else "commit", args.logn),
            "value": value, "unit": "field-ops/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None,
            "dtype": "u64 (F_{p^2}, p=2^61-1) + u32 (BLAKE3)", "data": "synthetic (device splitmix64 full-range coefficients; libc-drawn expander graphs)",
            "config": {"workload": ("Our_PC test_PC(2^%d,4,%d): commit_standard + open_standard/recursive_prover_Spielman %s "
                                    "(aggregate, tensor code of the aggregate, shockwave_commit C_f/C_c, %d queries + replies + Merkle paths, prove_linear_code, "
                                    "three 2-product sumchecks (4096, 2^%d, 2^%d), prove_fft_matrix%s); trs=%d, cols=4096, tensor retained in HBM"
                                    % (args.logn, K, "prover side in full" if full_open else "WITHOUT the two shockwave_prove/WHIR proofs", args.queries,
                                       args.logn - 3, args.logn - 3, ", shockwave_prove(C_c) and shockwave_prove(C_f) with their WHIR proofs" if full_open else "", trs))
                       if do_open else
                       "Our_PC commit_standard (test_PC(2^%d,4,%d) commit phase): trs=%d, cols=4096, tensor retained in HBM; open phase not in the timed region" % (args.logn, K, trs),
                       "N": N, "K": K, "trs": trs, "mode": args.mode, "exchange": args.exchange if sharded else None, "polynomials_per_gpu": (1.0 / world) if sharded else 1},
            "prover_s": wall_max / args.steps, "step_ms_rank0": step_ms, "hip_event_ms_per_step": ev_ms / args.steps,
            "step_ms_median": float(np.median(step_ms)), "step_ms_mean": float(np.mean(step_ms)),
            "step_ms_outliers": [x for x in step_ms if x > 1.25 * float(np.median(step_ms))],
            "f_mul_per_s": mul * (1 if sharded else world) / (wall_max / args.steps),
            "blake3_compressions_per_s": comp * (1 if sharded else world) / (wall_max / args.steps),
            "op_counts": {"f_mul": mul, "f_add": add, "blake3_compress": comp, "blake3_compress_nominal": comp_nominal, "expander_edges": edges, "open_f_mul": omul, "open_f_add": oadd},
            "kernels_ms_per_step": {k: v[0] / args.steps for k, v in sorted(prof.items())},
            "kernels_ms_note": "k_fft4096 and k_transpose run on two streams, overlapped (chunk group g's transpose beside group g+1's FFT): their durations sum to more than their wall-clock span; kernels_ms_extra_profiled_step / launches_extra_profiled_step come from one extra untimed step with every launch bracketed, which runs the open on ONE thread and stream (the timed steps run shockwave_prove(C_c) on a helper context from a second thread and the inner commitments on a third stream)",
            "kernels_ms_extra_profiled_step": {k: v[0] for k, v in sorted(prof_full.items())},
            "launches_extra_profiled_step": {"total": int(sum(v[1] for v in prof_full.values())),
                                             "open": int(sum(v[1] for k, v in prof_full.items() if not (k.startswith(("k_leaf_chain", "k_enc")) or k in ("k_fft4096", "k_transpose", "k_merkle_level", "k_merkle_top"))))},
            "roofline": {"bound": KERNEL_BOUND.get(dom, "hbm"), "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": measured_traffic(dom, args.logn, K)[0], "traffic_source": measured_traffic(dom, args.logn, K)[1],
                         "algorithmic_bytes_per_launch": ab[dom], "avg_launch_ms": dom_ms, "note": ROOFLINE_NOTES.get(dom, "")},
            "roofline_kernels": roofline_kernels,
            "root": root,
        }
        if dom == "k_leaf_chain" and not sharded:
            # the binding resource of the dominant kernel: VALU issue.  2023 SIMD-cycles per wave-compression (7 rounds x 8 G x 36 issue
            # cycles + message/finalisation moves; profiles/r01_microbench.txt), 64 compressions per wave, 1024 SIMDs: the clock at which
            # the chip would have to issue without a single bubble to finish in the measured time -- compare with the 1.78 GHz it
            # sustains under this load (2.4 GHz peak)
            wave_comp = ((1.0 + nz) * K * (N // K)) / 64.0   # per leaf and chunk: the chain compression, plus the inner one unless the group is all zero
            cyc = wave_comp * 2023.0 / 1024.0
            out["roofline"]["binding"] = {"resource": "VALU issue", "simd_cycles_per_launch": cyc,
                                          "implied_issue_clock_ghz": cyc / (dom_ms * 1e-3) / 1e9, "sustained_clock_ghz": SUSTAINED_GHZ, "peak_clock_ghz": PEAK_GHZ}
            # primary fraction for a VALU-bound kernel: issue cycles the instruction stream needs / cycles available at the sustained clock
            out["roofline"]["frac_valu"] = cyc / (dom_ms * 1e-3 * SUSTAINED_GHZ * 1e9)
            out["roofline"]["frac_valu_of_peak_clock"] = cyc / (dom_ms * 1e-3 * PEAK_GHZ * 1e9)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.cpu_logn, K)
            out["cpu_baseline_all_cores"] = cpu_all_cores(K)
            if do_open:
                out["cpu_baseline_open_port"] = cpu_open_port(args.cpu_logn, K, args.queries, full_open)
        if not args.no_dropin and world == 1 and do_open and full_open:
            out["dropin"] = dropin_leg(args.logn, K)
        print(json.dumps(out))
    hb.close()
    if dist is not None:
        dist.destroy_process_group()


def dropin_leg(logn, K):
    """The drop-in number (never `value`): the reference-shaped driver `host/test_pc <logN> 4 <K>` -- the reference's commented-out
    `./pigeon <logN> 4 <K>` hook (src/main.cpp:1176) over the C++ mirror -- run once as its own process, outside the timed region: test_PC
    draws the polynomial into a std::vector on the host, commit_standard(vector<F> &, ...) streams it over PCIe under the commit's kernels
    (hobbit_commit_standard_host) and hands back every Merkle level as the reference's signature demands, open_standard follows.  Reported:
    the driver's own `Commit time` / `Total time` prints and the proof size it prints (the reference's stdout fingerprint)."""
    import re, subprocess, time
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hobbit-space-efficient-zksnark-with-optimal-prover-time_amd", "host", "test_pc")
    if not os.path.exists(exe):
        return {"error": "host/test_pc not built"}
    out = {"driver": "host/test_pc %d 4 %d" % (logn, K), "includes": "host vector -> device over PCIe (pageable source, 64 MiB pinned pieces, pipelined with the row FFTs), "
           "first-use HIP context and allocations, all Merkle levels copied back to the host (the reference's MT_hashes), proof-size accounting"}
    for tag, env in (("pipelined_upload", {}), ("blocking_upload", {"HOBBIT_HOST_BLOCKING_UPLOAD": "1"})):
        t0 = time.time()
        p = subprocess.run([exe, str(logn), "4", str(K)], capture_output=True, text=True, env=dict(os.environ, **env), timeout=600)
        m1 = re.search(r"Commit time: ([0-9.eE+-]+) seconds", p.stdout); m2 = re.search(r"Total time: ([0-9.eE+-]+) seconds", p.stdout)
        last = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else ""
        out[tag] = {"rc": p.returncode, "commit_s": float(m1.group(1)) if m1 else None, "total_s": float(m2.group(1)) if m2 else None,
                    "proof_kb": float(last.split(",")[0]) if "," in last else None, "process_wall_s": time.time() - t0}
    return out


def cpu_baseline(logn, K):
    """The reference's own commit_standard (oracle/_ref, 1 thread: the reference is single-threaded
    by construction) on a bounded sample of the same workload; falls back to the C restatement
    ("port") when the prebuilt reference library did not travel."""
    from oracle import pyoracle
    n = 1 << logn
    trs = n // (K << 11)
    if pyoracle.ref_available():
        lib, kind = pyoracle.Ref(), "reference"
    else:
        lib, kind = pyoracle.Oracle(), "port"
    secs = lib.time_commit_standard(n, K)
    orc = pyoracle.Oracle()
    orc.rng_reset(); orc.generate_randomness(n); orc.expander_init_store(trs)
    edges, dep, m = 0, 0, trs
    while m > 13:
        edges += orc.graph(dep, 0)["L"] * 9 + orc.graph(dep, 1)["L"] * 12
        m = int(0.211 * m); dep += 1
    mul, add, comp = commit_op_counts(n, K, edges)
    return {"value": (mul + add) / secs, "unit": "field-ops/s", "cores": 1, "kind": kind, "seconds": secs,
            "sample": "commit_standard (commit phase only: the reference's open_standard ends in SHA3 from a prebuilt library that is not linked) on "
                      "test_PC(2^%d,4,%d) inputs (trs=%d): the bench workload at 1/%d of its size, single thread" % (logn, K, trs, 1 << (28 - logn))}


def host_threads():
    """Threads this process may really use: its affinity mask, cut to the cgroup's CPU quota when there is one."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def cpu_all_cores(K):
    """the same commit on all host cores -- the oracle's restatement with the rows / columns / leaves of each chunk dealt to
    threads (oracle/hobbit_oracle.c orc_commit_standard_mt; levels identical to the single-thread restatement, tests/test_oracle_selfcheck.py).
    The reference itself is single-threaded, so this leg is always a "port"."""
    from oracle import pyoracle
    orc = pyoracle.Oracle()
    T = host_threads()
    logn = 26 if T >= 8 else 24              # ~10-30 s of CPU work either way
    n = 1 << logn
    trs = n // (K << 11)
    secs = orc.time_commit_standard_mt(n, K, T)
    edges, dep, m = 0, 0, trs
    while m > 13:
        edges += orc.graph(dep, 0)["L"] * 9 + orc.graph(dep, 1)["L"] * 12
        m = int(0.211 * m); dep += 1
    mul, add, comp = commit_op_counts(n, K, edges)
    return {"value": (mul + add) / secs, "unit": "field-ops/s", "cores": T, "kind": "port", "seconds": secs,
            "sample": "commit_standard (commit phase only) on test_PC(2^%d,4,%d) inputs (trs=%d): the bench workload at 1/%d of its size, the oracle's "
                      "restatement on %d threads (rows, columns and leaves of each chunk dealt in ranges)" % (logn, K, trs, 1 << (28 - logn), T)}


def cpu_open_port(logn, K, queries, full=True):
    """The open on the CPU: the oracle's restatement (kind "port", 1 thread); the reference's own open_standard cannot
    run here because it ends in SHA3 from the prebuilt lib/libXKCP.a, which is not linked."""
    from oracle import pyoracle
    orc = pyoracle.Oracle()
    n = 1 << logn
    trs = n // (K << 11)
    orc.rng_reset(); poly = orc.generate_randomness(n); orc.expander_init_store(trs)
    x = orc.generate_randomness(logn)
    t0 = time.perf_counter()
    res = orc.open_standard(poly, K, trs, x, queries) if full else orc.open_core(poly, K, trs, x, queries)
    secs = time.perf_counter() - t0
    edges, dep, m = 0, 0, trs
    while m > 13:
        edges += orc.graph(dep, 0)["L"] * 9 + orc.graph(dep, 1)["L"] * 12
        m = int(0.211 * m); dep += 1
    mul, add = open_op_counts(n, K, edges, full)
    return {"value": (mul + add) / secs, "unit": "field-ops/s", "cores": 1, "kind": "port", "seconds": secs, "checks": res["checks"].tolist(),
            "sample": "open_standard prover side (%s) on test_PC(2^%d,4,%d) inputs, single thread"
                      % ("with both shockwave_prove/WHIR proofs" if full else "no shockwave_prove/WHIR", logn, K)}


if __name__ == "__main__":
    main()
