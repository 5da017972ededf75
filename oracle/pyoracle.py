"""oracle/pyoracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes/numpy front-ends for
  * ``Oracle``  : oracle/libhobbit_oracle.so, our plain-C restatement (hobbit_oracle.c), and
  * ``Ref``     : oracle/_ref/libhobbit_ref.so, the REAL reference compiled from /root/reference
                  (present only where `make -C oracle ref` has run; travels to the GPU box as a
                  prebuilt file).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
Field arrays are numpy uint64 of shape (..., 2) = (real, img); hashes are uint8 (..., 32).
"""
import ctypes
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.environ.get("HOBBIT_ORACLE_SO") or os.path.join(HERE, "libhobbit_oracle.so")     # (make -C oracle asan-test points this at the sanitizer build)
REF_SO = os.path.join(HERE, "_ref", "libhobbit_ref.so")
P = (1 << 61) - 1

c_sz = ctypes.c_size_t
c_vp = ctypes.c_void_p


def _p(a):
    return a.ctypes.data_as(c_vp)


def F(a):
    """contiguous uint64 (...,2) array"""
    a = np.ascontiguousarray(a, dtype=np.uint64)
    assert a.shape[-1] == 2
    return a


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])


def build_ref():
    subprocess.check_call(["make", "-s", "-j8", "-C", HERE, "ref"])


def ref_available():
    return os.path.exists(REF_SO)


def _dlopen_lazy(path):
    # ctypes.CDLL always adds RTLD_NOW; the reference library keeps SHA3_256 (lib/libXKCP.a, a
    # prebuilt binary we do not link) unresolved, so bind lazily through dlopen itself.
    libdl = ctypes.CDLL(None)
    libdl.dlopen.restype = c_vp
    libdl.dlopen.argtypes = [ctypes.c_char_p, ctypes.c_int]
    h = libdl.dlopen(path.encode(), os.RTLD_LAZY | os.RTLD_LOCAL)
    if not h:
        raise OSError("dlopen failed: " + path)
    return ctypes.CDLL(path, handle=h)


class _Base:
    """Common numpy wrappers; subclasses provide self.lib and self.pfx."""

    def fn(self, name):
        return getattr(self.lib, self.pfx + name)

    # ---- field
    def _bin(self, name, a, b):
        a, b = F(a), F(b)
        o = np.empty_like(a)
        self.fn(name)(_p(a), _p(b), _p(o), c_sz(a.size // 2))
        return o

    def f_add(self, a, b): return self._bin("f_add", a, b)
    def f_sub(self, a, b): return self._bin("f_sub", a, b)
    def f_mul(self, a, b): return self._bin("f_mul", a, b)

    def _un(self, name, a):
        a = F(a)
        o = np.empty_like(a)
        self.fn(name)(_p(a), _p(o), c_sz(a.size // 2))
        return o

    def f_neg(self, a): return self._un("f_neg", a)
    def f_inv(self, a): return self._un("f_inv", a)

    def root_of_unity(self, logn):
        o = np.zeros(2, np.uint64)
        self.fn("root_of_unity")(ctypes.c_int(logn), _p(o))
        return o

    def mimc(self, x, k): return self._bin("mimc", x, k)

    # ---- hashes
    def blake3_64(self, blocks):
        b = np.ascontiguousarray(blocks, dtype=np.uint8).reshape(-1, 64)
        o = np.empty((b.shape[0], 32), np.uint8)
        self.fn("blake3_64")(_p(b), _p(o), c_sz(b.shape[0]))
        return o

    def hash_md(self, xyzw, prev):
        x = F(xyzw).reshape(-1, 4, 2)
        pv = np.ascontiguousarray(prev, dtype=np.uint8).reshape(-1, 32)
        o = np.empty_like(pv)
        self.fn("hash_md")(_p(x), _p(pv), _p(o), c_sz(x.shape[0]))
        return o

    def mt_commit_blake(self, leafs):
        x = F(leafs).reshape(-1, 2)
        n = x.shape[0]
        o = np.zeros((2 * (n // 4), 32), np.uint8)
        f = self.fn("mt_commit_blake"); f.restype = c_sz
        cnt = f(_p(x), (c_sz if self.pfx == "orc_" else ctypes.c_int)(n), _p(o))
        return o[:cnt]

    def create_tree_blake(self, level0):
        l0 = np.ascontiguousarray(level0, dtype=np.uint8).reshape(-1, 32)
        n = l0.shape[0]
        o = np.zeros((2 * n, 32), np.uint8)
        f = self.fn("create_tree_blake"); f.restype = c_sz
        cnt = f(_p(l0), (c_sz if self.pfx == "orc_" else ctypes.c_int)(n), _p(o))
        return o[:cnt]

    # ---- rng / graphs / encode
    def rng_reset(self): self.fn("rng_reset")()

    def generate_randomness(self, n):
        o = np.zeros((n, 2), np.uint64)
        self.fn("generate_randomness")(ctypes.c_int(n), _p(o))
        return o

    def expander_init_store(self, n):
        f = self.fn("expander_init_store"); f.restype = ctypes.c_longlong
        return f(ctypes.c_longlong(n))

    def graph(self, dep, kind):
        R = ctypes.c_longlong(); d = ctypes.c_int()
        f = self.fn("graph_dims"); f.restype = ctypes.c_longlong
        L = f(ctypes.c_int(dep), ctypes.c_int(kind), ctypes.byref(R), ctypes.byref(d))
        nbr = np.zeros(L * d.value, np.int64); w = np.zeros((L * d.value, 2), np.uint64)
        self.fn("graph_edges")(ctypes.c_int(dep), ctypes.c_int(kind), _p(nbr), _p(w))
        return dict(L=L, R=R.value, degree=d.value, nbr=nbr, w=w)

    def graph_set_weights(self, dep, kind, w):
        w = F(w)
        self.fn("graph_set_weights")(ctypes.c_int(dep), ctypes.c_int(kind), _p(w))

    def encode_monolithic(self, src):
        s = F(src).reshape(-1, 2)
        n = s.shape[0]
        d = np.zeros((2 * n, 2), np.uint64)
        f = self.fn("encode_monolithic"); f.restype = ctypes.c_int
        ln = f(_p(s), _p(d), ctypes.c_longlong(n))
        return d, ln

    # ---- fft & friends
    def precompute_beta(self, r):
        r = F(r).reshape(-1, 2)
        k = r.shape[0]
        o = np.zeros((1 << k, 2), np.uint64)
        self.fn("precompute_beta")(_p(r), ctypes.c_int(k), _p(o))
        return o

    def evaluate_vector(self, v, r):
        v = F(v).reshape(-1, 2); r = F(r).reshape(-1, 2)
        o = np.zeros(2, np.uint64)
        self.fn("evaluate_vector")(_p(v), c_sz(v.shape[0]), _p(r), ctypes.c_int(r.shape[0]), _p(o))
        return o

    def compute_tensorcode(self, msg, trs, lin):
        m = F(msg).reshape(-1, 2)
        M = m.shape[0]
        o = np.zeros((2 * trs, 2 * M // trs, 2), np.uint64)
        self.fn("compute_tensorcode")(_p(m), c_sz(M), ctypes.c_int(trs), ctypes.c_int(lin), _p(o))
        return o

    # ---- sumchecks
    def sumcheck2(self, v1, v2, prev_r):
        v1 = F(v1).reshape(-1, 2); v2 = F(v2).reshape(-1, 2); pr = F(prev_r).reshape(2)
        n = v1.shape[0]; rounds = n.bit_length() - 1
        q = np.zeros((rounds, 3, 2), np.uint64); r = np.zeros((rounds, 2), np.uint64)
        vr = np.zeros((2, 2), np.uint64); fin = np.zeros(2, np.uint64)
        self.fn("sumcheck2")(_p(v1), _p(v2), c_sz(n), _p(pr), _p(q), _p(r), _p(vr), _p(fin))
        return dict(poly=q, r=r, vr=vr, fin=fin)

    def sumcheck3(self, v1, v2, v3, prev_r):
        v1 = F(v1).reshape(-1, 2); v2 = F(v2).reshape(-1, 2); v3 = F(v3).reshape(-1, 2); pr = F(prev_r).reshape(2)
        n = v1.shape[0]; rounds = n.bit_length() - 1
        q = np.zeros((rounds, 4, 2), np.uint64); r = np.zeros((rounds, 2), np.uint64)
        vr = np.zeros((3, 2), np.uint64); fin = np.zeros(2, np.uint64)
        self.fn("sumcheck3")(_p(v1), _p(v2), _p(v3), c_sz(n), _p(pr), _p(q), _p(r), _p(vr), _p(fin))
        return dict(poly=q, r=r, vr=vr, fin=fin)


    def gate_claim(self, tables, a):
        ts = [F(t).reshape(-1, 2) for t in tables]; a = F(a).reshape(4, 2); o = np.zeros(2, np.uint64)
        arr = (ctypes.c_void_p * 6)(*[t.ctypes.data for t in ts])
        self.lib.orc_gate_claim(arr, c_sz(ts[0].shape[0]), _p(a), _p(o))
        return o

    def gate_sumcheck(self, tables, a, rand, claimed_sum):
        """degree-4 gate-consistency sumcheck (src/sumcheck.cpp:875-929); tables = (add, beta, L, R, O, mul), copied"""
        ts = [F(t).reshape(-1, 2).copy() for t in tables]; a = F(a).reshape(4, 2)
        n = ts[0].shape[0]; rounds = n.bit_length() - 1
        rnd = F(rand).reshape(2).copy(); sm = F(claimed_sum).reshape(2).copy()
        q = np.zeros((rounds, 5, 2), np.uint64); r = np.zeros((rounds, 2), np.uint64); fin = np.zeros((6, 2), np.uint64); chk = np.zeros(1, np.int32)
        self.lib.orc_gate_sumcheck(*[_p(t) for t in ts], c_sz(n), _p(a), _p(rnd), _p(sm), _p(q), _p(r), _p(fin), _p(chk))
        return dict(poly=q, r=r, fin=fin, rand=rnd, sum=sm, check=chk)

    def gate_standard(self, L, R, O, add, r):
        """prove_gate_consistency_standard through the gate sumcheck: a = (1, 1, 1, -1), rand = F(213), sum = 0, mul = 1 - add"""
        return gate_standard_via(self, L, R, O, add, r)[0]

    # ---- code-membership / FFT-as-sumcheck helpers
    def evaluate_parity_matrix(self, beta, n):
        b = F(beta).reshape(-1, 2)
        A = np.zeros_like(b)
        f = self.fn("evaluate_parity_matrix"); f.restype = ctypes.c_longlong
        ln = f(_p(b), c_sz(b.shape[0]), ctypes.c_longlong(n), _p(A))
        return A, ln

    def phi_g_init(self, rx, scale=(1, 0), ifft=False):
        r = F(rx).reshape(-1, 2); n = r.shape[0]
        sc = np.array(scale, np.uint64)
        o = np.zeros((1 << n, 2), np.uint64)
        self.fn("phi_g_init")(_p(r), ctypes.c_int(n), _p(sc), ctypes.c_int(int(ifft)), _p(o))
        return o

    def prepare_matrix(self, M, r):
        m = F(M); rows, cols = m.shape[0], m.shape[1]
        r = F(r).reshape(-1, 2)
        o = np.zeros((rows, 2), np.uint64)
        self.fn("prepare_matrix")(_p(m), c_sz(rows), c_sz(cols), _p(r), ctypes.c_int(r.shape[0]), _p(o))
        return o

    @staticmethod
    def _proof2(rounds):
        return (np.zeros((rounds, 3, 2), np.uint64), np.zeros((rounds, 2), np.uint64), np.zeros((2, 2), np.uint64), np.zeros(2, np.uint64))

    @staticmethod
    def claim_of(q0):
        """q(0) + q(1) for a quadratic (a,b,c): 2c + a + b, as python ints mod p"""
        a, b, c = [tuple(int(x) for x in q0[i]) for i in range(3)]
        return np.array([(a[0] + b[0] + 2 * c[0]) % P, (a[1] + b[1] + 2 * c[1]) % P], np.uint64)

    # ---- inner PCS commitments of the opening
    def shockwave_commit(self, poly, k):
        p = F(poly).reshape(-1, 2); N = p.shape[0]; W = 2 * N // k
        enc = np.zeros((k, W, 2), np.uint64); lv = np.zeros((2 * W, 32), np.uint8)
        f = self.fn("shockwave_commit"); f.restype = c_sz
        cnt = f(_p(p), c_sz(N), ctypes.c_int(k), _p(enc), _p(lv))
        return enc, lv[:cnt]

    def change_form(self, poly):
        p = F(poly).reshape(-1, 2).copy()
        self.fn("change_form")(_p(p), ctypes.c_int(p.shape[0].bit_length() - 1))
        return p

    def whir_commit(self, poly):
        p = F(poly).reshape(-1, 2); N = p.shape[0]
        com = np.zeros((2 * N, 2), np.uint64); lv = np.zeros((N, 32), np.uint8)
        f = self.fn("whir_commit"); f.restype = c_sz
        cnt = f(_p(p), c_sz(N), _p(com), _p(lv))
        return com, lv[:cnt]

    # ---- batched cubic sumcheck / multiplication tree
    def batch_3product_sumcheck(self, t1, t2, t3, lens, a):
        """tables concatenated (sum(lens), 2); returns dict(poly, r, vr)"""
        T = [F(x).reshape(-1, 2).copy() for x in (t1, t2, t3)]
        lens = np.ascontiguousarray(lens, np.uint64); av = F(a).reshape(-1, 2)
        rounds = int(max(lens)).bit_length() - 1; nb = len(lens)
        q = np.zeros((rounds, 4, 2), np.uint64); r = np.zeros((rounds, 2), np.uint64); vr = np.zeros((nb, 3, 2), np.uint64)
        f = self.fn("batch_3product_sumcheck"); f.restype = ctypes.c_int
        f(_p(T[0]), _p(T[1]), _p(T[2]), _p(lens), ctypes.c_int(nb), _p(av), _p(q), _p(r), _p(vr))
        return dict(poly=q, r=r, vr=vr)

    def mul_tree(self, inp, previous_r, prev_x=None):
        """inp (vectors, size, 2), both powers of two"""
        x = F(inp); vectors, size = x.shape[0], x.shape[1]
        lt = (vectors * size).bit_length() - 1; depth = size.bit_length() - 1
        nr = sum(range(lt))                                   # generous bound on total rounds
        q = np.zeros((nr + 1, 4, 2), np.uint64); r = np.zeros((nr + 1, 2), np.uint64)
        vr = np.zeros((depth, 3, 2), np.uint64); fin = np.zeros((depth, 2), np.uint64)
        final_r = np.zeros((lt, 2), np.uint64); oe = np.zeros(2, np.uint64); fe = np.zeros(2, np.uint64)
        pr = F(previous_r).reshape(2); px = F(prev_x).reshape(-1, 2) if prev_x is not None else None
        f = self.fn("mul_tree"); f.restype = ctypes.c_int
        layers = f(_p(x), c_sz(vectors), c_sz(size), _p(pr), _p(px) if px is not None else None, _p(q), _p(r), _p(vr), _p(fin), _p(final_r), _p(oe), _p(fe))
        # rounds per layer: top layer has log2(vectors) rounds (or 1 after the first scalar step), growing by one per layer
        return dict(layers=np.array([layers]), poly=q, r=r, vr=vr[:layers], fin=fin[:layers], final_r=final_r, out_eval=oe, final_eval=fe)

    # ---- streaming-sumcheck error terms / folds
    def err2p(self, b1, b2, f1, f2):
        a = [F(x).reshape(-1, 2) for x in (b1, b2, f1, f2)]
        K = np.zeros((2, 2), np.uint64)
        self.fn("err2p")(*[_p(x) for x in a], c_sz(a[0].shape[0]), _p(K))
        return K

    def set_lookups(self, lookup_rand):
        """has_lookups / lookup_rand[0..1] (src/main.cpp:67,70); None = off"""
        if lookup_rand is None:
            self.fn("set_lookups")(ctypes.c_int(0), None)
        else:
            lr = F(lookup_rand).reshape(-1, 2)[:2].copy()
            self.fn("set_lookups")(ctypes.c_int(1), _p(lr))

    def err3p(self, b1, gate, f1, f2, f3, beta):
        g = np.ascontiguousarray(gate, np.int32)
        a = [F(x).reshape(-1, 2) for x in (b1, f1, f2, f3, beta)]
        K = np.zeros((3, 2), np.uint64)
        self.fn("err3p")(_p(a[0]), _p(g), _p(a[1]), _p(a[2]), _p(a[3]), _p(a[4]), c_sz(a[0].shape[0]), _p(K))
        return K

    def err4p(self, b1, b2, b3, gate, f1, f2, f3, f4):
        g = np.ascontiguousarray(gate, np.int32)
        a = [F(x).reshape(-1, 2) for x in (b1, b2, b3, f1, f2, f3, f4)]
        K = np.zeros((4, 2), np.uint64)
        self.fn("err4p")(_p(a[0]), _p(a[1]), _p(a[2]), _p(g), _p(a[3]), _p(a[4]), _p(a[5]), _p(a[6]), c_sz(a[0].shape[0]), _p(K))
        return K

    def batch_prod(self, f1, f2, f3, b1, b2, b3, r_last, a, rem_beta, Kf, Kp):
        """one batch_prod step; tables (batches, n, 2); returns dict(rand, Kf, Kp, f1, f2, f3)"""
        f = [F(x).copy() for x in (f1, f2, f3)]; b = [F(x) for x in (b1, b2, b3)]
        batches, n = f[0].shape[0], f[0].shape[1]
        rl = F(r_last).reshape(2); av = F(a).reshape(-1, 2); rb = F(rem_beta).reshape(-1, 2)
        kf = F(Kf).reshape(2).copy(); kp = F(Kp).reshape(-1, 2).copy(); ro = np.zeros(2, np.uint64)
        self.fn("batch_prod")(_p(f[0]), _p(f[1]), _p(f[2]), _p(b[0]), _p(b[1]), _p(b[2]), ctypes.c_int(batches), c_sz(n), _p(rl), _p(av), _p(rb),
                              _p(kf), _p(kp), _p(ro))
        return dict(rand=ro, Kf=kf, Kp=kp, f1=f[0], f2=f[1], f3=f[2])

    def elastic_commit(self, N, B, opt):
        o = np.zeros((8 * B, 32), np.uint8)
        f = self.fn("elastic_commit"); f.restype = c_sz
        cnt = f(c_sz(N), c_sz(B), ctypes.c_int(opt), _p(o))
        return o[:cnt]



    # ---- Elastic_PC open, RS x RS: stream passes 2 and 3 (src/Elastic_PC.cpp:316-347, 487-533)
    def read_stream(self, B):
        o = np.zeros((B, 2), np.uint64)
        self.fn("read_stream")(c_sz(B), _p(o))
        return o

    def elastic_aggregate(self, N, B, beta):
        b = F(beta).reshape(-1, 2)
        aggr = np.zeros((B, 2), np.uint64); root = np.zeros(32, np.uint8)
        self.fn("elastic_aggregate")(c_sz(N), c_sz(B), _p(b), _p(aggr), _p(root))
        return aggr, root

    def elastic_reply(self, N, B, I):
        Iq = np.ascontiguousarray(I, np.uint64).reshape(-1, 2)
        reply = np.zeros((Iq.shape[0], N // B, 2), np.uint64)
        self.fn("elastic_reply")(c_sz(N), c_sz(B), _p(Iq), c_sz(Iq.shape[0]), _p(reply))
        return reply

    # ---- Elastic_PC open, RS x expander (option 2): aggregate()'s linear_time branch (src/Elastic_PC.cpp:348-413) and update_reply_spielman (:431-485).
    # expander_init_store(B >> 14) must have been called (test_Elastic_PC option 2 draws the graphs before the commit).
    def elastic_aggregate2(self, N, B, beta, I, want_tensor=False):
        b = F(beta).reshape(-1, 2); Iq = np.ascontiguousarray(I, np.uint64).reshape(-1, 2)
        trs = max(B >> 14, 1); nq = Iq.shape[0]
        aggr = np.zeros((B, 2), np.uint64); roots = np.zeros((2, 32), np.uint8)
        aux = np.zeros((nq, 2 * trs, 2), np.uint64)
        ten = np.zeros((trs, 2 * B // trs, 2), np.uint64) if want_tensor else None
        f = self.fn("elastic_aggregate2"); f.restype = c_sz
        nr = f(c_sz(N), c_sz(B), _p(b), _p(Iq), c_sz(nq), _p(aggr), _p(roots[0]), _p(roots[1]), _p(aux), _p(ten) if ten is not None else None)
        return dict(aggr=aggr, cf_root=roots[0].copy(), cc_root=roots[1].copy(), aux=aux[:nr].copy(), tensor=ten)

    def elastic_reply2(self, N, B, I):
        Iq = np.ascontiguousarray(I, np.uint64).reshape(-1, 2)
        reply = np.zeros((Iq.shape[0], N // B, 2), np.uint64)
        self.fn("elastic_reply2")(c_sz(N), c_sz(B), _p(Iq), c_sz(Iq.shape[0]), _p(reply))
        return reply

class Oracle(_Base):
    pfx = "orc_"

    def __init__(self, build=True):
        if build and not os.path.exists(ORACLE_SO):
            build_oracle()
        self.lib = ctypes.CDLL(ORACLE_SO)

    def fft(self, arr, inverse=False, cached=False):
        a = F(arr).reshape(-1, 2).copy()
        logn = a.shape[0].bit_length() - 1
        self.fn("fft_cached" if cached else "fft")(_p(a), ctypes.c_int(logn), ctypes.c_int(int(inverse)))
        return a

    def fft_cache_reset(self): self.lib.orc_fft_cache_reset()

    def commit_standard(self, poly, K, trs, lin, want_tensor=False):
        p = F(poly).reshape(-1, 2)
        N = p.shape[0]; M = N // K
        lv = np.zeros((2 * M, 32), np.uint8)
        t = np.zeros((K, 2 * trs, 2 * M // trs, 2), np.uint64) if want_tensor else None
        f = self.lib.orc_commit_standard; f.restype = c_sz
        cnt = f(_p(p), c_sz(N), ctypes.c_int(K), ctypes.c_int(trs), ctypes.c_int(lin), _p(lv), _p(t) if want_tensor else None)
        return lv[:cnt], t

    def commit_standard_mt(self, poly, K, trs, lin, threads, want_tensor=False):
        """commit_standard on `threads` threads (BASELINE.md 3.2(b)'s all-cores CPU leg): identical levels"""
        p = F(poly).reshape(-1, 2)
        N = p.shape[0]; M = N // K
        lv = np.zeros((2 * M, 32), np.uint8)
        t = np.zeros((K, 2 * trs, 2 * M // trs, 2), np.uint64) if want_tensor else None
        f = self.lib.orc_commit_standard_mt; f.restype = c_sz
        cnt = f(_p(p), c_sz(N), ctypes.c_int(K), ctypes.c_int(trs), ctypes.c_int(lin), ctypes.c_int(threads), _p(lv), _p(t) if want_tensor else None)
        return lv[:cnt], t

    def open_tree_blake(self, levels, n_leaves, col, row, columns):
        lv = np.ascontiguousarray(levels, dtype=np.uint8)
        path = np.zeros((64, 32), np.uint8)
        f = self.lib.orc_open_tree_blake; f.restype = ctypes.c_int
        d = f(_p(lv), c_sz(n_leaves), c_sz(col), c_sz(row), c_sz(columns), _p(path))
        return path[:d]

    def aggregate(self, poly, beta):
        p = F(poly).reshape(-1, 2); b = F(beta).reshape(-1, 2)
        K = b.shape[0]
        o = np.zeros((p.shape[0] // K, 2), np.uint64)
        self.lib.orc_aggregate(_p(p), c_sz(p.shape[0]), _p(b), ctypes.c_int(K), _p(o))
        return o


    def prove_linear_code(self, codeword, n, r1):
        cw = F(codeword).reshape(-1, 2); r1 = F(r1).reshape(-1, 2)
        rounds = cw.shape[0].bit_length() - 1
        q, r, vr, fin = self._proof2(rounds)
        self.lib.orc_prove_linear_code(_p(cw), c_sz(cw.shape[0]), ctypes.c_longlong(n), _p(r1), _p(q), _p(r), _p(vr), _p(fin))
        return dict(poly=q, r=r, vr=vr, fin=fin)

    def prove_fft(self, m, rr):
        m = F(m).reshape(-1, 2); rr = F(rr).reshape(-1, 2)
        rounds = (2 * m.shape[0]).bit_length() - 1
        q, r, vr, fin = self._proof2(rounds)
        self.lib.orc_prove_fft(_p(m), c_sz(m.shape[0]), _p(rr), _p(q), _p(r), _p(vr), _p(fin))
        return dict(poly=q, r=r[:rounds - 1], vr=vr, fin=fin)

    def prove_fft_matrix(self, M, rr):
        m = F(M); rows, cols = m.shape[0], m.shape[1]; rr = F(rr).reshape(-1, 2)
        rounds = (2 * cols).bit_length() - 1
        q, r, vr, fin = self._proof2(rounds)
        self.lib.orc_prove_fft_matrix(_p(m), c_sz(rows), c_sz(cols), _p(rr), _p(q), _p(r), _p(vr), _p(fin))
        return dict(poly=q, r=r, vr=vr, fin=fin)

    def open_core(self, poly, K, trs, x, queries, tensor=None):
        """open_standard + recursive_prover_Spielman without the inner shockwave/WHIR PCS"""
        p = F(poly).reshape(-1, 2); x = F(x).reshape(-1, 2)
        N = p.shape[0]; M = N // K; cols = 2 * M // trs
        R1 = (2 * trs).bit_length() - 1; logc = cols.bit_length() - 1
        rounds = R1 + logc + 2 * (R1 + logc) + logc
        I = np.zeros((queries, 2), np.uint32)
        reply = np.zeros((queries, K, 2), np.uint64)
        sc = np.zeros((5, 2), np.uint64); q = np.zeros((rounds, 3, 2), np.uint64); r = np.zeros((rounds, 2), np.uint64)
        vr = np.zeros((5, 2, 2), np.uint64); fin = np.zeros((5, 2), np.uint64); chk = np.zeros(3, np.int32)
        roots = np.zeros((2, 32), np.uint8)
        t = F(tensor) if tensor is not None else None
        f = self.lib.orc_open_core; f.restype = ctypes.c_int
        f(_p(p), c_sz(N), ctypes.c_int(K), ctypes.c_int(trs), _p(x), ctypes.c_int(queries), _p(I), _p(reply) if t is not None else None,
          _p(t) if t is not None else None, _p(sc), _p(q), _p(r), _p(vr), _p(fin), _p(chk), _p(roots))
        return dict(I=I, reply=reply, scalars=sc, poly=q, r=r, vr=vr, fin=fin, checks=chk, roots=roots)

    def open_standard_from_aggregate(self, aggr, K, trs, queries):
        """open_standard's prover side from a given aggregate (multi-GPU open): no replies (they come from the tensor shards)"""
        a = F(aggr).reshape(-1, 2); M = a.shape[0]; cols = 2 * M // trs
        R1 = (2 * trs).bit_length() - 1; logc = cols.bit_length() - 1; R3 = R1 + logc
        rounds = R1 + logc + 2 * R3 + logc
        I = np.zeros((queries, 2), np.uint32)
        sc = np.zeros((5, 2), np.uint64); q = np.zeros((rounds, 3, 2), np.uint64); r = np.zeros((rounds, 2), np.uint64)
        vr = np.zeros((5, 2, 2), np.uint64); fin = np.zeros((5, 2), np.uint64); chk = np.zeros(3, np.int32); roots = np.zeros((2, 32), np.uint8)
        f = self.lib.orc_open_core_aggr; f.restype = ctypes.c_int
        f(_p(a), c_sz(M), ctypes.c_int(K), ctypes.c_int(trs), ctypes.c_int(queries), _p(I), _p(sc), _p(q), _p(r), _p(vr), _p(fin), _p(chk), _p(roots))
        res = dict(I=I, scalars=sc, poly=q, r=r, vr=vr, fin=fin, checks=chk, roots=roots)
        C = self.compute_tensorcode(a, trs, 1).reshape(2 * trs, cols, 2)[trs:].reshape(-1, 2)
        o4 = R1 + logc + R3; r4 = r[o4:o4 + R3]; r5 = r[o4 + R3:o4 + R3 + logc]
        enc_c, lv_c = self.shockwave_commit(C, 32)
        res["sp_c"] = self.shockwave_prove(C, enc_c, 32, r4[:-1], lv_c)
        enc_f, lv_f = self.shockwave_commit(a, 32)
        res["sp_f"] = self.shockwave_prove(a, enc_f, 32, np.concatenate([r5, r4[logc:logc + R1 - 1]])[:-1], lv_f)
        return res

    def open_standard(self, poly, K, trs, x, queries, tensor=None):
        """Prover side of open_standard (src/Our_PC.cpp:604-661): open_core, then shockwave_prove(C_c, P4.r[:-1])
        (src/PC_utils.cpp:368) and shockwave_prove(C_f, P5.randomness[:-1]) (:385).  The libc generator runs on across the
        three calls as in the reference (shockwave_commit / aggregate / tensorcode draw nothing)."""
        p = F(poly).reshape(-1, 2); x = F(x).reshape(-1, 2)
        N = p.shape[0]; M = N // K; cols = 2 * M // trs
        R1 = (2 * trs).bit_length() - 1; logc = cols.bit_length() - 1; R3 = R1 + logc; logK = K.bit_length() - 1
        res = self.open_core(p, K, trs, x, queries, tensor)
        aggr = self.aggregate(p, self.precompute_beta(x[:logK]))
        C = self.compute_tensorcode(aggr, trs, 1).reshape(2 * trs, cols, 2)[trs:].reshape(-1, 2)
        o4 = R1 + logc + R3; r4 = res["r"][o4:o4 + R3]; r5 = res["r"][o4 + R3:o4 + R3 + logc]
        enc_c, lv_c = self.shockwave_commit(C, 32)
        res["sp_c"] = self.shockwave_prove(C, enc_c, 32, r4[:-1], lv_c)
        enc_f, lv_f = self.shockwave_commit(aggr, 32)
        x5 = np.concatenate([r5, r4[logc:logc + R1 - 1]])[:-1]
        res["sp_f"] = self.shockwave_prove(aggr, enc_f, 32, x5, lv_f)
        return res


    def aggregate_roots(self, poly, beta, trs):
        """_aggregate (src/Our_PC.cpp:258-289): the aggregate and the roots of C_f / C_c"""
        aggr = self.aggregate(poly, beta)
        M = aggr.shape[0]; cols = 2 * M // trs
        C = self.compute_tensorcode(aggr, trs, 1).reshape(2 * trs, cols, 2)[trs:].reshape(-1, 2)
        return aggr, np.stack([self.shockwave_commit(aggr, 32)[1][-1], self.shockwave_commit(C, 32)[1][-1]])

    def elastic_open(self, N, B, x, queries=700, commit_levels=None, want_reply=True):
        """Prover side of Elastic_PC::open, option 1 (src/Elastic_PC.cpp:625-726): orc_elastic_open_rs, then shockwave_prove(C_f, r_x)
        (src/PC_utils.cpp:507) with the libc generator running on, as Oracle.open_standard does for Our_PC."""
        x = F(x).reshape(-1, 2)
        trs = B >> 11; cols = 2 * B // trs; K = N // B
        logc = cols.bit_length() - 1; logr = (2 * trs).bit_length() - 1; logt = trs.bit_length() - 1
        maxr = (1024 * 2 * trs).bit_length() - 1 + logr + (logt + logc) + logc
        I = np.zeros((queries, 2), np.uint32); rv0 = np.zeros(2, np.uint64); aggr = np.zeros((B, 2), np.uint64); root = np.zeros(32, np.uint8)
        reply = np.zeros((queries, K, 2), np.uint64) if want_reply else None
        depth = (4 * B).bit_length() - 1
        lv = np.ascontiguousarray(commit_levels, np.uint8) if commit_levels is not None else None
        paths = np.zeros((queries, depth, 32), np.uint8) if lv is not None else None
        nc = ctypes.c_int()
        q = np.zeros((maxr, 3, 2), np.uint64); r = np.zeros((maxr, 2), np.uint64); vr = np.zeros((4, 2, 2), np.uint64); fin = np.zeros((4, 2), np.uint64)
        chk = np.zeros(2, np.int32); rx = np.zeros((logc + logt, 2), np.uint64)
        f = self.lib.orc_elastic_open_rs; f.restype = ctypes.c_int
        rounds = f(c_sz(N), c_sz(B), _p(x), ctypes.c_int(queries), _p(lv) if lv is not None else None, _p(I), _p(rv0), _p(aggr), _p(root),
                   _p(reply) if reply is not None else None, _p(paths) if paths is not None else None, ctypes.byref(nc), _p(q), _p(r), _p(vr), _p(fin), _p(chk), _p(rx))
        res = dict(I=I, rv0=rv0, aggr=aggr, cf_root=root, reply=reply, paths=paths, ncols=np.array([nc.value]), poly=q[:rounds], r=r[:rounds], vr=vr, fin=fin,
                   checks=chk, rx=rx)
        enc_f, lv_f = self.shockwave_commit(aggr, 32)
        res["sp_f"] = self.shockwave_prove(aggr, enc_f, 32, rx, lv_f)
        return res

    def elastic_reply2(self, N, B, I, stale_parity_quirk=1):
        Iq = np.ascontiguousarray(I, np.uint64).reshape(-1, 2)
        reply = np.zeros((Iq.shape[0], N // B, 2), np.uint64)
        f = self.lib.orc_elastic_reply2; f.restype = c_sz
        filled = f(c_sz(N), c_sz(B), _p(Iq), c_sz(Iq.shape[0]), _p(reply), ctypes.c_int(stale_parity_quirk))
        return reply[:, :filled]

    def elastic_open2(self, N, B, x, queries=5900, commit_levels=None, want_reply=True, stale_parity_quirk=1, prove=True):
        """Prover side of Elastic_PC::open, option 2 (src/Elastic_PC.cpp:625-726 under linear_time): orc_elastic_open_spielman, then
        shockwave_prove(C_c, P3.randomness[0]) and shockwave_prove(C_f, r_x) (src/PC_utils.cpp:252, 269) with the libc generator running on
        (nothing between them draws).  expander_init_store(B >> 14) must have been called."""
        x = F(x).reshape(-1, 2)
        trs = B >> 14; cols = 2 * B // trs; K = N // B
        logc = cols.bit_length() - 1; R1 = (2 * trs).bit_length() - 1; logt = R1 - 1
        maxr = R1 + logc + (queries * 2 * trs).bit_length() + logc
        I = np.zeros((queries, 2), np.uint32); rv0 = np.zeros(2, np.uint64); aggr = np.zeros((B, 2), np.uint64); roots = np.zeros((2, 32), np.uint8)
        reply = np.zeros((queries, K, 2), np.uint64) if want_reply else None
        depth = (4 * B).bit_length() - 1
        lv = np.ascontiguousarray(commit_levels, np.uint8) if commit_levels is not None else None
        paths = np.zeros((queries, depth, 32), np.uint8) if lv is not None else None
        nr = ctypes.c_int(); aux = np.zeros((queries, 2 * trs, 2), np.uint64); scal = np.zeros((3, 2), np.uint64)
        q = np.zeros((maxr, 3, 2), np.uint64); r = np.zeros((maxr, 2), np.uint64); vr = np.zeros((4, 2, 2), np.uint64); fin = np.zeros((4, 2), np.uint64)
        chk = np.zeros(1, np.int32); rx = np.zeros((logc + logt - 1, 2), np.uint64)
        f = self.lib.orc_elastic_open_spielman; f.restype = ctypes.c_int
        rounds = f(c_sz(N), c_sz(B), _p(x), ctypes.c_int(queries), _p(lv) if lv is not None else None, ctypes.c_int(stale_parity_quirk), _p(I), _p(rv0), _p(aggr), _p(roots),
                   _p(reply) if reply is not None else None, _p(paths) if paths is not None else None, ctypes.byref(nr), _p(aux), _p(scal), _p(q), _p(r), _p(vr), _p(fin),
                   _p(chk), _p(rx))
        aux = aux[:nr.value].copy()
        res = dict(I=I, rv0=rv0, aggr=aggr, cf_root=roots[0].copy(), cc_root=roots[1].copy(), reply=reply, paths=paths, nr=np.array([nr.value]), aux=aux, scal=scal,
                   poly=q[:rounds], r=r[:rounds], vr=vr, fin=fin, checks=chk, rx=rx)
        if prove:
            npad = 1 << (max(aux.size // 2, 1) - 1).bit_length()
            flat = np.zeros((npad, 2), np.uint64); flat[:aux.size // 2] = aux.reshape(-1, 2)
            R3 = npad.bit_length() - 1
            p3r = r[R1 + logc:R1 + logc + R3]
            enc_c, lv_c = self.shockwave_commit(flat, 32)
            res["sp_c"] = self.shockwave_prove(flat, enc_c, 32, p3r, lv_c)
            enc_f, lv_f = self.shockwave_commit(aggr, 32)
            res["sp_f"] = self.shockwave_prove(aggr, enc_f, 32, rx, lv_f)
        return res

    def open_standard_rs(self, poly, K, trs, x, queries=790, commit_levels=None, tensor=None):
        """Prover side of Our_PC open_standard with linear_time == false (test_PC option 1, src/Our_PC.cpp:604-692 + recursive_prover_RS,
        src/PC_utils.cpp:396-512): orc_open_standard_rs, then shockwave_prove(C_f, r_x) with the libc generator running on."""
        poly = F(poly).reshape(-1, 2); x = F(x).reshape(-1, 2)
        N = poly.shape[0]; B = N // K; cols = 2 * B // trs
        logc = cols.bit_length() - 1; logr = (2 * trs).bit_length() - 1; logt = trs.bit_length() - 1
        maxr = (1024 * 2 * trs).bit_length() - 1 + logr + (logt + logc) + logc
        I = np.zeros((queries, 2), np.uint32); rv0 = np.zeros(2, np.uint64); aggr = np.zeros((B, 2), np.uint64); root = np.zeros(32, np.uint8)
        T = F(tensor) if tensor is not None else None
        reply = np.zeros((queries, K, 2), np.uint64) if T is not None else None
        depth = B.bit_length() - 1
        lv = np.ascontiguousarray(commit_levels, np.uint8) if commit_levels is not None else None
        paths = np.zeros((queries, depth, 32), np.uint8) if lv is not None else None
        nc = ctypes.c_int()
        q = np.zeros((maxr, 3, 2), np.uint64); r = np.zeros((maxr, 2), np.uint64); vr = np.zeros((4, 2, 2), np.uint64); fin = np.zeros((4, 2), np.uint64)
        chk = np.zeros(2, np.int32); rx = np.zeros((logc + logt, 2), np.uint64)
        f = self.lib.orc_open_standard_rs; f.restype = ctypes.c_int
        rounds = f(_p(poly), c_sz(N), ctypes.c_int(K), ctypes.c_int(trs), _p(x), ctypes.c_int(queries), _p(lv) if lv is not None else None, _p(T) if T is not None else None,
                   _p(I), _p(rv0), _p(aggr), _p(root), _p(reply) if reply is not None else None, _p(paths) if paths is not None else None, ctypes.byref(nc),
                   _p(q), _p(r), _p(vr), _p(fin), _p(chk), _p(rx))
        res = dict(I=I, rv0=rv0, aggr=aggr, cf_root=root, reply=reply, paths=paths, ncols=np.array([nc.value]), poly=q[:rounds], r=r[:rounds], vr=vr, fin=fin,
                   checks=chk, rx=rx)
        enc_f, lv_f = self.shockwave_commit(aggr, 32)
        res["sp_f"] = self.shockwave_prove(aggr, enc_f, 32, rx, lv_f)
        return res


    # ---- streaming multiplication-tree prover (src/sumcheck.cpp:1014-1054, 1150-1393; src/witness_stream.cpp:2413-2510)
    def stream_config(self, kind=0, seed=0):
        """kind 0: the reference's default stream (every read alike); kind 1: read c = splitmix_field(n, seed + c) (tests only)"""
        self.lib.orc_stream_config(ctypes.c_int(kind), ctypes.c_uint64(seed))

    def read_mul_tree_layer(self, fd_size, size, layer):
        o = np.zeros((size, 2), np.uint64)
        self.lib.orc_read_mul_tree_layer(c_sz(size), ctypes.c_int(layer), _p(o))
        return o

    def read_mul_tree_data(self, fd_size, size, layer, distance, batches):
        tot = sum(size >> (i * distance) for i in range(batches))
        o = np.zeros((tot, 2), np.uint64)
        self.lib.orc_read_mul_tree_data(c_sz(size), ctypes.c_int(layer), ctypes.c_int(distance), ctypes.c_int(batches), _p(o))
        return o

    def sumcheck3_stream_batch(self, fd_size, B, r, batches, distance, layer_id, old_claims, full=False):
        """generate_3product_sumcheck_beta_stream_batch_optimized; r: (batches, rlen, 2).  Returns dict(new_claims, new_r[, transcripts])"""
        r = F(r); rlen = r.shape[1]; oc = F(old_claims).reshape(-1, 2)
        size = fd_size >> layer_id; logB = B.bit_length() - 1; nR = size // (2 * B); lR = nR.bit_length() - 1
        ld = 1 + logB + lR
        nc = np.zeros((batches, 2), np.uint64); nr = np.zeros((batches, ld, 2), np.uint64)
        c1 = np.zeros((logB, 4, 2), np.uint64); r1 = np.zeros((logB, 2), np.uint64); vr1 = np.zeros((batches, 3, 2), np.uint64)
        q2 = np.zeros((lR, 3, 2), np.uint64); r2 = np.zeros((lR, 2), np.uint64); vr2 = np.zeros((2, 2), np.uint64); f2 = np.zeros(2, np.uint64)
        R = np.zeros((nR, 2), np.uint64); chk = np.zeros(3, np.int32)
        f = self.lib.orc_sumcheck3_stream_batch; f.restype = ctypes.c_int
        f(c_sz(fd_size), c_sz(B), _p(r), ctypes.c_int(rlen), ctypes.c_int(batches), ctypes.c_int(distance), ctypes.c_int(layer_id), _p(oc), ctypes.c_int(oc.shape[0]),
          _p(nc), _p(nr), ctypes.c_int(ld), _p(c1), _p(r1), _p(vr1), _p(q2), _p(r2), _p(vr2), _p(f2), _p(R), _p(chk))
        rows = [nr[i, :1 + logB - i * distance + lR].copy() for i in range(batches)]
        out = dict(new_claims=nc, new_r=rows)
        if full:
            out.update(cpoly1=c1, r1=r1, vr1=vr1, qpoly2=q2, r2=r2, vr2=vr2, fin2=f2, R=R, checks=chk)
        return out



    def commit_layers(self, fd_size, B, batches, layer_id, distance):
        """commit_layers (src/sumcheck.cpp:983-1003): roots of the Elastic commitments to the PC_layer streams; returns (sizes, layers, roots)"""
        size = fd_size >> layer_id
        sizes, layers, roots = [], [], []
        for i in range(max(batches - 1, 0)):
            sz = size >> (distance * i); ly = layer_id + i * distance
            sizes.append(sz); layers.append(ly)
            if sz > B:
                lv = np.zeros((8 * B, 32), np.uint8)
                f = self.lib.orc_elastic_commit_pc_layer; f.restype = c_sz
                cnt = f(c_sz(sz), c_sz(B), ctypes.c_int(ly), _p(lv))
                roots.append(lv[cnt - 1].copy())
            else:
                roots.append(np.zeros(32, np.uint8))
        return np.array(sizes, np.uint64), np.array(layers, np.uint64), np.stack(roots) if roots else np.zeros((0, 32), np.uint8)

    def generate_claims_opt(self, fd_size, B, r, batches, layer_id, distance):
        r = F(r).reshape(-1, 2); c = np.zeros((batches, 2), np.uint64)
        self.lib.orc_generate_claims_opt(c_sz(fd_size), c_sz(B), _p(r), ctypes.c_int(batches), ctypes.c_int(layer_id), ctypes.c_int(distance), _p(c))
        return c

    def field_prod(self, v):
        """product of the entries of v (n a power of two)"""
        v = F(v).reshape(-1, 2).copy()
        while v.shape[0] > 1:
            v = self.f_mul(v[0::2], v[1::2])
        return v[0]

    def mul_tree_stream_shallow(self, fd_size, B, vectors, size, previous_r, distance, prev_x, naive=True):
        """prove_multiplication_tree_stream_shallow (src/sumcheck.cpp:1746-1915) without commit_layers / open_layers (they end in
        Elastic_PC commit / open over "PC_layer" streams): composition of the restated pieces, libc draws in the reference's order.
        Returns dict(output, tree=<mul_tree transcript>, steps=[per streaming sumcheck: the full transcript dict])."""
        assert size * vectors > 2 * B, "in-memory case (src/sumcheck.cpp:1755-1774): call mul_tree on read_stream's data directly"
        layers = ((size * vectors) // (2 * B)).bit_length() - 1
        if layers % distance != 0 and layers > distance:
            layers = distance + layers - (layers % distance)
        n1 = fd_size >> layers
        buff1 = self.read_mul_tree_layer(fd_size, n1, layers)
        inp = buff1.reshape(vectors, n1 // vectors, 2)
        tree = self.mul_tree(inp, previous_r, prev_x)
        out = dict(output=np.stack([self.field_prod(inp[i]) for i in range(vectors)]), tree=tree, steps=[], layers=layers)
        if layers == 0:
            return out
        lt = n1.bit_length() - 1
        old_claims = tree["final_eval"].reshape(1, 2); old_r = [tree["final_r"][:lt]]
        if layers <= distance or naive:
            for i in range(layers - 1, -1, -1):
                st = self.sumcheck3_stream_batch(fd_size, B, np.stack(old_r), 1, 1, i, old_claims, full=True)
                out["steps"].append(st); old_claims = st["new_claims"]; old_r = st["new_r"]
        else:
            batches = layers // distance
            extra = self.generate_randomness(((size * vectors) >> distance).bit_length() - 1 - lt)
            r_temp = np.concatenate([tree["final_r"][:lt], extra])
            old_r = [r_temp] * batches
            old_claims = self.generate_claims_opt(fd_size, B, r_temp, batches, distance - 1, distance)
            out["claims0"] = old_claims
            for i in range(distance - 1, -1, -1):
                ln = min(len(x) for x in old_r)
                rr = np.stack([np.concatenate([x, np.zeros((max(len(y) for y in old_r) - len(x), 2), np.uint64)]) for x in old_r])
                st = self.sumcheck3_stream_batch(fd_size, B, rr, batches, distance, i, old_claims, full=True)
                out["steps"].append(st); old_claims = st["new_claims"]; old_r = st["new_r"]
        return out

    def gate_consistency_stream(self, L, R, O, S, B, r):
        """prove_gate_consistency (src/sumcheck.cpp:796-975) over a caller-supplied trace (chunks of B)"""
        L, R, O = [F(v).reshape(-1, 2) for v in (L, R, O)]; S = np.ascontiguousarray(S, np.int32); r = F(r).reshape(-1, 2)
        nch = L.shape[0] // B; logB = B.bit_length() - 1; lR = nch.bit_length() - 1
        out = dict(R=np.zeros((nch, 2), np.uint64), a=np.zeros((4, 2), np.uint64), poly=np.zeros((logB, 5, 2), np.uint64), gr=np.zeros((logB, 2), np.uint64),
                   fin6=np.zeros((6, 2), np.uint64), Peval=np.zeros((6, nch, 2), np.uint64), b=np.zeros((6, 2), np.uint64), q2=np.zeros((lR, 3, 2), np.uint64),
                   r2=np.zeros((lR, 2), np.uint64), vr2=np.zeros((2, 2), np.uint64), fin2=np.zeros(2, np.uint64), checks=np.zeros(3, np.int32))
        self.lib.orc_gate_consistency_stream(_p(L), _p(R), _p(O), _p(S), c_sz(nch), c_sz(B), _p(r), *[_p(out[k]) for k in ("R", "a", "poly", "gr", "fin6", "Peval", "b", "q2", "r2", "vr2", "fin2", "checks")])
        return out

    def gate_consistency_lookups_stream(self, L, R, O, S, B, r, lookup_rand):
        """prove_gate_consistency_lookups (src/sumcheck.cpp:503-795) over a caller-supplied trace; S in {0, 1, 2}; lookup_rand: 2 elements"""
        L, R, O = [F(v).reshape(-1, 2) for v in (L, R, O)]; S = np.ascontiguousarray(S, np.int32); r = F(r).reshape(-1, 2); lr = F(lookup_rand).reshape(2, 2)
        nch = L.shape[0] // B; logB = B.bit_length() - 1; lR = nch.bit_length() - 1
        out = dict(R=np.zeros((nch, 2), np.uint64), a=np.zeros((5, 2), np.uint64), poly=np.zeros((logB, 5, 2), np.uint64), gr=np.zeros((logB, 2), np.uint64),
                   fin9=np.zeros((9, 2), np.uint64), Peval=np.zeros((8, nch, 2), np.uint64), b=np.zeros((8, 2), np.uint64), q2=np.zeros((lR, 3, 2), np.uint64),
                   r2=np.zeros((lR, 2), np.uint64), vr2=np.zeros((2, 2), np.uint64), fin2=np.zeros(2, np.uint64), checks=np.zeros(5, np.int32))
        self.lib.orc_gate_consistency_lookups_stream(_p(L), _p(R), _p(O), _p(S), c_sz(nch), c_sz(B), _p(r), _p(lr),
                                                     *[_p(out[k]) for k in ("R", "a", "poly", "gr", "fin9", "Peval", "b", "q2", "r2", "vr2", "fin2", "checks")])
        return out

    class _WQ(ctypes.Structure):
        _fields_ = [(n, ctypes.c_void_p) for n in ("qidx", "qreply", "qpaths", "final_pb", "nq")]

    @staticmethod
    def _wq_buffers():
        b = dict(qidx=np.zeros(256, np.int32), qreply=np.zeros((256, 16, 2), np.uint64), qpaths=np.zeros(256 * 24 * 32, np.uint8),
                 final_pb=np.zeros((32, 2), np.uint64), nq=np.zeros(8, np.int32))
        return b, Oracle._WQ(*[b[n].ctypes.data for n in ("qidx", "qreply", "qpaths", "final_pb", "nq")])

    @staticmethod
    def whir_query_trim(b, N, iters):
        """cut the over-allocated query buffers to what `iters` rounds on an N-coefficient polynomial produce"""
        nq = b["nq"][:iters].copy(); tot = int(nq.sum())
        pbytes = sum(int(nq[t]) * 32 * (((2 * N) >> t).bit_length() - 1 - 2) for t in range(iters))
        rem = N >> (4 * iters)
        return dict(qn=nq, qidx=b["qidx"][:tot].copy(), qreply=b["qreply"][:tot].copy(), qpaths=b["qpaths"][:pbytes].copy(), final_pb=b["final_pb"][:2 * rem].copy())

    def whir_prove(self, poly, x, com=None, com_levels=None):
        p = F(poly).reshape(-1, 2); x = F(x).reshape(-1, 2); N = p.shape[0]
        logN = N.bit_length() - 1
        q = np.zeros((logN + 8, 3, 2), np.uint64); a = np.zeros((logN + 8, 2), np.uint64); roots = np.zeros((logN, 32), np.uint8)
        sc = np.zeros((2, 2), np.uint64); chk = np.zeros(2, np.int32)
        if com is None:
            com, com_levels = self.whir_commit(p)
        com = F(com).reshape(-1, 2); lv = np.ascontiguousarray(com_levels, np.uint8)
        qb, Q = self._wq_buffers()
        f = self.lib.orc_whir_prove_ex; f.restype = ctypes.c_int
        it = f(_p(p), c_sz(N), _p(x), _p(com), _p(lv), _p(q), _p(a), _p(roots), _p(sc), _p(chk), ctypes.byref(Q))
        res = dict(iters=np.array([it]), poly=q[:4 * it], a=a[:4 * it], roots=roots[:it], scal=sc, checks=chk)
        res.update(self.whir_query_trim(qb, N, it))
        return res

    def shockwave_prove(self, matrix, enc, k, x, levels=None):
        m = F(matrix).reshape(-1, 2); e = F(enc).reshape(-1, 2); x = F(x).reshape(-1, 2)
        N = m.shape[0]; w = N // k; W = 2 * w; lgW = W.bit_length() - 1; lw = w.bit_length() - 1
        if levels is None:
            _, levels = self.shockwave_commit(m, k)
        lv = np.ascontiguousarray(levels, np.uint8)
        I = np.zeros(240, np.uint32)
        q1 = np.zeros((lgW, 3, 2), np.uint64); r1 = np.zeros((lgW, 2), np.uint64); vr1 = np.zeros((2, 2), np.uint64); f1 = np.zeros(2, np.uint64)
        q2 = np.zeros((lgW, 3, 2), np.uint64); r2 = np.zeros((lgW, 2), np.uint64); vr2 = np.zeros((2, 2), np.uint64); f2 = np.zeros(2, np.uint64)
        wq = np.zeros((lw + 8, 3, 2), np.uint64); wa = np.zeros((lw + 8, 2), np.uint64); wr = np.zeros((lw + 1, 32), np.uint8)
        ws = np.zeros((2, 2), np.uint64); wc = np.zeros(2, np.int32); wroot = np.zeros(32, np.uint8)
        reply = np.zeros((240, k, 2), np.uint64); paths = np.zeros((240, lgW, 32), np.uint8)
        qb, Q = self._wq_buffers()
        f = self.lib.orc_shockwave_prove_ex; f.restype = ctypes.c_int
        it = f(_p(m), _p(e), _p(lv), c_sz(N), ctypes.c_int(k), _p(x), ctypes.c_int(x.shape[0]), _p(I), _p(q1), _p(r1), _p(vr1), _p(f1), _p(q2), _p(r2), _p(vr2), _p(f2),
               _p(wq), _p(wa), _p(wr), _p(ws), _p(wc), _p(wroot), _p(reply), _p(paths), ctypes.byref(Q))
        res = dict(I=I, q1=q1, r1=r1, vr1=vr1, fin1=f1, q2=q2, r2=r2[:lgW - 1], vr2=vr2, fin2=f2, iters=np.array([it]), wq=wq[:4 * it], wa=wa[:4 * it],
                   wroots=wr[:it], wscal=ws, wchecks=wc, whir_root=wroot, reply=reply, paths=paths)
        if it:
            res.update(self.whir_query_trim(qb, w, it))
        else:
            res.update(qn=np.zeros(0, np.int32), qidx=np.zeros(0, np.int32), qreply=np.zeros((0, 16, 2), np.uint64), qpaths=np.zeros(0, np.uint8), final_pb=np.zeros((0, 2), np.uint64))
        return res

    def read_stream_pc(self, B):
        o = np.zeros((B, 2), np.uint64)
        self.lib.orc_read_stream_pc(c_sz(B), _p(o))
        return o

    def time_commit_standard(self, N, K):
        f = self.lib.orc_time_commit_standard; f.restype = ctypes.c_double
        return f(c_sz(N), ctypes.c_int(K))

    def time_commit_standard_mt(self, N, K, threads):
        f = self.lib.orc_time_commit_standard_mt; f.restype = ctypes.c_double
        return f(c_sz(N), ctypes.c_int(K), ctypes.c_int(threads))


def _path_ps(n_leaves, depth, pos):
    """proof-size accounting of verify_claim_opt_blake (src/merkle_tree.cpp:326-361): 32 B per sibling not yet seen"""
    visited = set(); ps = 0.0
    for p in pos:
        pe = n_leaves + int(p)
        for _ in range(depth):
            if (pe ^ 1) in visited:
                break
            visited.add(pe ^ 1); pe //= 2; visited.add(pe)
            ps += 32.0 / 1024.0
    return ps


def _sc_ps(rounds):
    return rounds * 3 * 16.0 / 1024.0 + 2 * 16.0 / 1024.0                 # src/sumcheck.cpp:2431, 2449


def _shockwave_ps(sp, N, k):
    """ps of one shockwave_prove (src/Virgo.cpp:435-517) from its transcript (Oracle.shockwave_prove / Hobbit.shockwave_prove)"""
    w = N // k; W = 2 * w; lgW = W.bit_length() - 1
    ps = _sc_ps(lgW) + _sc_ps(lgW)
    it = int(sp["iters"][0])
    if w > 256:                # :479 reads aggr.size()/2 after prove_fft doubled aggr in place (src/sumcheck.cpp:2984-2985)
        q = 0
        for t in range(1, it + 1):
            ps += 4 * 3 * 16.0 / 1024.0
            size = 2 * w if t == 1 else (2 * w) >> (t - 1)
            n = int(sp["qn"][t - 1])
            if t == it:
                ps += (w >> (4 * it)) * 2 * 16.0 / 1024.0
            ps += 16.0 * n * 16.0 / 1024.0
            ps += _path_ps(size // 4, (size // 4).bit_length() - 1, sp["qidx"][q:q + n]); q += n
    else:
        ps += 2 * w * 16.0 / 1024.0        # :482: aggr.size() is the doubled vector
    ps += 240.0 * k * 16.0 / 1024.0
    return ps + _path_ps(W, lgW, sp["I"])


def open_proof_size(res, N, K, trs, queries=5900):
    """The `ps` (KB) the reference's open_standard accumulates and test_PC prints (src/Our_PC.cpp:653, 676-683; src/PC_utils.cpp:310-385;
    src/Virgo.cpp:435-686), recomputed from an open transcript: it depends on every query index drawn from libc across the whole
    open (through the Merkle-path de-duplication), so it fingerprints the RNG stream end to end."""
    M = N // K; cols = 2 * M // trs
    R1 = (2 * trs).bit_length() - 1; logc = cols.bit_length() - 1; R3 = R1 + logc
    ps = queries * K * 16.0 / 1024.0
    ps += _sc_ps(R1) + _sc_ps(logc) + _sc_ps(R3) + _sc_ps(R3)
    ps += _shockwave_ps(res["sp_c"], trs * cols, 32)
    ps += _sc_ps(logc)
    ps += _shockwave_ps(res["sp_f"], M, 32)
    pos = (res["I"][:, 1].astype(np.int64) // 4) * cols + res["I"][:, 0].astype(np.int64)
    return ps + _path_ps(M, M.bit_length() - 1, pos)


def elastic_open_proof_size(res, N, B, queries=700):
    """The `ps` (KB) Elastic_PC::open accumulates for option 1 (src/Elastic_PC.cpp:701, 716; src/PC_utils.cpp:474-507), from a transcript"""
    trs = B >> 11; cols = 4096
    logr = (2 * trs).bit_length() - 1; logt = logr - 1
    nc = int(res["ncols"][0]); np2 = 1 << max(nc - 1, 0).bit_length()
    ps = queries * res["reply"].shape[1] * 16.0 / 1024.0
    ps += _sc_ps((np2 * 2 * trs).bit_length() - 1) + _sc_ps(logr) + _sc_ps(logt + 12) + _sc_ps(12)
    ps += _shockwave_ps(res["sp_f"], B, 32)
    pos = (res["I"][:, 1].astype(np.int64) // 4) * cols + res["I"][:, 0].astype(np.int64)
    return ps + _path_ps(4 * B, (4 * B).bit_length() - 1, pos)


def elastic_open2_proof_size(res, N, B, queries=5900):
    """The `ps` (KB) Elastic_PC::open accumulates for option 2 (src/Elastic_PC.cpp:701, 716; src/PC_utils.cpp:221-269), from a transcript"""
    trs = B >> 14; cols = 2 * B // trs
    R1 = (2 * trs).bit_length() - 1; logc = cols.bit_length() - 1
    npad = 1 << (max(int(res["nr"][0]) * 2 * trs, 1) - 1).bit_length()
    ps = queries * res["reply"].shape[1] * 16.0 / 1024.0
    ps += _sc_ps(R1) + _sc_ps(logc) + _sc_ps(npad.bit_length() - 1)
    ps += _shockwave_ps(res["sp_c"], npad, 32)
    ps += _sc_ps(logc)
    ps += _shockwave_ps(res["sp_f"], B, 32)
    pos = (res["I"][:, 1].astype(np.int64) // 4) * cols + res["I"][:, 0].astype(np.int64)
    return ps + _path_ps(4 * B, (4 * B).bit_length() - 1, pos)


def gate_standard_inputs(n, seed):
    """consistent gates: selector s in {0,1}; O = L + R where s = 1, L * R where s = 0 (so the claimed sum 0 holds)"""
    P = (1 << 61) - 1
    rng = np.random.default_rng(seed)
    sel = rng.integers(0, 2, n).astype(np.uint64)
    Lr = rng.integers(0, 1 << 31, n).astype(np.uint64); Rr = rng.integers(0, 1 << 31, n).astype(np.uint64)
    z = np.zeros(n, np.uint64)
    L = np.stack([Lr, z], 1); R = np.stack([Rr, z], 1)
    prod = np.array([(int(a) * int(b)) % P for a, b in zip(Lr, Rr)], np.uint64)
    O = np.stack([np.where(sel == 1, (Lr + Rr) % np.uint64(P), prod), z], 1)
    add = np.stack([sel, z], 1)
    return L, R, O, add


def gate_lookup_inputs(n, seed):
    """consistent gates for the lookup variant (src/sumcheck.cpp:503-795): selector 0 -> O = L + R, 1 -> O = L * R, 2 -> lookup gate (any O: the
    lookup term cancels it identically).  Returns L, R, O (n x 2 uint64) and S (int32)."""
    P = (1 << 61) - 1
    rng = np.random.default_rng(seed)
    S = rng.integers(0, 3, n).astype(np.int32)
    Lr = rng.integers(0, 1 << 31, n).astype(np.uint64); Rr = rng.integers(0, 1 << 31, n).astype(np.uint64)
    z = np.zeros(n, np.uint64)
    prod = np.array([(int(a) * int(b)) % P for a, b in zip(Lr, Rr)], np.uint64)
    anyO = rng.integers(0, P, n).astype(np.uint64)
    O = np.stack([np.where(S == 0, (Lr + Rr) % np.uint64(P), np.where(S == 1, prod, anyO)), np.where(S == 2, rng.integers(0, P, n).astype(np.uint64), z)], 1)
    return np.stack([Lr, z], 1), np.stack([Rr, z], 1), O, S


def gate_standard_via(be, L, R, O, add, r):
    """prove_gate_consistency_standard on any backend `be` that has precompute_beta, f_sub and gate_sumcheck"""
    L, R, O, add = [F(v).reshape(-1, 2) for v in (L, R, O, add)]
    P = (1 << 61) - 1
    one = np.zeros_like(add); one[:, 0] = 1
    mul = np.stack([(one[:, 0] + np.uint64(P) - add[:, 0]) % np.uint64(P), (np.uint64(P) - add[:, 1]) % np.uint64(P)], 1).astype(np.uint64)
    beta = be.precompute_beta(r)
    a = np.array([[1, 0], [1, 0], [1, 0], [P - 1, 0]], np.uint64)
    res = be.gate_sumcheck((add, beta, L, R, O, mul), a, np.array([213, 0], np.uint64), np.array([0, 0], np.uint64))
    return np.stack([res["fin"][2], res["fin"][3], res["fin"][4], res["fin"][0]]), res


class Ref(_Base):
    pfx = "ref_"

    def __init__(self):
        self.lib = _dlopen_lazy(REF_SO)
        self.lib.ref_init()

    def fft(self, arr, inverse=False, cached=False):
        a = F(arr).reshape(-1, 2).copy()
        logn = a.shape[0].bit_length() - 1
        self.fn("fft_raw" if cached else "fft_vec")(_p(a), ctypes.c_int(logn), ctypes.c_int(int(inverse)))
        return a

    def encode_reset_scratch(self): self.lib.ref_encode_reset_scratch()

    def commit_standard(self, poly, K, trs, lin, want_tensor=False):
        p = F(poly).reshape(-1, 2)
        N = p.shape[0]; M = N // K
        lv = np.zeros((2 * M, 32), np.uint8)
        f = self.lib.ref_commit_standard; f.restype = c_sz
        cnt = f(_p(p), c_sz(N), ctypes.c_int(K), ctypes.c_int(trs), ctypes.c_int(lin), _p(lv))
        t = None
        if want_tensor:
            t = np.zeros((K, 2 * trs, 2 * M // trs, 2), np.uint64)
            for i in range(K):
                for r in range(2 * trs):
                    self.lib.ref_tensor_row(ctypes.c_int(i), ctypes.c_int(r), _p(t[i, r]))
        return lv[:cnt], t

    def open_tree_blake(self, col, row, columns):
        path = np.zeros((64, 32), np.uint8)
        f = self.lib.ref_open_tree_blake; f.restype = ctypes.c_int
        d = f(c_sz(col), c_sz(row), ctypes.c_int(columns), _p(path))
        return path[:d]

    def release_commit(self): self.lib.ref_release_commit()

    def aggregate(self, poly, beta, trs=16, lin=0):
        p = F(poly).reshape(-1, 2); b = F(beta).reshape(-1, 2)
        K = b.shape[0]
        o = np.zeros((p.shape[0] // K, 2), np.uint64)
        self.lib.ref_aggregate(_p(p), c_sz(p.shape[0]), _p(b), ctypes.c_int(K), ctypes.c_int(trs), ctypes.c_int(lin), _p(o))
        return o



    def aggregate_roots(self, poly, beta, trs):
        p = F(poly).reshape(-1, 2); b = F(beta).reshape(-1, 2)
        K = b.shape[0]
        o = np.zeros((p.shape[0] // K, 2), np.uint64); roots = np.zeros((2, 32), np.uint8)
        self.lib.ref_aggregate_roots(_p(p), c_sz(p.shape[0]), _p(b), ctypes.c_int(K), ctypes.c_int(trs), _p(o), _p(roots))
        return o, roots


    def read_mul_tree_layer(self, fd_size, size, layer):
        o = np.zeros((size, 2), np.uint64)
        self.lib.ref_read_mul_tree_layer(c_sz(fd_size), c_sz(size), ctypes.c_int(layer), _p(o))
        return o

    def read_mul_tree_data(self, fd_size, size, layer, distance, batches):
        tot = sum(size >> (i * distance) for i in range(batches))
        o = np.zeros((tot, 2), np.uint64)
        self.lib.ref_read_mul_tree_data(c_sz(fd_size), c_sz(size), ctypes.c_int(layer), ctypes.c_int(distance), ctypes.c_int(batches), _p(o))
        return o

    def generate_claims_opt(self, fd_size, B, r, batches, layer_id, distance):
        r = F(r).reshape(-1, 2); c = np.zeros((batches, 2), np.uint64)
        self.lib.ref_generate_claims_opt(c_sz(fd_size), c_sz(B), _p(r), ctypes.c_int(r.shape[0]), ctypes.c_int(batches), ctypes.c_int(layer_id), ctypes.c_int(distance), _p(c))
        return c

    def commit_layers(self, fd_size, B, batches, layer_id, distance):
        n = max(batches - 1, 0)
        roots = np.zeros((n, 32), np.uint8); sizes = np.zeros(n, np.uint64); layers = np.zeros(n, np.uint64)
        self.lib.ref_commit_layers(c_sz(fd_size), c_sz(B), ctypes.c_int(batches), ctypes.c_int(layer_id), ctypes.c_int(distance), _p(roots), _p(sizes), _p(layers))
        return sizes, layers, roots

    def sumcheck3_stream_batch(self, fd_size, B, r, batches, distance, layer_id, old_claims, full=False):
        r = F(r); rlen = r.shape[1]; oc = F(old_claims).reshape(-1, 2)
        size = fd_size >> layer_id; logB = B.bit_length() - 1; nR = size // (2 * B); lR = nR.bit_length() - 1
        ld = 1 + logB + lR
        nc = np.zeros((batches, 2), np.uint64); nr = np.zeros((batches, ld, 2), np.uint64)
        self.lib.ref_sumcheck3_stream_batch(c_sz(fd_size), c_sz(B), _p(r), ctypes.c_int(rlen), ctypes.c_int(batches), ctypes.c_int(distance), ctypes.c_int(layer_id),
                                            _p(oc), ctypes.c_int(oc.shape[0]), _p(nc), _p(nr), ctypes.c_int(ld))
        return dict(new_claims=nc, new_r=[nr[i, :1 + logB - i * distance + lR].copy() for i in range(batches)])

    def mul_tree_stream_shallow(self, fd_size, B, vectors, size, previous_r, distance, prev_x):
        pr = F(previous_r).reshape(2); px = F(prev_x).reshape(-1, 2)
        o = np.zeros((vectors, 2), np.uint64)
        f = self.lib.ref_mul_tree_stream_shallow; f.restype = ctypes.c_int
        n = f(c_sz(fd_size), c_sz(B), ctypes.c_int(vectors), c_sz(size), _p(pr), ctypes.c_int(distance), _p(px), ctypes.c_int(px.shape[0]), _p(o))
        return o[:n]

    def prove_linear_code(self, codeword, n, seed):
        """returns (r1 the reference drew from the libc generator seeded with `seed`, proof)"""
        cw = F(codeword).reshape(-1, 2)
        rounds = cw.shape[0].bit_length() - 1
        q, r, vr, fin = self._proof2(rounds)
        r1 = np.zeros((rounds, 2), np.uint64)
        self.lib.ref_prove_linear_code(_p(cw), c_sz(cw.shape[0]), ctypes.c_longlong(n), ctypes.c_uint(seed), _p(r1), _p(q), _p(r), _p(vr), _p(fin))
        return r1, dict(poly=q, r=r, vr=vr, fin=fin)

    def prove_fft(self, m, rr, prev_sum):
        m = F(m).reshape(-1, 2); rr = F(rr).reshape(-1, 2); ps = F(prev_sum).reshape(2)
        rounds = (2 * m.shape[0]).bit_length() - 1
        q, r, vr, fin = self._proof2(rounds)
        self.lib.ref_prove_fft(_p(m), c_sz(m.shape[0]), _p(rr), _p(ps), _p(q), _p(r), _p(vr), _p(fin))
        return dict(poly=q, r=r[:rounds - 1], vr=vr, fin=fin)

    def prove_fft_matrix(self, M, rr, prev_sum):
        m = F(M); rows, cols = m.shape[0], m.shape[1]; rr = F(rr).reshape(-1, 2); ps = F(prev_sum).reshape(2)
        rounds = (2 * cols).bit_length() - 1
        q, r, vr, fin = self._proof2(rounds)
        self.lib.ref_prove_fft_matrix(_p(m), c_sz(rows), c_sz(cols), _p(rr), _p(ps), _p(q), _p(r), _p(vr), _p(fin))
        return dict(poly=q, r=r, vr=vr, fin=fin)

    def gate_standard(self, L, R, O, add, r):
        """prove_gate_consistency_standard (src/sumcheck.cpp:434-501): returns the four folded values (L, R, O, add)[0]"""
        L, R, O, add, r = [F(v).reshape(-1, 2) for v in (L, R, O, add, r)]
        out = np.zeros((4, 2), np.uint64)
        self.lib.ref_gate_standard(_p(L), _p(R), _p(O), _p(add), c_sz(L.shape[0]), _p(r), ctypes.c_int(r.shape[0]), _p(out))
        return out

    def read_stream_pc(self, N, B, chunk_idx=0):
        o = np.zeros((B, 2), np.uint64)
        self.lib.ref_read_stream_pc(c_sz(N), c_sz(B), c_sz(chunk_idx), _p(o))
        return o

    def time_commit_standard(self, N, K):
        f = self.lib.ref_time_commit_standard; f.restype = ctypes.c_double
        return f(c_sz(N), ctypes.c_int(K))


# ---- deterministic full-range test data (splitmix64, reduced mod p per limb) ------------------
def splitmix_field(n, seed):
    """n full-range F_{p^2} elements, SURVEY 8(d) C2 input generator (splitmix64 mod p)."""
    idx = np.arange(1, 2 * n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) * np.uint64(0x632BE59BD9B4E019) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z % np.uint64(P)).reshape(n, 2)
