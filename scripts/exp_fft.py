"""GPU experiment: FFT-4096 kernel with contiguous vs transposed output."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
mod = load_package(); hb = mod.Hobbit(0)
rows = 1 << 17
d = hb.fill_splitmix(rows * 4096, 3)
hb.profile(True)
for it in range(3):
    hb._chk(hb.lib.hobbit_fft_batch(hb.ctx, d.ptr, 12, rows, 4096, 0))
hb.sync()
print("contiguous in-place full FFT:", {k: (v[0] / v[1]) for k, v in hb.profile_report().items()})
hb.profile_reset()
# tensorcode of one huge 'chunk': M = rows*2048 -> trs = rows: not allowed (codeword). use commit path instead:
N, K = 1 << 28, 32
trs = N // (K << 11)
hb.rng_reset(); hb.expander_init_store(trs)
p = hb.fill_splitmix(N, 5)
for it in range(2):
    c = hb.commit_standard((p, N), K, trs, 1); c.free()
print("commit:", {k: (v[0] / v[1]) for k, v in hb.profile_report().items()})
hb.close()
