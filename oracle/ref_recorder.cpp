// oracle/ref_recorder.cpp -- TEST INFRASTRUCTURE ONLY (built by `make -C oracle ref` into oracle/_ref/).
//
// A transcript recorder that sits IN FRONT of the real reference (oracle/_ref/libhobbit_ref.so) in one process:
//
//   libref_recorder.so  : * `mimc_hash` -- a call-through interposer: the reference's own mimc_hash (src/mimc.cpp:95-107) computes every value;
//                           this one only appends (input, k, result) to a buffer when it is called on the recording thread.  Every Fiat-Shamir
//                           challenge of the streaming provers is a mimc_hash, so the buffer is the complete transcript of whatever ran.
//                         * `open` (src/Elastic_PC.cpp:625) -- Elastic_PC::open ends in SHA3 (verify_claim_opt_blake -> my_hhash, from the prebuilt
//                           lib/libXKCP.a that is never linked here), so the reference's prove_circuit() cannot run past it.  It is the LAST prover
//                           call of prove_circuit (src/main.cpp:880-882, 914-915): this stand-in stops the recording and returns, so that the
//                           reference's own commit, prove_multiplication_tree_stream_shallow and prove_gate_consistency[_lookups] run to their end
//                           on the reference's own Seval streams and main() returns normally.  Nothing it skips feeds anything recorded.
//   libref_openstub.so  : (-DOPENSTUB_ONLY) the same `open` stand-in alone, for the device-backed run of the same command
//                         (tests/mlp_e2e.py loads it in front of the mirror): it stops the library's own recorder (hobbit_transcript_record(0)) at
//                         the same point, so both transcripts end where Elastic_PC::open would begin.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <pthread.h>
#include <dlfcn.h>
#include "config_pc.hpp"
#include "mimc.h"
#include "witness_stream.h"
#include "Blake3_hash.h"

#ifndef OPENSTUB_ONLY
static std::vector<uint64_t> g_rec;
static bool g_on = false;
// HOBBIT_REC_STREAM=<file>: every record is also appended to that file and flushed at once -- for runs of the reference that cannot return
// (Our_PC's open_standard reaches SHA3 inside its first _whir_prove: the process dies on the unresolved symbol, the file keeps what ran before)
static FILE *g_stream = nullptr;
static pthread_t g_thread;
static uint64_t g_other_threads = 0;

typedef F (*mimc_fn_t)(F, F);
static mimc_fn_t g_next = nullptr;           // the reference's own mimc_hash: rec_set_next(dlsym(<handle of libhobbit_ref.so>, "_Z9mimc_hash..."))

F mimc_hash(F input, F k) {
    if (!g_next) { fprintf(stderr, "[ref_recorder] rec_set_next() was not called\n"); abort(); }
    F out = g_next(input, k);
    if (g_on) {
        if (pthread_equal(pthread_self(), g_thread)) {
            const uint64_t r[6] = {input.real, input.img, k.real, k.img, out.real, out.img}; g_rec.insert(g_rec.end(), r, r + 6);
            if (g_stream) { fwrite(r, 8, 6, g_stream); fflush(g_stream); }
        }
        else g_other_threads++;
    }
    return out;
}
extern "C" {
void rec_set_next(void *fn) { g_next = (mimc_fn_t)fn; }
void rec_start(void) {
    g_rec.clear(); g_thread = pthread_self(); g_other_threads = 0; g_on = true;
    const char *f = getenv("HOBBIT_REC_STREAM");
    if (f && !g_stream) g_stream = fopen(f, "wb");
}
void rec_stop(void) { g_on = false; }
size_t rec_count(void) { return g_rec.size() / 6; }
uint64_t rec_other_threads(void) { return g_other_threads; }
size_t rec_read(uint64_t *out, size_t max_records) { size_t n = g_rec.size() / 6; if (n > max_records) n = max_records; if (n) memcpy(out, g_rec.data(), n * 48); return n; }
}
static void stop_recording() { g_on = false; }
#else
static void stop_recording() {
    typedef void (*fn_t)(int);
    fn_t f = (fn_t)dlsym(RTLD_DEFAULT, "hobbit_transcript_record");
    if (f) f(0);
}
#endif

//   libref_recorder_mimc.so : (-DMIMC_ONLY) the mimc_hash recorder WITHOUT the `open` stand-in: for runs in which the reference's own Elastic_PC::open is
//                             what is being recorded (oracle/gen_open_transcript.py; the run ends where the reference first needs SHA3, by itself)
#ifndef MIMC_ONLY
void open(stream_descriptor fd, vector<F> x, vector<vector<_hash>> &Commitment_MT, double &vt, double &ps) {
    (void)x; (void)Commitment_MT; (void)vt; (void)ps;
    stop_recording();
    printf("[ref_recorder] Elastic_PC::open(%s, %lld) not run: the transcript recording ends here\n", fd.name.c_str(), (long long)fd.size);
}
#endif
