/* test infrastructure: print a backtrace on SIGSEGV / SIGABRT (loaded by tests/mlp_e2e.py with HOBBIT_E2E_BACKTRACE=1) */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
static void handler(int sig) {
    void *bt[64];
    int n = backtrace(bt, 64);
    fprintf(stderr, "signal %d, backtrace:\n", sig);
    backtrace_symbols_fd(bt, n, 2);
    _exit(128 + sig);
}
void segv_bt_install(void) { signal(SIGSEGV, handler); signal(SIGABRT, handler); signal(SIGBUS, handler); }

/* and on every C++ throw (the library is loaded RTLD_GLOBAL ahead of libstdc++, so the PLT calls land here first) */
#include <dlfcn.h>
void __cxa_throw(void *obj, void *tinfo, void (*dest)(void *)) {
    void *bt[64];
    int n = backtrace(bt, 64);
    fprintf(stderr, "__cxa_throw, backtrace:\n");
    backtrace_symbols_fd(bt, n, 2);
    void (*real)(void *, void *, void (*)(void *)) = (void (*)(void *, void *, void (*)(void *)))dlsym(RTLD_NEXT, "__cxa_throw");
    real(obj, tinfo, dest);
    abort();
}
