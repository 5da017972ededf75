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
