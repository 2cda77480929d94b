// segv_bt.c — LD_PRELOAD helper for crash hunting (tools/soak.py): prints the NATIVE backtrace of a SIGSEGV / SIGABRT / SIGBUS to stderr, then re-raises.
// build: gcc -shared -fPIC -O1 -g tools/dbg/segv_bt.c -o /tmp/segv_bt.so ; run: LD_PRELOAD=/tmp/segv_bt.so python tools/soak.py ...
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>
static void on_sig(int sig, siginfo_t* si, void* uc) {
    (void)uc;
    void* frames[64];
    const char msg[] = "\n==== native backtrace (segv_bt) ====\n";
    write(2, msg, sizeof(msg) - 1);
    char line[64]; int n = 0; unsigned long a = (unsigned long)si->si_addr;
    line[n++] = 's'; line[n++] = 'i'; line[n++] = 'g'; line[n++] = ' '; line[n++] = '0' + (sig / 10); line[n++] = '0' + (sig % 10); line[n++] = ' '; line[n++] = '@';
    for (int s = 60; s >= 0; s -= 4) line[n++] = "0123456789abcdef"[(a >> s) & 15];
    line[n++] = '\n'; write(2, line, n);
    const int k = backtrace(frames, 64);
    backtrace_symbols_fd(frames, k, 2);
    signal(sig, SIG_DFL); raise(sig);
}
__attribute__((constructor)) static void install(void) {
    struct sigaction sa; memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = on_sig; sa.sa_flags = SA_SIGINFO | SA_ONSTACK | SA_RESETHAND;
    static char stack[1 << 16]; stack_t ss = {.ss_sp = stack, .ss_size = sizeof(stack), .ss_flags = 0}; sigaltstack(&ss, 0);
    sigaction(SIGSEGV, &sa, 0); sigaction(SIGBUS, &sa, 0); sigaction(SIGABRT, &sa, 0);
}
