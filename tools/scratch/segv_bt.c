// LD_PRELOAD helper: native backtrace on SIGSEGV / SIGBUS / SIGABRT (no gdb on the GPU box).
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <unistd.h>
static void handler(int sig) {
  void* bt[64];
  int n = backtrace(bt, 64);
  dprintf(2, "\n=== native backtrace on signal %d ===\n", sig);
  backtrace_symbols_fd(bt, n, 2);
  _exit(128 + sig);
}
void tbx_install_bt(void) {
  struct sigaction sa;
  sa.sa_handler = handler;
  sigemptyset(&sa.sa_mask);
  sa.sa_flags = SA_NODEFER | SA_RESETHAND;
  sigaction(SIGSEGV, &sa, NULL);
  sigaction(SIGBUS, &sa, NULL);
  sigaction(SIGABRT, &sa, NULL);
}
__attribute__((constructor)) static void init(void) { tbx_install_bt(); }
