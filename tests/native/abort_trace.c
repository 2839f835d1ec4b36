/* abort_trace.c - test infrastructure (tests/conftest.py loads it; nothing in the product does).
 *
 * A process that dies from abort() inside native code (glibc's heap checks, libstdc++'s terminate, the ROCm runtime's
 * own abort on a GPU memory fault or a failed queue) leaves Python's faulthandler dump of the PYTHON threads only -
 * which says where the main thread was, not who called abort(). This handler runs ON THE ABORTING THREAD, writes that
 * thread's native backtrace (module + offset per frame: async-signal-safe backtrace_symbols_fd) and its thread id to a
 * duplicate of the process's original stderr, then hands the signal on to whatever handler was there before
 * (faulthandler), so the log of the next unexplained abort names the library and the thread it came from.
 */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>

static int g_fd = 2;
static struct sigaction g_prev[3];
static const int g_sigs[3] = {SIGABRT, SIGSEGV, SIGBUS};

static void put(const char* s) { ssize_t r = write(g_fd, s, strlen(s)); (void)r; }

static void put_num(long v) {
  char buf[24]; int i = 23; buf[i] = 0;
  if (v == 0) buf[--i] = '0';
  while (v > 0 && i > 0) { buf[--i] = (char)('0' + v % 10); v /= 10; }
  put(buf + i);
}

static void on_fatal(int sig, siginfo_t* info, void* ctx) {
  (void)info; (void)ctx;
  put("\n[abort_trace] fatal signal "); put_num(sig);
  put(" on thread "); put_num((long)syscall(SYS_gettid));
  put(" (process "); put_num((long)getpid()); put("); native frames of that thread:\n");
  void* frames[64];
  const int n = backtrace(frames, 64);
  backtrace_symbols_fd(frames, n, g_fd);
  put("[abort_trace] end of native frames\n");
  /* hand on: restore the previous action and raise again (faulthandler, or the default action = core) */
  for (int k = 0; k < 3; ++k)
    if (g_sigs[k] == sig) sigaction(sig, &g_prev[k], NULL);
  raise(sig);
}

/* fd: a descriptor that will still be the real log when the process dies (a dup of the original fd 2). */
int abort_trace_install(int fd) {
  g_fd = fd;
  void* warm[4];
  (void)backtrace(warm, 4);      /* loads libgcc now: backtrace() must not call the dynamic loader inside the handler */
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = on_fatal;
  sa.sa_flags = SA_SIGINFO | SA_NODEFER;
  sigemptyset(&sa.sa_mask);
  for (int k = 0; k < 3; ++k)
    if (sigaction(g_sigs[k], &sa, &g_prev[k]) != 0) return -1;
  return 0;
}
