/* Test infrastructure (never part of the product): a NATIVE backtrace when a GPU test process is killed by SIGABRT / SIGSEGV /
 * SIGBUS / SIGILL / SIGFPE.  Python's faulthandler names the Python frame (`_native.py: gz_encode_batch`); it cannot say WHO
 * raised inside the native call -- libhsa-runtime64 (a GPU memory fault ends in abort()), libamdhip64, glibc's heap checks, or
 * libstdc++'s std::terminate.  This object prints the native frames (backtrace_symbols_fd: async-signal-safe, no malloc) to fd 2,
 * then hands the signal to whoever had it before (faulthandler, or the default action).
 *
 * Built by tests/conftest.py with gcc; loaded into the main pytest process with ctypes and into the tests' child processes with
 * LD_PRELOAD (the constructor installs the handlers either way). */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static struct sigaction old_act[65];
static const int sigs[] = {SIGABRT, SIGSEGV, SIGBUS, SIGILL, SIGFPE};

static void put(const char* s) { ssize_t r = write(2, s, strlen(s)); (void)r; }

static void on_fatal(int sig, siginfo_t* si, void* uc)
{
    (void)si; (void)uc;
    void* frames[96];
    put("\n[sigtrace] fatal signal ");
    put(sig == SIGABRT ? "SIGABRT" : sig == SIGSEGV ? "SIGSEGV" : sig == SIGBUS ? "SIGBUS" : sig == SIGILL ? "SIGILL" : "SIGFPE");
    put(": native frames of the thread that received it (innermost first)\n");
    const int n = backtrace(frames, 96);
    backtrace_symbols_fd(frames, n, 2);
    put("[sigtrace] end of native frames\n");
    /* back to the previous disposition and once more: faulthandler (if it was there first) prints the Python frames, the default
     * action ends the process with the signal's status */
    sigaction(sig, &old_act[sig], NULL);
    raise(sig);
}

__attribute__((constructor)) static void sigtrace_install(void)
{
    void* warm[4];
    (void)backtrace(warm, 4);                 /* loads libgcc's unwinder NOW: the first call allocates, a signal handler must not */
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_fatal;
    sa.sa_flags = SA_SIGINFO | SA_NODEFER | SA_ONSTACK;
    sigemptyset(&sa.sa_mask);
    for (unsigned i = 0; i < sizeof sigs / sizeof sigs[0]; ++i) sigaction(sigs[i], &sa, &old_act[sigs[i]]);
}

int sigtrace_loaded(void) { return 1; }
