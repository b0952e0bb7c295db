// In-process native stack dumper for hang diagnosis (no debugger attaches to a GPU process on this pool).
//   nbt_install()   installs a SIGUSR2 handler that writes the calling thread's native frames to stderr
//   nbt_dump_all()  signals every thread of this process in turn (tgkill) and waits for each dump
// Build: gcc -O1 -g -shared -fPIC -o libnative_bt.so native_bt.c
#define _GNU_SOURCE
#include <dirent.h>
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>

static volatile int nbt_done;

static void nbt_handler(int sig) {
    (void)sig;
    void* frames[64];
    char head[96];
    int n = backtrace(frames, 64);
    int len = snprintf(head, sizeof head, "=== native stack of thread %ld ===\n", (long)syscall(SYS_gettid));
    if (write(2, head, len) < 0) {}
    backtrace_symbols_fd(frames, n, 2);
    nbt_done = 1;
}

int nbt_install(void) {
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = nbt_handler;
    sa.sa_flags = SA_RESTART;
    void* warm[4];
    backtrace(warm, 4);                       // loads libgcc now, not inside the handler
    return sigaction(SIGUSR2, &sa, NULL);
}

int nbt_dump_all(void) {
    DIR* d = opendir("/proc/self/task");
    if (!d) return -1;
    struct dirent* e;
    long self = syscall(SYS_gettid);
    pid_t pid = getpid();
    int count = 0;
    while ((e = readdir(d)) != NULL) {
        long tid = atol(e->d_name);
        if (tid <= 0 || tid == self) continue;
        nbt_done = 0;
        if (syscall(SYS_tgkill, pid, tid, SIGUSR2) != 0) continue;
        for (int spin = 0; spin < 2000 && !nbt_done; ++spin) usleep(500);
        ++count;
    }
    closedir(d);
    return count;
}
