// Drives awsm-renderer_amd/csrc/dirty_log.hpp on the CPU (tests/test_dirty_log_cpu.py): reads commands from stdin, prints results.
//   w <buf> <lo> <hi> <seq>      log a write
//   q <since> <max>              ranges written after `since`: "ok n  buf lo hi ..." or "all"
//   p <oldest>                   prune
//   s                            number of entries kept
#include <cstdio>
#include <cstring>
#include "../../awsm-renderer_amd/csrc/dirty_log.hpp"
int main() {
    awsm::DirtyLog log;
    char op[8];
    while (scanf("%7s", op) == 1) {
        if (!strcmp(op, "w")) { unsigned buf; unsigned long long lo, hi, seq; if (scanf("%u %llu %llu %llu", &buf, &lo, &hi, &seq) != 4) return 2; log.log(buf, (size_t)lo, (size_t)hi, seq); }
        else if (!strcmp(op, "q")) {
            unsigned long long since; unsigned max; if (scanf("%llu %u", &since, &max) != 2) return 2;
            std::vector<awsm::DirtyRange> r;
            if (!log.ranges_since(since, max, r)) { printf("all\n"); continue; }
            printf("ok %zu", r.size());
            for (const awsm::DirtyRange& d : r) printf("  %u %u %u", d.buf, d.lo, d.hi);
            printf("\n");
        }
        else if (!strcmp(op, "p")) { unsigned long long o; if (scanf("%llu", &o) != 1) return 2; log.prune(o); }
        else if (!strcmp(op, "s")) printf("%zu\n", log.size());
        else return 2;
    }
    return 0;
}
