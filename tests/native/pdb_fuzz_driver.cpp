// Sanitizer driver for the native PDB reader (thermompnn_amd/csrc/tmpnn_pdb.cpp), built by
//   python -m thermompnn_amd.build --pdb-sanitizer-driver      (g++ -fsanitize=address,undefined -fno-sanitize-recover)
// and fed malformed files by tests/test_host.py::test_native_parser_survives_malformed_input.
// usage: pdb_fuzz_driver [--threads N] FILE...   -> one line per file: "<rc> <length> <chains> <checksum>"; any sanitizer
// report aborts with a non-zero exit status. Parses every file alone, then all of them again through the threaded batch entry.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/tmpnn.h"

static thread_local char g_err[512];
int tm_set_error(int code, const char *fmt, ...) {       // the library's error sink (tmpnn_api.hip), restated for the driver
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

static double use(tmpnn_pdb_t *h) {                      // touch every output byte the Python front end would read
    const int64_t L = tmpnn_pdb_length(h);
    std::vector<float> X((size_t)L * 12 + 1), mask(L + 1), ca(L + 1);
    std::vector<int32_t> S(L + 1), ridx(L + 1), cenc(L + 1);
    std::vector<char> seq(L + 1);
    if (tmpnn_pdb_fill(h, X.data(), S.data(), mask.data(), ridx.data(), cenc.data(), seq.data(), ca.data()) != TMPNN_OK) return -1;
    double sum = 0;
    for (int64_t i = 0; i < L; ++i) sum += S[i] + mask[i] + ca[i] + ridx[i] * 1e-3 + cenc[i] + X[12 * i + 3] * 1e-6 + seq[i];
    return sum;
}

int main(int argc, char **argv) {
    int threads = 4, first = 1;
    if (argc > 2 && strcmp(argv[1], "--threads") == 0) { threads = atoi(argv[2]); first = 3; }
    std::vector<const char *> paths(argv + first, argv + argc);
    for (const char *p : paths) {
        tmpnn_pdb_t *h = nullptr;
        const int rc = tmpnn_pdb_parse(p, nullptr, &h);
        if (rc == TMPNN_OK) {
            printf("%d %lld %d %.6f\n", rc, (long long)tmpnn_pdb_length(h), tmpnn_pdb_num_chains(h), use(h));
            tmpnn_pdb_free(h);
            tmpnn_pdb_t *h2 = nullptr;                  // and with an explicit chain selection
            if (tmpnn_pdb_parse(p, "BA", &h2) == TMPNN_OK) { use(h2); tmpnn_pdb_free(h2); }
        } else {
            printf("%d -1 -1 0\n", rc);
        }
    }
    std::vector<tmpnn_pdb_t *> hs(paths.size() + 1, nullptr);
    const int rc = tmpnn_pdb_parse_batch(paths.data(), nullptr, (int)paths.size(), threads, hs.data());
    for (size_t i = 0; i < paths.size(); ++i)
        if (hs[i]) { use(hs[i]); tmpnn_pdb_free(hs[i]); }
    printf("batch %d\n", rc);
    return 0;
}
