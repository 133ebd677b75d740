// Sanitizer driver for the native PDB reader (thermompnn_amd/csrc/tmpnn_pdb.cpp), built by
//   python -m thermompnn_amd.build --pdb-sanitizer-driver      (g++ -fsanitize=address,undefined -fno-sanitize-recover)
// and fed malformed files by tests/test_host.py::test_native_parser_survives_malformed_input.
// usage: pdb_fuzz_driver [--threads N] FILE...   -> one line per file: "<rc> <length> <chains> <checksum>"; any sanitizer
// report aborts with a non-zero exit status. Parses every file alone, then all of them again through the threaded batch entry,
// packs the batch into one ragged staging buffer (tmpnn_pdb_pack_batch) and runs the columnar CSV writer (tmpnn_csv.cpp: both
// schemas, every post-processing flag, the listed-mutation form) over a synthetic table of the parsed structures.
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
    {   // the per-file status form: every file gets a code, a failing one leaves a NULL handle and does not void the others
        std::vector<tmpnn_pdb_t *> hs2(paths.size() + 1, nullptr);
        std::vector<int32_t> st(paths.size() + 1, 12345);
        if (tmpnn_pdb_parse_batch_status(paths.data(), nullptr, (int)paths.size(), threads, hs2.data(), st.data()) != TMPNN_OK) return 6;
        for (size_t i = 0; i < paths.size(); ++i) {
            if ((st[i] == TMPNN_OK) != (hs2[i] != nullptr)) return 6;
            if (hs2[i]) { use(hs2[i]); tmpnn_pdb_free(hs2[i]); }
        }
    }
    std::vector<tmpnn_pdb_t *> hs(paths.size() + 1, nullptr);
    const int rc = tmpnn_pdb_parse_batch(paths.data(), nullptr, (int)paths.size(), threads, hs.data());
    // a failing batch releases every handle; parse the survivors one by one for the pack / writer legs
    std::vector<tmpnn_pdb_t *> ok;
    std::vector<const char *> names;
    if (rc != TMPNN_OK)
        for (const char *p : paths) { tmpnn_pdb_t *h = nullptr; if (tmpnn_pdb_parse(p, nullptr, &h) == TMPNN_OK) { ok.push_back(h); names.push_back(p); } }
    else
        for (size_t i = 0; i < paths.size(); ++i) if (hs[i]) { use(hs[i]); ok.push_back(hs[i]); names.push_back(paths[i]); }
    int64_t T = 0;
    for (tmpnn_pdb_t *h : ok) T += tmpnn_pdb_length(h);
    if (!ok.empty() && T > 0 && T < (1 << 24)) {
        const int n = (int)ok.size();
        std::vector<float> X((size_t)T * 12), mask(T), ca(T), table((size_t)T * 21);
        std::vector<int32_t> S(T), ridx(T), cenc(T), off(n + 1), nb(T);
        if (tmpnn_pdb_pack_batch(ok.data(), n, threads, T - 1, X.data(), S.data(), mask.data(), ridx.data(), cenc.data(), ca.data(), off.data()) != TMPNN_E_WORKSPACE) return 3;
        if (tmpnn_pdb_pack_batch(ok.data(), n, threads, T, X.data(), S.data(), mask.data(), ridx.data(), cenc.data(), ca.data(), off.data()) != TMPNN_OK) return 3;
        for (int64_t i = 0; i < T * 21; ++i) table[i] = (float)((i * 2654435761u % 20011) - 10000) * 1.37e-3f;
        for (int64_t i = 0; i < T; ++i) nb[i] = (int32_t)(i % 37) - 1;
        std::vector<const char *> seqs;
        for (tmpnn_pdb_t *h : ok) seqs.push_back(tmpnn_pdb_seq(h));
        std::vector<int64_t> tri;
        for (int i = 0; i < n; ++i)
            for (int32_t pos = 0; pos < off[i + 1] - off[i]; pos += 7) { tri.push_back(i); tri.push_back(pos); tri.push_back((pos * 3) % 20); }
        for (int schema = 0; schema < 2; ++schema)
            for (int flags = 0; flags < 4; ++flags) {
                tmpnn_csv_t *c = nullptr;
                if (tmpnn_csv_open("/dev/null", schema, &c) != TMPNN_OK) return 4;
                if (tmpnn_csv_write_ssm(c, table.data(), 21, off.data(), n, seqs.data(), flags & 2 ? names.data() : nullptr, names.data(), flags & 1 ? nb.data() : nullptr, "Thermo,MPNN",
                                        "a \"quoted\" set", nullptr, "A", flags, threads) != TMPNN_OK) return 4;
                if (schema == 0 && tmpnn_csv_write_listed(c, table.data(), 21, off.data(), n, seqs.data(), names.data(), nb.data(), "m", "d",
                                                          tri.data(), (int64_t)tri.size() / 3) != TMPNN_OK) return 4;
                int64_t rows = 0, bytes = 0;
                if (tmpnn_csv_close(c, &rows, &bytes) != TMPNN_OK || rows < 0 || bytes <= 0) return 4;
                // the same listing into the MEMORY sink of a sharded scan: explicit running indices, bytes per protein, and the sum of those
                // bytes must be the buffer's fill; then a capacity that is too small must fail cleanly (no write beyond the mapping)
                if (schema == 0) {
                    std::vector<int64_t> first(n), nbytes(n, -1);
                    for (int i = 0; i < n; ++i) first[i] = 1000000007LL * (i + 1);
                    tmpnn_csv_t *mc = nullptr;
                    if (tmpnn_csv_open_mem(schema, flags, bytes + 4096, &mc) != TMPNN_OK) return 5;
                    if (tmpnn_csv_write_ssm_ex(mc, table.data(), 21, off.data(), n, seqs.data(), nullptr, names.data(), flags & 1 ? nb.data() : nullptr, "Thermo,MPNN",
                                               "a \"quoted\" set", nullptr, "A", flags, threads, first.data(), nbytes.data()) != TMPNN_OK) return 5;
                    int64_t fill = 0, sum = 0;
                    const char *mem = tmpnn_csv_mem(mc, &fill);
                    for (int i = 0; i < n; ++i) sum += nbytes[i];
                    if (!mem || fill != sum || (fill > 0 && mem[fill - 1] != '\n')) return 5;
                    if (tmpnn_csv_close(mc, nullptr, nullptr) != TMPNN_OK) return 5;
                    if (fill > 64) {
                        if (tmpnn_csv_open_mem(schema, flags, fill / 2, &mc) != TMPNN_OK) return 5;
                        if (tmpnn_csv_write_ssm_ex(mc, table.data(), 21, off.data(), n, seqs.data(), nullptr, names.data(), nullptr, "m", "d", nullptr, "A", flags, threads,
                                                   nullptr, nullptr) == TMPNN_OK) return 5;
                        if (tmpnn_csv_close(mc, nullptr, nullptr) != TMPNN_OK) return 5;
                    }
                }
            }
    }
    for (tmpnn_pdb_t *h : ok) tmpnn_pdb_free(h);
    printf("batch %d\n", rc);
    return 0;
}
