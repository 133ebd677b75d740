// Columnar result writer (host C++). The reference appends to a pandas frame cell by cell with one
// device sync per value and lets DataFrame.to_csv format it (/root/reference/analysis/SSM.py:128-176,
// analysis/custom_inference.py:94-111); at engine speed (10^8 predictions/s) the writer is the bottleneck, so this formats the
// gathered [T, 21] ddG table straight into the reference's two CSV layouts:
//   schema 0 (SSM.py:102-103,176):              ,WT Seq,Model,Dataset,ddG_pred,position,wildtype,mutation,neighbors,best_AA,pdb
//   schema 1 (custom_inference.py:64,96-111):   ,Model,Dataset,ddG_pred,position,wildtype,mutation,pdb,chain
// byte for byte what pandas writes: '\n' line ends, the unnamed running index first, floats as repr(float(x)) (shortest
// round-trip digits of the DOUBLE the fp32 value converts to, Python's fixed / exponent switch), empty cells for missing
// values, minimal quoting. Proteins are formatted by a pool of host threads, each into its own buffer; buffers are
// committed in protein order (a ticket hands out the file offset) and written with pwrite outside the lock, so formatting
// and the page-cache copies of different proteins overlap and memory stays at one protein per thread.
#include <errno.h>
#include <fcntl.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <charconv>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/tmpnn.h"
#include "tmpnn_host_guard.hpp"

namespace {

const char kAA20[21] = "ACDEFGHIKLMNPQRSTVWY";        // ALPHABET[:-1], datasets.py:13

// repr(float) of CPython (float_repr_style 'short', format code 'r'): shortest digits that round-trip, fixed notation when
// -4 <= exponent < 16, else d[.ddd]e+XX. -> number of chars written (at most 25).
inline int repr_double(double v, char *out) {
    if (v != v) { memcpy(out, "nan", 3); return 3; }
    char *o = out;
    if (std::signbit(v)) { *o++ = '-'; v = -v; }
    if (v == INFINITY) { memcpy(o, "inf", 3); return (int)(o - out) + 3; }
    if (v == 0.0) { memcpy(o, "0.0", 3); return (int)(o - out) + 3; }
    char sci[40];
    const std::to_chars_result r = std::to_chars(sci, sci + sizeof(sci), v, std::chars_format::scientific);
    // sci = d[.ddd]e[+-]XX
    char digits[24];
    int nd = 0;
    const char *q = sci;
    for (; q < r.ptr && *q != 'e'; ++q)
        if (*q != '.') digits[nd++] = *q;
    ++q;                                                         // 'e'
    const bool eneg = *q == '-';
    ++q;
    int ex = 0;
    for (; q < r.ptr; ++q) ex = ex * 10 + (*q - '0');
    if (eneg) ex = -ex;
    const int decpt = ex + 1;                                    // digits[0..decpt) are the integer part
    if (decpt > -4 && decpt <= 16) {
        if (decpt <= 0) {
            *o++ = '0'; *o++ = '.';
            for (int i = 0; i < -decpt; ++i) *o++ = '0';
            memcpy(o, digits, nd); o += nd;
        } else if (decpt >= nd) {
            memcpy(o, digits, nd); o += nd;
            for (int i = nd; i < decpt; ++i) *o++ = '0';
            *o++ = '.'; *o++ = '0';
        } else {
            memcpy(o, digits, decpt); o += decpt;
            *o++ = '.';
            memcpy(o, digits + decpt, nd - decpt); o += nd - decpt;
        }
    } else {
        *o++ = digits[0];
        if (nd > 1) { *o++ = '.'; memcpy(o, digits + 1, nd - 1); o += nd - 1; }
        *o++ = 'e';
        *o++ = ex < 0 ? '-' : '+';
        int a = ex < 0 ? -ex : ex;
        char t[8]; int nt = 0;
        do { t[nt++] = (char)('0' + a % 10); a /= 10; } while (a);
        if (nt < 2) t[nt++] = '0';
        while (nt) *o++ = t[--nt];
    }
    return (int)(o - out);
}

inline char *put_int(char *o, int64_t v) {
    if (v < 0) { *o++ = '-'; v = -v; }
    char t[24]; int n = 0;
    do { t[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *o++ = t[--n];
    return o;
}

// csv QUOTE_MINIMAL (what both csv.writer and pandas do): quote a field that holds the delimiter, a quote or a line break
std::string csv_field(const char *s) {
    const std::string f = s ? s : "";
    if (f.find_first_of(",\"\r\n") == std::string::npos) return f;
    std::string q = "\"";
    for (char c : f) { if (c == '"') q.push_back('"'); q.push_back(c); }
    q.push_back('"');
    return q;
}

}  // namespace

struct tmpnn_csv {
    int fd = -1;
    int schema = 0;
    int64_t rows = 0;          // data rows written so far = the next running index
    int64_t bytes = 0;         // file offset of the next byte
    char *mem = nullptr;       // tmpnn_csv_open_mem: the text goes into this anonymous mapping instead of a file (fd = -1)
    int64_t mem_cap = 0;
    bool header = true;        // false: a part file of a sharded scan (no header line; tmpnn_csv_open_ex NO_HEADER)
    bool pick_header = false;  // the header line ends in ",dupe_detector" (PICK_BEST listings, SSM.py:161)
    std::string path;
};

static const char *header_text(int schema, bool pick) {
    if (schema == 1) return ",Model,Dataset,ddG_pred,position,wildtype,mutation,pdb,chain\n";
    return pick ? ",WT Seq,Model,Dataset,ddG_pred,position,wildtype,mutation,neighbors,best_AA,pdb,dupe_detector\n"
                : ",WT Seq,Model,Dataset,ddG_pred,position,wildtype,mutation,neighbors,best_AA,pdb\n";
}

static bool write_all_at(int fd, const char *p, size_t n, int64_t off) {
    while (n) {
        const ssize_t w = pwrite(fd, p, n, (off_t)off);
        if (w < 0) { if (errno == EINTR) continue; return false; }
        p += w; n -= (size_t)w; off += w;
    }
    return true;
}

// file or memory sink
static bool put_at(tmpnn_csv *c, const char *p, size_t n, int64_t off) {
    if (c->mem) {
        if (off < 0 || off + (int64_t)n > c->mem_cap) { errno = ENOSPC; return false; }
        memcpy(c->mem + off, p, n);
        return true;
    }
    return write_all_at(c->fd, p, n, off);
}

// One rank's share of a sharded scan kept in MEMORY until the ranks have exchanged their byte counts (dist.scan_files_to_csv): an
// anonymous mapping of `capacity` bytes (address space only — pages are touched as text arrives; transparent huge pages asked
// for), no header line. A part FILE on tmpfs would pay the file system's page allocation twice (part + final file).
extern "C" int tmpnn_csv_open_mem(int schema, int flags, int64_t capacity, tmpnn_csv_t **out) {
    if (!out || schema < 0 || schema > 1 || capacity <= 0) return tm_set_error(TMPNN_E_INVALID, "csv_open_mem: bad argument");
    void *m = mmap(nullptr, (size_t)capacity, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (m == MAP_FAILED) return tm_set_error(TMPNN_E_WORKSPACE, "csv_open_mem: cannot reserve %lld bytes: %s", (long long)capacity, strerror(errno));
#ifdef MADV_HUGEPAGE
    (void)madvise(m, (size_t)capacity, MADV_HUGEPAGE);
#endif
    const int rc = tm_host_guard("csv_open_mem", [&]() -> int {
        std::unique_ptr<tmpnn_csv> c(new tmpnn_csv());
        c->schema = schema; c->path = "<memory>"; c->header = false; c->pick_header = schema == 0 && (flags & TMPNN_CSV_PICK_BEST);
        c->mem = (char *)m; c->mem_cap = capacity;
        *out = c.release();
        return TMPNN_OK;
    });
    if (rc != TMPNN_OK) munmap(m, (size_t)capacity);
    return rc;
}
extern "C" const char *tmpnn_csv_mem(const tmpnn_csv_t *c, int64_t *bytes_out) {
    if (!c || !c->mem) return nullptr;
    if (bytes_out) *bytes_out = c->bytes;
    return c->mem;
}

extern "C" int tmpnn_csv_open_ex(const char *path, int schema, int flags, tmpnn_csv_t **out) {
    if (!path || !out || schema < 0 || schema > 1) return tm_set_error(TMPNN_E_INVALID, "csv_open: bad argument");
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644);
    if (fd < 0) return tm_set_error(TMPNN_E_INVALID, "csv_open: cannot create %s: %s", path, strerror(errno));
    const bool with_header = !(flags & TMPNN_CSV_NO_HEADER), pick = schema == 0 && (flags & TMPNN_CSV_PICK_BEST);
    const char *hdr = header_text(schema, pick);
    const int rc = tm_host_guard("csv_open", [&]() -> int {
        std::unique_ptr<tmpnn_csv> c(new tmpnn_csv());
        c->fd = fd; c->schema = schema; c->path = path; c->header = with_header; c->pick_header = pick;
        if (with_header) {
            if (!write_all_at(fd, hdr, strlen(hdr), 0))
                return tm_set_error(TMPNN_E_INVALID, "csv_open: write to %s failed: %s", path, strerror(errno));
            c->bytes = (int64_t)strlen(hdr);
        }
        *out = c.release();
        return TMPNN_OK;
    });
    if (rc != TMPNN_OK) close(fd);
    return rc;
}
extern "C" int tmpnn_csv_open(const char *path, int schema, tmpnn_csv_t **out) { return tmpnn_csv_open_ex(path, schema, 0, out); }

extern "C" int tmpnn_csv_header(int schema, int flags, char *buf, int cap) {
    if (schema < 0 || schema > 1 || !buf) return tm_set_error(TMPNN_E_INVALID, "csv_header: bad argument");
    const char *h = header_text(schema, schema == 0 && (flags & TMPNN_CSV_PICK_BEST));
    const int n = (int)strlen(h);
    if (cap <= n) return tm_set_error(TMPNN_E_INVALID, "csv_header: buffer of %d bytes, need %d", cap, n + 1);
    memcpy(buf, h, (size_t)n + 1);
    return n;
}

extern "C" int tmpnn_csv_close(tmpnn_csv_t *c, int64_t *rows_out, int64_t *bytes_out) {
    if (!c) return tm_set_error(TMPNN_E_INVALID, "csv_close: null handle");
    if (rows_out) *rows_out = c->rows;
    if (bytes_out) *bytes_out = c->bytes;
    const int rc = c->mem ? munmap(c->mem, (size_t)c->mem_cap) : close(c->fd);
    const std::string path = c->path;
    delete c;
    if (rc != 0) return tm_set_error(TMPNN_E_INVALID, "csv_close: %s: %s", path.c_str(), strerror(errno));
    return TMPNN_OK;
}

// The full site-saturation listing of n proteins (SSM.py:128-166 / custom_inference.py:94-111).
//   table [T, ld] fp32 HOST (ld >= 20; columns 0..19 = ddG of mutating to ALPHABET[a]); offsets [n+1] int32 rows of `table`;
//   seqs[i] = parsed sequence of protein i ('-' positions are skipped: the reference's None mutations); names[i] = the 'pdb'
//   cell; neighbors (may be NULL) [T] int32 -> 'neighbors' cell (schema 0); datasets (may be NULL: `dataset` for all) per
//   protein 'Dataset' cells; chain = schema 1's last cell.
//   flags: TMPNN_CSV_PICK_BEST = one row per position (mutation 'A', drop_duplicates keep='first') carrying best_AA = argmin
//   ddG (first minimum; C excluded unless TMPNN_CSV_INCLUDE_CYS); without PICK_BEST, rows mutating to C are dropped unless
//   INCLUDE_CYS (SSM.py:153-166). Both schemas honour the flags; the custom_inference layout is written with INCLUDE_CYS
//   (that script lists all 20 mutants).
//   first_rows (may be NULL) [n]: the running index of every protein's first row when it is not the writer's own count — a rank of
//   a sharded scan formats ITS proteins with the indices they have in the one output file (dist.scan_files_to_csv);
//   bytes_out (may be NULL) [n]: bytes of text written for each protein (what the ranks exchange to place their text).
extern "C" int tmpnn_csv_write_ssm_ex(tmpnn_csv_t *c, const float *table, int ld, const int32_t *offsets, int n,
                                      const char *const *seqs, const char *const *wt_cells, const char *const *names, const int32_t *neighbors,
                                      const char *model, const char *dataset, const char *const *datasets, const char *chain,
                                      int flags, int n_threads, const int64_t *first_rows, int64_t *bytes_out) {
    if (!c || !offsets || n < 0 || ld < 20 || (n > 0 && (!table || !seqs || !names)))
        return tm_set_error(TMPNN_E_INVALID, "csv_write_ssm: bad argument");
    return tm_host_guard("csv_write_ssm", [&]() -> int {
    const bool pick = flags & TMPNN_CSV_PICK_BEST, cys = flags & TMPNN_CSV_INCLUDE_CYS;
    const int schema = c->schema;
    const bool dupe = pick && schema == 0;                  // the reference's PICK_BEST frames carry one more column (SSM.py:161-162)
    if (dupe != c->pick_header) {
        // a file opened with tmpnn_csv_open (no flags) takes its header from the first listing written into it
        if (c->rows != 0 || (c->header && c->bytes != (int64_t)strlen(header_text(schema, c->pick_header))))
            return tm_set_error(TMPNN_E_INVALID, "csv_write_ssm: %s: PICK_BEST must be the same for every listing of a file", c->path.c_str());
        if (c->header) {
            const char *hdr = header_text(schema, dupe);
            if (!put_at(c, hdr, strlen(hdr), 0))
                return tm_set_error(TMPNN_E_INVALID, "csv_write_ssm: write to %s failed: %s", c->path.c_str(), strerror(errno));
            c->bytes = (int64_t)strlen(hdr);
        }
        c->pick_header = dupe;
    }
    // running index of every protein's first row, and its number of rows
    std::vector<int64_t> first((size_t)n + 1), nrows_of((size_t)n);
    first[0] = c->rows;
    int64_t total_rows = 0;
    for (int i = 0; i < n; ++i) {
        const int32_t L = offsets[i + 1] - offsets[i];
        if (L < 0 || !seqs[i] || (int64_t)strlen(seqs[i]) != L)
            return tm_set_error(TMPNN_E_INVALID, "csv_write_ssm: protein %d: sequence length does not match its %d table rows", i, L);
        int64_t npos = 0;
        for (int32_t k = 0; k < L; ++k) npos += seqs[i][k] != '-';
        nrows_of[i] = npos * (pick ? 1 : (cys ? 20 : 19));
        if (first_rows) first[i] = first_rows[i];
        first[i + 1] = first[i] + nrows_of[i];
        total_rows += nrows_of[i];
    }
    const std::string f_model = csv_field(model), f_chain = csv_field(chain);
    std::atomic<int> next(0), write_errno(0);
    std::atomic<bool> failed(false);
    std::mutex mu;
    std::condition_variable cv;
    int commit = 0;                       // next protein allowed to take its file offset
    int64_t off = c->bytes;
    auto write_failed = [&]() { int z = 0; write_errno.compare_exchange_strong(z, errno); failed = true; };
    auto work = [&]() {
        std::vector<char> buf;
        for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) {
          bool ticket_taken = false, ticket_returned = false;   // the others wait for protein i's ticket: it is handed on even if this protein throws
          try {
            const char *seq = seqs[i];
            const int32_t L = offsets[i + 1] - offsets[i];
            const float *tab = table + (size_t)offsets[i] * ld;
            const int32_t *nb = neighbors ? neighbors + offsets[i] : nullptr;
            const std::string f_name = csv_field(names[i]);
            // the dupe_detector cell: the name with the position appended, quoted as a whole when the name needs quoting
            const bool f_dupe_quoted = f_name.size() >= 2 && f_name[0] == '"' && f_name != names[i];
            const std::string f_dupe = f_dupe_quoted ? f_name.substr(1, f_name.size() - 2) : f_name;
            const std::string f_data = csv_field(datasets ? datasets[i] : dataset);
            // the cells between the running index and the ddG value, and after the per-row cells
            std::string head = ",";
            if (schema == 0) { head += csv_field(wt_cells && wt_cells[i] ? wt_cells[i] : seq); head += ","; }
            head += f_model; head += ","; head += f_data; head += ",";
            const int64_t nrows = nrows_of[i];
            const size_t per_row = 20 + head.size() + 26 + 12 + 4 + 12 + 2 + 2 * f_name.size() + 12 + f_chain.size() + 4;
            // rows of positions [p0, p1) -> buf (from its start); returns the running index after them
            auto format = [&](int32_t p0, int32_t p1, int64_t row, size_t *len_out) {
                char *o = buf.data();
                for (int32_t pos = p0; pos < p1; ++pos) {
                    const char wt = seq[pos];
                    if (wt == '-') continue;
                    const float *t = tab + (size_t)pos * ld;
                    char best = 0;
                    if (pick) {                                       // idxmin: first minimum wins (SSM.py:32-42)
                        int bi = -1;
                        double bv = 0;
                        for (int a = 0; a < 20; ++a) {
                            if ((!cys && kAA20[a] == 'C') || t[a] != t[a]) continue;       // (idxmin skips missing values)
                            if (bi < 0 || (double)t[a] < bv) { bi = a; bv = t[a]; }
                        }
                        best = bi < 0 ? 0 : kAA20[bi];
                    }
                    for (int a = 0; a < (pick ? 1 : 20); ++a) {
                        if (!pick && !cys && kAA20[a] == 'C') continue;
                        o = put_int(o, row++);
                        memcpy(o, head.data(), head.size()); o += head.size();
                        if (t[a] == t[a]) o += repr_double((double)t[a], o);      // (a missing value is an empty cell in pandas' CSV)
                        *o++ = ',';
                        o = put_int(o, pos);
                        *o++ = ','; *o++ = wt; *o++ = ','; *o++ = kAA20[a]; *o++ = ',';
                        if (schema == 0) {
                            if (nb) o = put_int(o, nb[pos]);
                            *o++ = ',';
                            if (best) *o++ = best;
                            *o++ = ',';
                            memcpy(o, f_name.data(), f_name.size()); o += f_name.size();
                            if (dupe) {                                   // 'dupe_detector' = pdb + str(position) (SSM.py:161)
                                *o++ = ',';
                                if (f_dupe_quoted) *o++ = '"';
                                memcpy(o, f_dupe.data(), f_dupe.size()); o += f_dupe.size();
                                o = put_int(o, pos);
                                if (f_dupe_quoted) *o++ = '"';
                            }
                        } else {
                            memcpy(o, f_name.data(), f_name.size()); o += f_name.size();
                            *o++ = ',';
                            memcpy(o, f_chain.data(), f_chain.size()); o += f_chain.size();
                        }
                        *o++ = '\n';
                    }
                }
                *len_out = (size_t)(o - buf.data());
                return row;
            };
            // The layout repeats the whole sequence in every row, so a protein's text grows with L^2 (L = 512: 5.8 MB; L = 35 000:
            // 24 GB). Up to kWholeProtein bytes a protein is formatted as one buffer, off the critical path, and committed in order;
            // beyond that it waits for its turn and streams position blocks of about kBlock bytes straight to its place in the file.
            constexpr size_t kWholeProtein = (size_t)64 << 20, kBlock = (size_t)8 << 20;
            const size_t rows_per_pos = pick ? 1 : (cys ? 20 : 19);
            if ((size_t)nrows * per_row <= kWholeProtein) {
                buf.resize((size_t)nrows * per_row + 64);
                size_t len = 0;
                format(0, L, first[i], &len);
                int64_t at;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return commit == i; });
                    at = off;
                    off += (int64_t)len;
                    ++commit;
                    ticket_taken = ticket_returned = true;
                }
                cv.notify_all();
                if (bytes_out) bytes_out[i] = (int64_t)len;
                if (len && !put_at(c, buf.data(), len, at)) write_failed();
            } else {
                const int32_t step = (int32_t)std::max<size_t>(1, kBlock / (rows_per_pos * per_row));
                buf.resize((size_t)step * rows_per_pos * per_row + 64);
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return commit == i; });
                    ticket_taken = true;
                }
                int64_t row = first[i], at = off;                  // (this thread owns the tail of the file until it bumps `commit`)
                const int64_t at0 = at;
                for (int32_t p0 = 0; p0 < L; p0 += step) {
                    size_t len = 0;
                    row = format(p0, std::min<int32_t>(L, p0 + step), row, &len);
                    if (len && !put_at(c, buf.data(), len, at)) write_failed();
                    at += (int64_t)len;
                }
                if (bytes_out) bytes_out[i] = at - at0;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    off = at;
                    ++commit;
                    ticket_returned = true;
                }
                cv.notify_all();
            }
          } catch (...) {
            if (!ticket_returned) {
                std::unique_lock<std::mutex> lk(mu);
                if (!ticket_taken) cv.wait(lk, [&] { return commit == i; });
                ++commit;
                lk.unlock();
                cv.notify_all();
            }
            throw;                                              // tm_run_pool keeps the first one for the caller's thread
          }
        }
    };
    tm_run_pool(std::max(1, std::min(n_threads, std::max(n, 1))), work);
    if (failed) return tm_set_error(TMPNN_E_INVALID, "csv_write_ssm: write to %s failed: %s", c->path.c_str(), strerror(write_errno.load()));
    c->rows += total_rows;
    c->bytes = off;
    return TMPNN_OK;
    });
}
extern "C" int tmpnn_csv_write_ssm(tmpnn_csv_t *c, const float *table, int ld, const int32_t *offsets, int n,
                                   const char *const *seqs, const char *const *wt_cells, const char *const *names, const int32_t *neighbors,
                                   const char *model, const char *dataset, const char *const *datasets, const char *chain,
                                   int flags, int n_threads) {
    return tmpnn_csv_write_ssm_ex(c, table, ld, offsets, n, seqs, wt_cells, names, neighbors, model, dataset, datasets, chain, flags, n_threads,
                                  nullptr, nullptr);
}

// An explicit mutation list (BASELINE config 4; ssm_scan --mutations): triples [m, 3] int64 HOST of (protein, 0-based
// position, amino-acid index < 20), rows in list order, schema 0 cells, best_AA empty.
extern "C" int tmpnn_csv_write_listed(tmpnn_csv_t *c, const float *table, int ld, const int32_t *offsets, int n,
                                      const char *const *seqs, const char *const *names, const int32_t *neighbors,
                                      const char *model, const char *dataset, const int64_t *triples, int64_t m) {
    if (!c || c->schema != 0 || !offsets || n < 0 || ld < 20 || m < 0 || (m > 0 && (!table || !seqs || !names || !triples)))
        return tm_set_error(TMPNN_E_INVALID, "csv_write_listed: bad argument");
    return tm_host_guard("csv_write_listed", [&]() -> int {
    const std::string mid = std::string(",") + csv_field(model) + "," + csv_field(dataset) + ",";
    std::vector<char> buf;
    buf.reserve(1 << 20);
    std::vector<std::string> f_seq((size_t)n), f_name((size_t)n);
    std::vector<char> have((size_t)n, 0);
    int64_t at = c->bytes, row = c->rows;
    auto flush = [&]() -> bool {
        if (buf.empty()) return true;
        const bool ok = put_at(c, buf.data(), buf.size(), at);
        at += (int64_t)buf.size();
        buf.clear();
        return ok;
    };
    for (int64_t k = 0; k < m; ++k) {
        const int64_t i = triples[3 * k], pos = triples[3 * k + 1], a = triples[3 * k + 2];
        if (i < 0 || i >= n || a < 0 || a >= 20 || pos < 0 || pos >= offsets[i + 1] - offsets[i])
            return tm_set_error(TMPNN_E_INVALID, "csv_write_listed: triple %lld out of range", (long long)k);
        if (!have[i]) { f_seq[i] = csv_field(seqs[i]); f_name[i] = csv_field(names[i]); have[i] = 1; }
        char cell[96];
        char *o = put_int(cell, row++);
        buf.insert(buf.end(), cell, o);
        buf.push_back(',');
        buf.insert(buf.end(), f_seq[i].begin(), f_seq[i].end());
        buf.insert(buf.end(), mid.begin(), mid.end());
        const size_t r = (size_t)(offsets[i] + pos);
        o = cell;
        o += repr_double((double)table[r * ld + a], o);
        *o++ = ',';
        o = put_int(o, pos);
        *o++ = ','; *o++ = seqs[i][pos]; *o++ = ','; *o++ = kAA20[a]; *o++ = ',';
        if (neighbors) o = put_int(o, neighbors[r]);
        *o++ = ','; *o++ = ',';
        buf.insert(buf.end(), cell, o);
        buf.insert(buf.end(), f_name[i].begin(), f_name[i].end());
        buf.push_back('\n');
        if (buf.size() > (1 << 20) - 4096 && !flush())
            return tm_set_error(TMPNN_E_INVALID, "csv_write_listed: write to %s failed: %s", c->path.c_str(), strerror(errno));
    }
    if (!flush()) return tm_set_error(TMPNN_E_INVALID, "csv_write_listed: write to %s failed: %s", c->path.c_str(), strerror(errno));
    c->rows = row;
    c->bytes = at;
    return TMPNN_OK;
    });
}

// repr(float(x)) as a C string (what the writer puts in the ddG_pred cell); exported so the host tests can pin the number
// format against Python's own repr on arbitrary values. buf must hold 32 bytes. -> length.
extern "C" int tmpnn_csv_format_double(double v, char *buf) {
    if (!buf) return tm_set_error(TMPNN_E_INVALID, "csv_format_double: null buffer");
    const int n = repr_double(v, buf);
    buf[n] = '\0';
    return n;
}
