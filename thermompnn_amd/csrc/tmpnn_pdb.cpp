// Native PDB reader + packer (host C++). At GPU speeds the Python parser
// (/root/reference/protein_mpnn_utils.py:183-350, which re-reads the file once per chain letter) and the numpy
// packing of tied_featurize (:353-605) dominate a many-PDB scan; this does both in one pass per file and
// parses a batch of files on several threads. Semantics restated from the reference parser:
//   - bytes decoded leniently, trailing whitespace stripped; HETATM+MSE lines become ATOM+MET (:217-220)
//   - fixed columns: chain 21, atom 12-15, residue name 17-19, residue number + insertion code 22-26, xyz 30-53
//   - residue number - 1 with an insertion-code sub-key; first occurrence of a residue name / atom wins (:241-250)
//   - every number between min and max appears: missing numbers -> '-' with NaN coords; insertion codes sorted
//   - unknown residue names -> '-'  (:262-266);  packing: '-' -> 'X' (20), mask = all 12 backbone coords finite,
//     NaN -> 0, residue_idx = 100 (c-1) + position, chain_encoding = c (1-based), chains in the requested order
#include <ctype.h>
#include <errno.h>
#include <fcntl.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <memory>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

// Host-only translation unit: it needs the C-ABI header and the library's error sink, nothing of HIP — so the same file
// also builds with plain g++ under -fsanitize=address,undefined into the fuzz driver of tests/native/pdb_fuzz_driver.cpp
// (python -m thermompnn_amd.build --pdb-sanitizer-driver).
#include "../../include/tmpnn.h"
#include "tmpnn_host_guard.hpp"
int tm_set_error(int code, const char *fmt, ...);

namespace {

// Residue numbers come from a 5-column field (-9999 .. 99999), so a chain can span at most ~110 000 numbers; the bound is
// enforced anyway: every number between a chain's lowest and highest becomes a row (gaps included), and a hostile file must
// get an error, not an allocation proportional to a number it made up.
constexpr long kMaxChainSpan = 200000;

struct Residue {
    long num = 0;               // residue number - 1
    char ins = 0;               // insertion code, 0 = none (sorts first, like the reference's "" sub-key)
    char name[3] = {0, 0, 0};   // first residue name seen
    unsigned have = 0;          // bit a = backbone atom a seen (first occurrence wins)
    double xyz[4][3];
};
struct ChainAcc {
    std::vector<Residue> res;                       // in order of first appearance; sorted by (num, ins) at the end if needed
    std::unordered_map<int64_t, int> index;         // (num, ins) -> slot; consulted only when the key changes between records
    long lo = 0, hi = 0;
    int last = -1;
    bool any = false, sorted = true;
};

const char kAA3[20][4] = {"ALA", "ARG", "ASN", "ASP", "CYS", "GLN", "GLU", "GLY", "HIS", "ILE",
                          "LEU", "LYS", "MET", "PHE", "PRO", "SER", "THR", "TRP", "TYR", "VAL"};
const char kAA1[21] = "ARNDCQEGHILKMFPSTWYV";
const char kMpnn[22] = "ACDEFGHIKLMNPQRSTVWYX";

char one_letter(const char name[3]) {
    for (int i = 0; i < 20; ++i)
        if (name[0] == kAA3[i][0] && name[1] == kAA3[i][1] && name[2] == kAA3[i][2]) return kAA1[i];
    return '-';
}

inline bool is_space(char c) { return c == ' ' || (c >= '\t' && c <= '\r'); }

// float(field) of the reference (:230). Fast path: [spaces][sign]digits[.digits][spaces] with at most 15 digits — the
// integer mantissa and the power of ten are both exact doubles, so ONE correctly rounded division gives the correctly rounded
// value, i.e. what strtod / Python return (Clinger's fast path). Everything else (exponents, inf / nan, long fields) goes
// through strtod on a copy.
bool parse_double(const char *p, size_t n, size_t a, size_t b, double *out) {
    if (n <= a) return false;
    b = std::min(b, n);
    const char *s = p + a, *e = p + b;
    while (s < e && is_space(*s)) ++s;
    while (e > s && is_space(e[-1])) --e;
    if (s == e) return false;
    static const double kPow10[16] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15};
    const char *q = s;
    bool neg = false;
    if (*q == '-' || *q == '+') { neg = *q == '-'; ++q; }
    uint64_t m = 0;
    int digits = 0, frac = 0;
    bool dot = false, ok = q < e;
    for (; q < e; ++q) {
        const char c = *q;
        if (c >= '0' && c <= '9') { m = m * 10 + (uint64_t)(c - '0'); ++digits; frac += dot; }
        else if (c == '.' && !dot) dot = true;
        else { ok = false; break; }
    }
    if (ok && digits >= 1 && digits <= 15) {
        const double v = (double)m / kPow10[frac];
        *out = neg ? -v : v;
        return true;
    }
    char buf[64];
    const size_t len = (size_t)(e - s);
    if (len >= sizeof(buf)) return false;
    memcpy(buf, s, len);
    buf[len] = '\0';
    if (memchr(buf, '\0', len) != nullptr && strlen(buf) != len) return false;     // an embedded NUL is not a number
    char *end = nullptr;
    *out = strtod(buf, &end);
    return end != buf && *end == '\0';
}

// bytes.decode("utf-8", "ignore") of the reference (:211), column-exact: a well-formed multi-byte sequence is ONE column
// (kept as '?': no ATOM field compares equal to a non-ASCII character), ill-formed bytes vanish (maximal-subpart rule, as
// CPython's decoder applies it), ASCII passes through.
std::string utf8_ignore(const char *in, size_t n) {
    std::string out;
    out.reserve(n);
    size_t i = 0;
    while (i < n) {
        const unsigned char b = (unsigned char)in[i];
        if (b < 0x80) { out.push_back((char)b); ++i; continue; }
        int need = 0;
        unsigned char lo = 0x80, hi = 0xBF;
        if (b >= 0xC2 && b <= 0xDF) need = 1;
        else if (b == 0xE0) { need = 2; lo = 0xA0; }
        else if (b >= 0xE1 && b <= 0xEC) need = 2;
        else if (b == 0xED) { need = 2; hi = 0x9F; }
        else if (b >= 0xEE && b <= 0xEF) need = 2;
        else if (b == 0xF0) { need = 3; lo = 0x90; }
        else if (b >= 0xF1 && b <= 0xF3) need = 3;
        else if (b == 0xF4) { need = 3; hi = 0x8F; }
        else { ++i; continue; }                                   // 80..C1, F5..FF: never a lead byte
        size_t j = i + 1;
        int got = 0;
        while (got < need && j < n) {
            const unsigned char c = (unsigned char)in[j];
            if (c < lo || c > hi) break;
            lo = 0x80; hi = 0xBF;
            ++j; ++got;
        }
        if (got == need) out.push_back('?');
        i = j;                                                    // ill-formed: the lead and its valid prefix are dropped
    }
    return out;
}

void replace_all(std::string &s, const char *from, const char *to) {
    const size_t nf = strlen(from), nt = strlen(to);
    for (size_t p = s.find(from); p != std::string::npos; p = s.find(from, p + nt)) s.replace(p, nf, to);
}

bool read_file(const char *path, std::vector<char> *buf) {
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return false;
    struct stat st;
    size_t cap = (fstat(fd, &st) == 0 && st.st_size > 0) ? (size_t)st.st_size + 1 : 1 << 16, len = 0;
    buf->resize(cap);
    for (;;) {
        if (len == cap) { cap *= 2; buf->resize(cap); }
        const ssize_t r = read(fd, buf->data() + len, cap - len);
        if (r < 0) { if (errno == EINTR) continue; close(fd); return false; }
        if (r == 0) break;
        len += (size_t)r;
    }
    close(fd);
    buf->resize(len);
    return true;
}

enum LineResult { kSkip, kTaken, kBad, kSpan };

// One ATOM record (already decoded, right-stripped, MSE-rewritten): columns as at :222-231.
inline LineResult take_atom(const char *p, size_t n, const bool *want, bool any_wanted, ChainAcc *acc) {
    if (n < 22 || p[0] != 'A' || p[1] != 'T' || p[2] != 'O' || p[3] != 'M') return kSkip;
    const unsigned char ch = (unsigned char)p[21];
    if (any_wanted && !want[ch]) return kSkip;
    // residue number + insertion code, columns 22-26 stripped
    size_t ra = 22, rb = std::min(n, (size_t)27);
    while (ra < rb && is_space(p[ra])) ++ra;
    while (rb > ra && is_space(p[rb - 1])) --rb;
    double xyz[3];
    if (ra == rb || !parse_double(p, n, 30, 38, &xyz[0]) || !parse_double(p, n, 38, 46, &xyz[1]) ||
        !parse_double(p, n, 46, 54, &xyz[2])) return kBad;
    char ins = 0;
    if (isalpha((unsigned char)p[rb - 1])) { ins = p[rb - 1]; --rb; }
    size_t q = ra;
    bool neg = false;
    if (q < rb && (p[q] == '-' || p[q] == '+')) { neg = p[q] == '-'; ++q; }
    if (q == rb) return kBad;
    long num = 0;
    for (; q < rb; ++q) {
        if (p[q] < '0' || p[q] > '9') return kBad;
        num = num * 10 + (p[q] - '0');
    }
    num = (neg ? -num : num) - 1;
    ChainAcc &c = acc[ch];
    if (!c.any) { c.lo = c.hi = num; c.any = true; }
    c.lo = std::min(c.lo, num);
    c.hi = std::max(c.hi, num);
    if (c.hi - c.lo >= kMaxChainSpan) return kSpan;
    Residue *r;
    if (c.last >= 0 && c.res[c.last].num == num && c.res[c.last].ins == ins) {
        r = &c.res[c.last];
    } else {
        const int64_t key = (int64_t)num * 256 + (unsigned char)ins;
        auto it = c.index.find(key);
        if (it == c.index.end()) {
            if (!c.res.empty()) {
                const Residue &b = c.res.back();
                if (num < b.num || (num == b.num && (unsigned char)ins < (unsigned char)b.ins)) c.sorted = false;
            }
            c.index.emplace(key, (int)c.res.size());
            c.last = (int)c.res.size();
            c.res.emplace_back();
            r = &c.res.back();
            r->num = num;
            r->ins = ins;
            memcpy(r->name, p + 17, 3);                       // n >= 22: the three name columns exist
        } else {
            c.last = it->second;
            r = &c.res[c.last];
        }
    }
    // atom name, columns 12-15 stripped
    size_t aa = 12, ab = 16;
    while (aa < ab && is_space(p[aa])) ++aa;
    while (ab > aa && is_space(p[ab - 1])) --ab;
    int ai = -1;
    if (ab - aa == 1) ai = p[aa] == 'N' ? 0 : p[aa] == 'C' ? 2 : p[aa] == 'O' ? 3 : -1;
    else if (ab - aa == 2 && p[aa] == 'C' && p[aa + 1] == 'A') ai = 1;
    if (ai >= 0 && !(r->have & (1u << ai))) { r->have |= 1u << ai; memcpy(r->xyz[ai], xyz, sizeof(xyz)); }
    return kTaken;
}

}  // namespace

struct tmpnn_pdb {
    std::vector<float> X;            // [L,4,3], NaN kept (fill() zeroes them)
    std::vector<int32_t> S, ridx, cenc;
    std::vector<float> mask;
    std::string seq;                 // parser alphabet, '-' for gaps
    int n_chains = 0;
};

static int parse_one(const char *path, const char *chains, tmpnn_pdb **out, std::string *err) {
    std::vector<char> file;
    if (!read_file(path, &file)) { *err = std::string("cannot open ") + path; return TMPNN_E_INVALID; }
    const std::string want_s = chains ? chains : "";
    bool want[256] = {false};
    for (unsigned char c : want_s) want[c] = true;
    if (want_s.empty()) {            // no filter = the reference's default chain alphabet (protein_mpnn_utils.py:286-293): records of any
        for (int c = 'A'; c <= 'Z'; ++c) want[c] = true;      // other chain id (a blank one, punctuation) are never looked at there, so a
        for (int c = 'a'; c <= 'z'; ++c) want[c] = true;      // malformed record in such a chain must not fail the file here
        for (int c = '0'; c <= '9'; ++c) want[c] = true;
    }
    std::vector<ChainAcc> acc(256);
    std::string scratch;
    bool bad = false, span = false;
    const char *p = file.data(), *end = p + file.size();
    while (p < end && !bad && !span) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *le = nl ? nl : end;
        const char *lp = p;
        size_t n = (size_t)(le - lp);
        p = nl ? nl + 1 : end;
        bool ascii = true;
        for (size_t i = 0; i < n; ++i) if ((unsigned char)lp[i] >= 0x80) { ascii = false; break; }
        if (!ascii) {                                          // rare: decode, then look at the decoded columns
            scratch = utf8_ignore(lp, n);
            lp = scratch.data();
            n = scratch.size();
        }
        while (n > 0 && is_space(lp[n - 1])) --n;
        if (n >= 20 && memcmp(lp, "HETATM", 6) == 0 && memcmp(lp + 17, "MSE", 3) == 0) {
            std::string t(lp, n);                              // (scratch may alias lp)
            replace_all(t, "HETATM", "ATOM  ");
            replace_all(t, "MSE", "MET");
            scratch.swap(t);
            lp = scratch.data();
            n = scratch.size();
        }
        const LineResult r = take_atom(lp, n, want, true, acc.data());
        bad = r == kBad;
        span = r == kSpan;
    }
    if (span) { *err = std::string("residue numbers of one chain span more than 200000 in ") + path; return TMPNN_E_INVALID; }
    if (bad) { *err = std::string("malformed ATOM record in ") + path; return TMPNN_E_INVALID; }

    std::string order = want_s;
    if (order.empty()) {             // the reference's default chain alphabet: A-Z, a-z, then digits as found
        for (char c = 'A'; c <= 'Z'; ++c) order.push_back(c);
        for (char c = 'a'; c <= 'z'; ++c) order.push_back(c);
        for (char c = '0'; c <= '9'; ++c) order.push_back(c);
    }
    std::unique_ptr<tmpnn_pdb> holder(new tmpnn_pdb());      // released to the caller at the end; freed if anything below throws
    tmpnn_pdb *pd = holder.get();
    size_t rows = 0;
    for (unsigned char ch : order)
        if (acc[ch].any) rows += std::max<size_t>(acc[ch].res.size(), (size_t)(acc[ch].hi - acc[ch].lo + 1));
    pd->X.reserve(rows * 12); pd->S.reserve(rows); pd->ridx.reserve(rows); pd->cenc.reserve(rows); pd->mask.reserve(rows);
    pd->seq.reserve(rows);
    int cnum = 1;
    long pos = 0;
    const float nanf_ = nanf("");
    for (unsigned char ch : order) {
        ChainAcc &c = acc[ch];
        if (!c.any) continue;
        if (!c.sorted)
            std::stable_sort(c.res.begin(), c.res.end(), [](const Residue &a, const Residue &b) {
                return a.num != b.num ? a.num < b.num : (unsigned char)a.ins < (unsigned char)b.ins; });
        auto emit = [&](const Residue *r) {
            const char aa = r ? one_letter(r->name) : '-';
            pd->seq.push_back(aa);
            const char m = aa == '-' ? 'X' : aa;
            pd->S.push_back((int32_t)(strchr(kMpnn, m) - kMpnn));
            bool finite = true;
            for (int a = 0; a < 4; ++a)
                for (int k = 0; k < 3; ++k) {
                    const bool ok = r && (r->have & (1u << a));
                    pd->X.push_back(ok ? (float)r->xyz[a][k] : nanf_);
                    finite = finite && ok && std::isfinite(r->xyz[a][k]);
                }
            pd->mask.push_back(finite ? 1.f : 0.f);
            pd->ridx.push_back((int32_t)(100 * (cnum - 1) + pos));
            pd->cenc.push_back(cnum);
            ++pos;
        };
        size_t k = 0;
        for (long num = c.lo; num <= c.hi; ++num) {
            if (k < c.res.size() && c.res[k].num == num)
                for (; k < c.res.size() && c.res[k].num == num; ++k) emit(&c.res[k]);   // insertion codes sorted, none first
            else
                emit(nullptr);
        }
        ++cnum;
        ++pd->n_chains;
    }
    *out = holder.release();
    return TMPNN_OK;
}

extern "C" int tmpnn_pdb_parse(const char *path, const char *chains, tmpnn_pdb_t **out) {
    if (!path || !out) return tm_set_error(TMPNN_E_INVALID, "pdb_parse: null argument");
    *out = nullptr;
    return tm_host_guard("pdb_parse", [&]() -> int {
        std::string err;
        int rc = parse_one(path, chains, out, &err);
        if (rc != TMPNN_OK) return tm_set_error(rc, "pdb_parse: %s", err.c_str());
        return TMPNN_OK;
    });
}

// status (may be NULL) [n]: per-file result code. With it a failing file does not void the batch: its handle stays NULL, its code
// says why, the call returns TMPNN_OK and the caller decides (skip / report); without it the first failure fails the whole batch
// (nothing is handed out) and the message names every failing file (up to eight).
extern "C" int tmpnn_pdb_parse_batch_status(const char *const *paths, const char *const *chains, int n, int n_threads,
                                            tmpnn_pdb_t **outs, int32_t *status) {
    if (n < 0 || (n > 0 && (!paths || !outs))) return tm_set_error(TMPNN_E_INVALID, "pdb_parse_batch: bad argument");
    for (int i = 0; i < n; ++i) outs[i] = nullptr;
    if (n_threads < 1) n_threads = 1;
    n_threads = std::min(n_threads, std::max(n, 1));
    const int rc = tm_host_guard("pdb_parse_batch", [&]() -> int {
        std::atomic<int> next(0), n_failed(0);
        std::vector<std::string> errs(n);
        std::vector<int> codes(n, TMPNN_OK);
        tm_run_pool(n_threads, [&]() {
            for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) {
                codes[i] = parse_one(paths[i], chains ? chains[i] : nullptr, &outs[i], &errs[i]);
                if (codes[i] != TMPNN_OK) n_failed.fetch_add(1);
            }
        });
        if (status) for (int i = 0; i < n; ++i) status[i] = codes[i];
        if (n_failed.load() > 0) {
            std::string msg;
            int shown = 0;
            for (int i = 0; i < n && shown < 8; ++i)
                if (codes[i] != TMPNN_OK) { msg += (shown++ ? "; " : ""); msg += errs[i]; }
            if (n_failed.load() > shown) msg += "; ... (" + std::to_string(n_failed.load()) + " files failed)";
            tm_set_error(TMPNN_E_INVALID, "pdb_parse_batch: %s", msg.c_str());      // (readable through tmpnn_last_error in both forms)
            if (!status) return TMPNN_E_INVALID;
        }
        return TMPNN_OK;
    });
    if (rc != TMPNN_OK)                       // a failed file (no status array) or an exception (out of memory) in any worker: nothing is handed out
        for (int i = 0; i < n; ++i) { delete outs[i]; outs[i] = nullptr; }
    return rc;
}
extern "C" int tmpnn_pdb_parse_batch(const char *const *paths, const char *const *chains, int n, int n_threads,
                                     tmpnn_pdb_t **outs) {
    return tmpnn_pdb_parse_batch_status(paths, chains, n, n_threads, outs, nullptr);
}

extern "C" int64_t tmpnn_pdb_length(const tmpnn_pdb_t *p) { return p ? (int64_t)p->S.size() : -1; }
extern "C" int tmpnn_pdb_num_chains(const tmpnn_pdb_t *p) { return p ? p->n_chains : -1; }

extern "C" int tmpnn_pdb_fill(const tmpnn_pdb_t *p, float *X, int32_t *S, float *mask, int32_t *residue_idx,
                              int32_t *chain_enc, char *seq, float *ca_mask) {
    if (!p) return tm_set_error(TMPNN_E_INVALID, "pdb_fill: null handle");
    const size_t L = p->S.size();
    if (L == 0) {                    // an empty structure (no ATOM record of the requested chains): nothing to copy —
        if (seq) seq[0] = '\0';      // and memcpy from an empty vector's null data() is undefined even for 0 bytes (UBSan)
        return TMPNN_OK;
    }
    if (X) for (size_t i = 0; i < L * 12; ++i) X[i] = std::isnan(p->X[i]) ? 0.f : p->X[i];
    if (S) memcpy(S, p->S.data(), L * sizeof(int32_t));
    if (mask) memcpy(mask, p->mask.data(), L * sizeof(float));
    if (ca_mask)        // compute_centrality masks on the CA atom only (thermompnn_benchmarking.py:20-27)
        for (size_t i = 0; i < L; ++i)
            ca_mask[i] = (std::isnan(p->X[i * 12 + 3]) || std::isnan(p->X[i * 12 + 4]) || std::isnan(p->X[i * 12 + 5])) ? 0.f : 1.f;
    if (residue_idx) memcpy(residue_idx, p->ridx.data(), L * sizeof(int32_t));
    if (chain_enc) memcpy(chain_enc, p->cenc.data(), L * sizeof(int32_t));
    if (seq) { memcpy(seq, p->seq.data(), L); seq[L] = '\0'; }
    return TMPNN_OK;
}

extern "C" void tmpnn_pdb_free(tmpnn_pdb_t *p) { delete p; }

extern "C" const char *tmpnn_pdb_seq(const tmpnn_pdb_t *p) { return p ? p->seq.c_str() : nullptr; }

// Many parsed structures -> ONE ragged batch in the caller's (pinned) host buffers, protein after protein: the layout
// tmpnn_ssm_forward consumes, so a many-PDB scan goes file -> handle -> staging buffer -> one async H2D copy with no
// per-protein Python object in between.
extern "C" int tmpnn_pdb_pack_batch(tmpnn_pdb_t *const *handles, int n, int n_threads, int64_t capacity, float *X,
                                    int32_t *S, float *mask, int32_t *residue_idx, int32_t *chain_enc, float *ca_mask,
                                    int32_t *offsets) {
    if (n < 0 || (n > 0 && !handles) || !offsets) return tm_set_error(TMPNN_E_INVALID, "pdb_pack_batch: bad argument");
    int64_t tot = 0;
    offsets[0] = 0;
    for (int i = 0; i < n; ++i) {
        if (!handles[i]) return tm_set_error(TMPNN_E_INVALID, "pdb_pack_batch: handle %d is null", i);
        tot += (int64_t)handles[i]->S.size();
        if (tot > INT32_MAX) return tm_set_error(TMPNN_E_INVALID, "pdb_pack_batch: more than 2^31 residues in one batch");
        offsets[i + 1] = (int32_t)tot;
    }
    if (tot > capacity)
        return tm_set_error(TMPNN_E_WORKSPACE, "pdb_pack_batch: %lld residues, buffers hold %lld", (long long)tot, (long long)capacity);
    return tm_host_guard("pdb_pack_batch", [&]() -> int {
        std::atomic<int> next(0);
        tm_run_pool(std::max(1, std::min(n_threads, n)), [&]() {
            for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) {
                const size_t o = (size_t)offsets[i];
                tmpnn_pdb_fill(handles[i], X ? X + o * 12 : nullptr, S ? S + o : nullptr, mask ? mask + o : nullptr,
                               residue_idx ? residue_idx + o : nullptr, chain_enc ? chain_enc + o : nullptr, nullptr,
                               ca_mask ? ca_mask + o : nullptr);
            }
        });
        return TMPNN_OK;
    });
}
