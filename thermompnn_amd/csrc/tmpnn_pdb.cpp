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
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <string>
#include <thread>
#include <vector>

// Host-only translation unit: it needs the C-ABI header and the library's error sink, nothing of HIP — so the same file
// also builds with plain g++ under -fsanitize=address,undefined into the fuzz driver of tests/native/pdb_fuzz_driver.cpp
// (python -m thermompnn_amd.build --pdb-sanitizer-driver).
#include "../../include/tmpnn.h"
int tm_set_error(int code, const char *fmt, ...);

namespace {

// Residue numbers come from a 5-column field (-9999 .. 99999), so a chain can span at most ~110 000 numbers; the bound is
// enforced anyway: every number between a chain's lowest and highest becomes a row (gaps included), and a hostile file must
// get an error, not an allocation proportional to a number it made up.
constexpr long kMaxChainSpan = 200000;

struct Residue {
    std::string name;           // first residue name seen
    bool have[4] = {false, false, false, false};
    double xyz[4][3];
};
struct ChainAcc {
    std::map<long, std::map<std::string, Residue>> res;   // number-1 -> insertion code -> residue
    long lo = 0, hi = 0;
    bool any = false;
};

const char *kAA3[20] = {"ALA", "ARG", "ASN", "ASP", "CYS", "GLN", "GLU", "GLY", "HIS", "ILE",
                        "LEU", "LYS", "MET", "PHE", "PRO", "SER", "THR", "TRP", "TYR", "VAL"};
const char kAA1[21] = "ARNDCQEGHILKMFPSTWYV";
const char kMpnn[22] = "ACDEFGHIKLMNPQRSTVWYX";

char one_letter(const std::string &name) {
    for (int i = 0; i < 20; ++i)
        if (name == kAA3[i]) return kAA1[i];
    return '-';
}

bool parse_double(const std::string &line, size_t a, size_t b, double *out) {
    if (line.size() <= a) return false;
    std::string f = line.substr(a, std::min(b, line.size()) - a);
    const char *p = f.c_str();
    while (*p && isspace((unsigned char)*p)) ++p;
    if (!*p) return false;
    char *end = nullptr;
    *out = strtod(p, &end);
    while (*end && isspace((unsigned char)*end)) ++end;
    return end != p && *end == '\0';
}

std::string strip(const std::string &s) {
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char)s[a])) ++a;
    while (b > a && isspace((unsigned char)s[b - 1])) --b;
    return s.substr(a, b - a);
}

// bytes.decode("utf-8", "ignore") of the reference (:211), column-exact: a well-formed multi-byte sequence is ONE column
// (kept as '?': no ATOM field compares equal to a non-ASCII character), ill-formed bytes vanish (maximal-subpart rule, as
// CPython's decoder applies it), ASCII passes through.
std::string utf8_ignore(const std::string &in) {
    std::string out;
    out.reserve(in.size());
    const size_t n = in.size();
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

}  // namespace

struct tmpnn_pdb {
    std::vector<float> X;            // [L,4,3], NaN kept (fill() zeroes them)
    std::vector<int32_t> S, ridx, cenc;
    std::vector<float> mask;
    std::string seq;                 // parser alphabet, '-' for gaps
    int n_chains = 0;
};

static int parse_one(const char *path, const char *chains, tmpnn_pdb **out, std::string *err) {
    FILE *fh = fopen(path, "rb");
    if (!fh) { *err = std::string("cannot open ") + path; return TMPNN_E_INVALID; }
    std::string want = chains ? chains : "";
    std::map<char, ChainAcc> acc;
    std::vector<char> first_seen;    // default order = the reference's A-Z, a-z scan; here: requested or alphabet order
    std::string line;
    char buf[512];
    bool bad = false, span = false;
    while (fgets(buf, sizeof(buf), fh)) {
        line.assign(buf);
        while (!line.empty() && strchr("\r\n", line.back()) == nullptr && !feof(fh) && line.size() % (sizeof(buf) - 1) == 0) {
            if (!fgets(buf, sizeof(buf), fh)) break;     // very long line: keep reading
            line += buf;
        }
        line = utf8_ignore(line);
        while (!line.empty() && isspace((unsigned char)line.back())) line.pop_back();
        if (line.compare(0, 6, "HETATM") == 0 && line.size() >= 20 && line.compare(17, 3, "MSE") == 0) {
            replace_all(line, "HETATM", "ATOM  ");
            replace_all(line, "MSE", "MET");
        }
        if (line.compare(0, 4, "ATOM") != 0 || line.size() < 22) continue;
        const char ch = line[21];
        if (!want.empty() && want.find(ch) == std::string::npos) continue;
        const std::string atom = strip(line.substr(12, 4));
        const std::string resname = line.size() >= 20 ? line.substr(17, 3) : line.substr(17);
        std::string resn = strip(line.size() >= 27 ? line.substr(22, 5) : line.substr(22));
        double xyz[3];
        if (resn.empty() || !parse_double(line, 30, 38, &xyz[0]) || !parse_double(line, 38, 46, &xyz[1]) ||
            !parse_double(line, 46, 54, &xyz[2])) { bad = true; break; }
        std::string ins;
        if (isalpha((unsigned char)resn.back())) { ins = resn.substr(resn.size() - 1); resn.pop_back(); }
        char *end = nullptr;
        const long num = strtol(resn.c_str(), &end, 10) - 1;
        if (end == resn.c_str() || *end != '\0') { bad = true; break; }
        ChainAcc &c = acc[ch];
        if (!c.any) { c.lo = c.hi = num; c.any = true; }
        c.lo = std::min(c.lo, num);
        c.hi = std::max(c.hi, num);
        if (c.hi - c.lo >= kMaxChainSpan) { span = true; break; }
        Residue &r = c.res[num][ins];
        if (r.name.empty()) r.name = resname;
        int ai = atom == "N" ? 0 : atom == "CA" ? 1 : atom == "C" ? 2 : atom == "O" ? 3 : -1;
        if (ai >= 0 && !r.have[ai]) { r.have[ai] = true; memcpy(r.xyz[ai], xyz, sizeof(xyz)); }
    }
    fclose(fh);
    if (span) { *err = std::string("residue numbers of one chain span more than 200000 in ") + path; return TMPNN_E_INVALID; }
    if (bad) { *err = std::string("malformed ATOM record in ") + path; return TMPNN_E_INVALID; }

    std::string order = want;
    if (order.empty()) {             // the reference's default chain alphabet: A-Z, a-z, then digits as found
        for (char c = 'A'; c <= 'Z'; ++c) order.push_back(c);
        for (char c = 'a'; c <= 'z'; ++c) order.push_back(c);
        for (char c = '0'; c <= '9'; ++c) order.push_back(c);
    }
    tmpnn_pdb *p = new tmpnn_pdb();
    int cnum = 1;
    long pos = 0;
    const float nanf_ = nanf("");
    for (char ch : order) {
        auto it = acc.find(ch);
        if (it == acc.end()) continue;
        const ChainAcc &c = it->second;
        for (long num = c.lo; num <= c.hi; ++num) {
            auto rit = c.res.find(num);
            auto emit = [&](const Residue *r) {
                const char aa = r ? one_letter(r->name) : '-';
                p->seq.push_back(aa);
                const char m = aa == '-' ? 'X' : aa;
                p->S.push_back((int32_t)(strchr(kMpnn, m) - kMpnn));
                bool finite = true;
                for (int a = 0; a < 4; ++a)
                    for (int k = 0; k < 3; ++k) {
                        const bool ok = r && r->have[a];
                        p->X.push_back(ok ? (float)r->xyz[a][k] : nanf_);
                        finite = finite && ok && std::isfinite(r->xyz[a][k]);
                    }
                p->mask.push_back(finite ? 1.f : 0.f);
                p->ridx.push_back((int32_t)(100 * (cnum - 1) + pos));
                p->cenc.push_back(cnum);
                ++pos;
            };
            if (rit == c.res.end()) emit(nullptr);
            else for (const auto &kv : rit->second) emit(&kv.second);   // std::map iterates insertion codes sorted, "" first
        }
        ++cnum;
        ++p->n_chains;
    }
    *out = p;
    return TMPNN_OK;
}

extern "C" int tmpnn_pdb_parse(const char *path, const char *chains, tmpnn_pdb_t **out) {
    if (!path || !out) return tm_set_error(TMPNN_E_INVALID, "pdb_parse: null argument");
    std::string err;
    int rc = parse_one(path, chains, out, &err);
    if (rc != TMPNN_OK) return tm_set_error(rc, "pdb_parse: %s", err.c_str());
    return TMPNN_OK;
}

extern "C" int tmpnn_pdb_parse_batch(const char *const *paths, const char *const *chains, int n, int n_threads,
                                     tmpnn_pdb_t **outs) {
    if (n < 0 || (n > 0 && (!paths || !outs))) return tm_set_error(TMPNN_E_INVALID, "pdb_parse_batch: bad argument");
    for (int i = 0; i < n; ++i) outs[i] = nullptr;
    if (n_threads < 1) n_threads = 1;
    n_threads = std::min(n_threads, std::max(n, 1));
    std::atomic<int> next(0), failed(-1);
    std::vector<std::string> errs(n);
    auto work = [&]() {
        for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) {
            if (parse_one(paths[i], chains ? chains[i] : nullptr, &outs[i], &errs[i]) != TMPNN_OK) {
                int exp = -1;
                failed.compare_exchange_strong(exp, i);
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < n_threads; ++t) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
    const int f = failed.load();
    if (f >= 0) {
        for (int i = 0; i < n; ++i) { delete outs[i]; outs[i] = nullptr; }
        return tm_set_error(TMPNN_E_INVALID, "pdb_parse_batch: %s", errs[f].c_str());
    }
    return TMPNN_OK;
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
