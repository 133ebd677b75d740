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

#include "tmpnn_internal.h"

namespace {

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
    bool bad = false;
    while (fgets(buf, sizeof(buf), fh)) {
        line.assign(buf);
        while (!line.empty() && strchr("\r\n", line.back()) == nullptr && !feof(fh) && line.size() % (sizeof(buf) - 1) == 0) {
            if (!fgets(buf, sizeof(buf), fh)) break;     // very long line: keep reading
            line += buf;
        }
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
        Residue &r = c.res[num][ins];
        if (r.name.empty()) r.name = resname;
        int ai = atom == "N" ? 0 : atom == "CA" ? 1 : atom == "C" ? 2 : atom == "O" ? 3 : -1;
        if (ai >= 0 && !r.have[ai]) { r.have[ai] = true; memcpy(r.xyz[ai], xyz, sizeof(xyz)); }
    }
    fclose(fh);
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
