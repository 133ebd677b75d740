// A host WITHOUT Python: PDB files -> the reference's SSM CSV through libtmpnn.so's C-ABI only (include/tmpnn.h) and the HIP
// runtime for memory, copies and one stream. What the Python package does with torch tensors (thermompnn_amd/pipeline.py) in
// its plainest form: one chunk of files at a time, no stage overlap. The CSV is byte-identical to
//     python -m thermompnn_amd.ssm_scan FILES --chain A --out OUT.csv
// (tests/test_gpu_e2e.py::test_native_host_example_matches_python_pipeline). Replaces, for such a host, the loop of
// analysis/SSM.py:105-176 of the reference.
//
//   python -m thermompnn_amd.weights weights.raw --model_path thermoMPNN_default.pt --vanilla_path v_48_020.pt
//   scan_native weights.raw out.csv [--chain A] [--threads N] [--precision f16x2|bf16x3|fp32] [--chunk_files N] a.pdb b.pdb ...
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "tmpnn.h"

namespace {

[[noreturn]] void die(const std::string &what) {
    std::fprintf(stderr, "scan_native: %s\n", what.c_str());
    std::exit(1);
}
void hip_ok(hipError_t e, const char *what) {
    if (e != hipSuccess) die(std::string(what) + ": " + hipGetErrorString(e));
}
void tm_ok(int rc, const char *what) {
    if (rc != TMPNN_OK) die(std::string(what) + ": " + tmpnn_last_error());
}
template <class T> T *dev_alloc(size_t n) {
    void *p = nullptr;
    hip_ok(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)), "hipMalloc");
    return static_cast<T *>(p);
}
template <class T> T *pinned_alloc(size_t n) {
    void *p = nullptr;
    hip_ok(hipHostMalloc(&p, std::max<size_t>(n, 1) * sizeof(T), hipHostMallocDefault), "hipHostMalloc");
    return static_cast<T *>(p);
}

// The flat weight file of thermompnn_amd.weights.export_raw -> device tensors in the library's canonical order.
struct RawWeights {
    std::vector<const float *> tensors;
    char *arena = nullptr;
};
RawWeights load_raw(const char *path) {
    FILE *f = std::fopen(path, "rb");
    if (!f) die(std::string("cannot open ") + path);
    char magic[8];
    int32_t n = 0;
    if (std::fread(magic, 1, 8, f) != 8 || std::memcmp(magic, "TMPNNRAW", 8) || std::fread(&n, 4, 1, f) != 1)
        die(std::string(path) + ": not a TMPNNRAW weight file");
    if (n <= 0 || n > tmpnn_num_tensors()) die(std::string(path) + ": tensor count the library does not know");
    std::vector<size_t> off(n + 1, 0);
    for (int i = 0; i < n; ++i)                                            // 256-byte aligned slots in one arena
        off[i + 1] = off[i] + ((static_cast<size_t>(tmpnn_tensor_numel(i)) * 4 + 255) & ~size_t(255));
    RawWeights w;
    w.arena = dev_alloc<char>(off[n]);
    std::vector<float> host;
    for (int i = 0; i < n; ++i) {
        int64_t numel = 0;
        if (std::fread(&numel, 8, 1, f) != 1 || numel != tmpnn_tensor_numel(i))
            die(std::string(path) + ": tensor " + tmpnn_tensor_name(i) + " has the wrong size");
        host.resize(static_cast<size_t>(numel));
        if (std::fread(host.data(), 4, host.size(), f) != host.size()) die(std::string(path) + ": truncated");
        hip_ok(hipMemcpy(w.arena + off[i], host.data(), host.size() * 4, hipMemcpyHostToDevice), "hipMemcpy(weights)");
        w.tensors.push_back(reinterpret_cast<const float *>(w.arena + off[i]));
    }
    std::fclose(f);
    return w;
}

tmpnn_weights_t *make_handle(const RawWeights &raw, const char *precision, hipStream_t st) {
    const size_t nbytes = tmpnn_weights_packed_bytes_p(precision);
    if (!nbytes) die(std::string("unknown precision ") + (precision ? precision : "(null)"));
    char *packed = dev_alloc<char>(nbytes);                               // lives as long as the handle (process lifetime here)
    tmpnn_weights_t *h = nullptr;
    tm_ok(tmpnn_weights_create_p(&h, raw.tensors.data(), static_cast<int>(raw.tensors.size()), packed, nbytes, precision, st),
          "tmpnn_weights_create_p");
    return h;
}

// Python's os.path.basename(p)[:-4].strip(".pdb"): the 'pdb' cell of the reference's frame (analysis/SSM.py:128).
std::string pdb_cell(const std::string &path) {
    std::string b = path.substr(path.find_last_of('/') == std::string::npos ? 0 : path.find_last_of('/') + 1);
    b = b.size() > 4 ? b.substr(0, b.size() - 4) : std::string();
    const char *set = ".pdb";
    size_t a = 0, e = b.size();
    while (a < e && std::strchr(set, b[a])) ++a;
    while (e > a && std::strchr(set, b[e - 1])) --e;
    return b.substr(a, e - a);
}

}  // namespace

int main(int argc, char **argv) {
    std::string chain = "A", precision;
    int threads = static_cast<int>(std::max(1u, std::thread::hardware_concurrency())), chunk_files = 256;
    std::vector<std::string> pos;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() -> const char * { if (i + 1 >= argc) die(a + " needs a value"); return argv[++i]; };
        if (a == "--chain") chain = val();
        else if (a == "--threads") threads = std::max(1, std::atoi(val()));
        else if (a == "--precision") precision = val();
        else if (a == "--chunk_files") chunk_files = std::max(1, std::atoi(val()));
        else pos.push_back(a);
    }
    if (pos.size() < 3) die("usage: scan_native WEIGHTS.raw OUT.csv [--chain A] [--threads N] [--precision P] [--chunk_files N] FILE.pdb ...");
    const std::vector<std::string> files(pos.begin() + 2, pos.end());

    hipStream_t st;
    hip_ok(hipStreamCreate(&st), "hipStreamCreate");
    int32_t *d_status = dev_alloc<int32_t>(1), h_status = 0;
    hip_ok(hipMemsetAsync(d_status, 0, 4, st), "hipMemsetAsync");
    tm_ok(tmpnn_selftest(d_status, st), "tmpnn_selftest");                // is this build's device code what the host expects?
    hip_ok(hipMemcpyAsync(&h_status, d_status, 4, hipMemcpyDeviceToHost, st), "hipMemcpyAsync");
    hip_ok(hipStreamSynchronize(st), "hipStreamSynchronize");
    tm_ok(tmpnn_status_error(h_status), "tmpnn_selftest");

    const RawWeights raw = load_raw(pos[0].c_str());
    if (raw.tensors.size() != static_cast<size_t>(tmpnn_num_tensors())) die("the weight file has no ddG head (ProteinMPNN tensors only)");
    tmpnn_weights_t *w = make_handle(raw, precision.empty() ? nullptr : precision.c_str(), st), *w_retry = nullptr;

    tmpnn_csv_t *csv = nullptr;
    tm_ok(tmpnn_csv_open(pos[1].c_str(), 0, &csv), "tmpnn_csv_open");

    for (size_t first = 0; first < files.size(); first += chunk_files) {
        const int n = static_cast<int>(std::min<size_t>(chunk_files, files.size() - first));
        std::vector<const char *> paths(n), chains(n, chain.c_str());
        for (int i = 0; i < n; ++i) paths[i] = files[first + i].c_str();
        std::vector<tmpnn_pdb_t *> pdb(n, nullptr);
        tm_ok(tmpnn_pdb_parse_batch(paths.data(), chains.data(), n, threads, pdb.data()), "tmpnn_pdb_parse_batch");
        int64_t T = 0;
        int max_len = 1;
        for (int i = 0; i < n; ++i) {
            T += tmpnn_pdb_length(pdb[i]);
            max_len = std::max<int>(max_len, static_cast<int>(tmpnn_pdb_length(pdb[i])));
        }
        // host staging (pinned) -> device, one ragged batch: protein after protein along the residue axis
        float *hX = pinned_alloc<float>(T * 12), *hmask = pinned_alloc<float>(T), *htab = pinned_alloc<float>(T * 21);
        int32_t *hS = pinned_alloc<int32_t>(T), *hri = pinned_alloc<int32_t>(T), *hce = pinned_alloc<int32_t>(T),
                *hoff = pinned_alloc<int32_t>(n + 1);
        tm_ok(tmpnn_pdb_pack_batch(pdb.data(), n, threads, T, hX, hS, hmask, hri, hce, nullptr, hoff), "tmpnn_pdb_pack_batch");
        float *dX = dev_alloc<float>(T * 12), *dmask = dev_alloc<float>(T), *dtab = dev_alloc<float>(T * 21);
        int32_t *dS = dev_alloc<int32_t>(T), *dri = dev_alloc<int32_t>(T), *dce = dev_alloc<int32_t>(T), *doff = dev_alloc<int32_t>(n + 1);
        const size_t ws_bytes = tmpnn_workspace_bytes(T);
        char *ws = dev_alloc<char>(ws_bytes);
        auto h2d = [&](void *d, const void *h, size_t b) { hip_ok(hipMemcpyAsync(d, h, b, hipMemcpyHostToDevice, st), "hipMemcpyAsync"); };
        h2d(dX, hX, T * 48); h2d(dmask, hmask, T * 4); h2d(dS, hS, T * 4); h2d(dri, hri, T * 4); h2d(dce, hce, T * 4);
        h2d(doff, hoff, (n + 1) * 4);
        auto forward = [&](tmpnn_weights_t *handle) {
            if (T == 0) { h_status = 0; return; }
            tm_ok(tmpnn_ssm_forward(handle, dX, dS, dmask, dri, dce, doff, n, T, max_len, 48, dtab, nullptr, nullptr, nullptr,
                                    d_status, ws, ws_bytes, st), "tmpnn_ssm_forward");
            hip_ok(hipMemcpyAsync(htab, dtab, T * 84, hipMemcpyDeviceToHost, st), "hipMemcpyAsync");
            hip_ok(hipMemcpyAsync(&h_status, d_status, 4, hipMemcpyDeviceToHost, st), "hipMemcpyAsync");
            hip_ok(hipStreamSynchronize(st), "hipStreamSynchronize");
        };
        forward(w);
        if ((h_status & TMPNN_STATUS_RANGE) && std::strcmp(tmpnn_weights_precision(w), "f16x2") == 0) {
            // an operand left the fp16 range: the same batch once more on the full-range path (Engine.ssm_forward does the same)
            std::fprintf(stderr, "scan_native: f16x2 left the fp16 range in files %zu..%zu, rerunning in bf16x3\n", first, first + n - 1);
            if (!w_retry) w_retry = make_handle(raw, "bf16x3", st);
            forward(w_retry);
        }
        tm_ok(tmpnn_status_error(h_status), "tmpnn_ssm_forward");
        std::vector<std::string> cells(n);
        std::vector<const char *> seqs(n), names(n);
        for (int i = 0; i < n; ++i) {
            cells[i] = pdb_cell(files[first + i]);
            names[i] = cells[i].c_str();
            seqs[i] = tmpnn_pdb_seq(pdb[i]);
        }
        tm_ok(tmpnn_csv_write_ssm(csv, htab, 21, hoff, n, seqs.data(), nullptr, names.data(), nullptr, "ThermoMPNN", "custom", nullptr, nullptr,
                                  0, threads), "tmpnn_csv_write_ssm");
        for (tmpnn_pdb_t *p : pdb) tmpnn_pdb_free(p);
        for (void *p : {static_cast<void *>(dX), static_cast<void *>(dmask), static_cast<void *>(dtab), static_cast<void *>(dS),
                        static_cast<void *>(dri), static_cast<void *>(dce), static_cast<void *>(doff), static_cast<void *>(ws)})
            hip_ok(hipFree(p), "hipFree");
        for (void *p : {static_cast<void *>(hX), static_cast<void *>(hmask), static_cast<void *>(htab), static_cast<void *>(hS),
                        static_cast<void *>(hri), static_cast<void *>(hce), static_cast<void *>(hoff)})
            hip_ok(hipHostFree(p), "hipHostFree");
    }
    int64_t rows = 0, bytes = 0;
    tm_ok(tmpnn_csv_close(csv, &rows, &bytes), "tmpnn_csv_close");
    std::printf("%lld rows, %lld bytes -> %s\n", static_cast<long long>(rows), static_cast<long long>(bytes), pos[1].c_str());
    tmpnn_weights_destroy(w);
    if (w_retry) tmpnn_weights_destroy(w_retry);
    return 0;
}
