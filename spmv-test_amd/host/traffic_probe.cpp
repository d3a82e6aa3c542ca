// traffic_probe.cpp -- bin/spmv_traffic_probe: the program bench.py puts after `rocprofv3 --pmc <counter> --` to measure the
// HBM traffic of every workload of its line (role of the reference's profile.sh:18-20: a profiler around the executable).
// Native on purpose: a Python child spends minutes of a cold box importing torch under the profiler; this one links
// libspmv_hip.so and nothing else.
//
//   spmv_traffic_probe JOB OUT
// JOB (text, written by bench.py): one record per line
//   V <variant id> <runs> <device>                                  first line
//   W <id> <seed> <row0> <n_local> <rows> <cols> <band> <row_ptr file>   a block of a synthetic workload (spmv_synth_fill);
//                                                                    several W lines with one id = the blocks of one rank
//   F <id> <rows> <cols> <nnz> <row_ptr file> <col_idx file> <vals file>  a host-built matrix (raw int32 / int32 / float32)
// For every id: marker 100+id, plan (the first block's plan numbers set on the others), marker 300+id, `runs` runs of every
// block, marker 500+id.  Before that the calibration kernels of include/spmv_hip.h behind markers 1..5, marker 99.
// OUT: one line per id -- "<id>\t<plan description of the first block>".  Exit 0 iff everything ran.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "spmv_hip.h"

#define OK_HIP(call)                                                                                  \
    do {                                                                                              \
        hipError_t e__ = (call);                                                                      \
        if (e__ != hipSuccess) { fprintf(stderr, "HIP error %s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e__)); exit(EXIT_FAILURE); } \
    } while (0)
#define OK_SPMV(call)                                                                                 \
    do {                                                                                              \
        int rc__ = (call);                                                                            \
        if (rc__ != SPMV_OK) { fprintf(stderr, "error %s:%d: %s\n", __FILE__, __LINE__, spmv_last_error()); exit(EXIT_FAILURE); } \
    } while (0)

namespace {

constexpr int64_t kCalStream = 1ll << 30, kCalLines = 1ll << 24, kCalStore = 1ll << 28;   // (bench.py: CAL_*)

struct Block {
    char kind = 'W';
    uint64_t seed = 0;
    int64_t row0 = 0, n_local = 0, rows = 0, cols = 0, band = 0, nnz = 0;
    std::string rp_file, ci_file, va_file;
};

template <typename T>
std::vector<T> read_raw(const std::string &path, size_t count)
{
    std::vector<T> v(count);
    std::ifstream f(path, std::ios::binary);
    if (!f || !f.read(reinterpret_cast<char *>(v.data()), (std::streamsize)(sizeof(T) * count))) {
        fprintf(stderr, "error: cannot read %zu items from %s\n", count, path.c_str());
        exit(EXIT_FAILURE);
    }
    return v;
}

struct Resident {
    int32_t *d_rp = nullptr, *d_ci = nullptr;
    float *d_va = nullptr, *d_y = nullptr;
    spmv_csr_t *A = nullptr;
    int64_t rows = 0;
};

}  // namespace

int main(int argc, char **argv)
{
    if (argc != 3) { fprintf(stderr, "usage: spmv_traffic_probe JOB OUT\n"); return 2; }
    std::ifstream job(argv[1]);
    if (!job) { fprintf(stderr, "error: cannot open %s\n", argv[1]); return 2; }
    int variant = SPMV_AUTO, runs = 2, device = 0;
    std::map<int, std::vector<Block>> work;
    std::vector<int> order;
    std::string line;
    while (std::getline(job, line)) {
        std::istringstream in(line);
        char kind = 0;
        in >> kind;
        if (kind == 'V') in >> variant >> runs >> device;
        else if (kind == 'W' || kind == 'F') {
            int id = 0;
            Block b;
            b.kind = kind;
            in >> id;
            if (kind == 'W') in >> b.seed >> b.row0 >> b.n_local >> b.rows >> b.cols >> b.band >> b.rp_file;
            else { in >> b.rows >> b.cols >> b.nnz >> b.rp_file >> b.ci_file >> b.va_file; b.n_local = b.rows; }
            if (!in) { fprintf(stderr, "error: bad job line: %s\n", line.c_str()); return 2; }
            if (!work.count(id)) order.push_back(id);
            work[id].push_back(b);
        }
    }
    if (spmv_device_count() <= device) { fprintf(stderr, "HIP error: device %d not visible\n", device); return EXIT_FAILURE; }
    OK_HIP(hipSetDevice(device));
    hipStream_t s = nullptr;

    // ---- kernels of known traffic (the corrections of the counters come from these, in the same pass)
    {
        float *table = nullptr, *sink = nullptr;
        OK_HIP(hipMalloc((void **)&table, (size_t)kCalLines * 128));
        OK_HIP(hipMalloc((void **)&sink, 64));
        OK_HIP(hipMemset(table, 0, (size_t)kCalLines * 128));
        OK_HIP(hipMemset(sink, 0, 64));
        for (int id = 1; id <= 5; ++id) {
            OK_SPMV(spmv_calib_marker(id, s));
            for (int r = 0; r < runs; ++r) {
                if (id == 1) OK_SPMV(spmv_calib_stream(table, kCalStream, sink, s));
                else if (id == 2) OK_SPMV(spmv_calib_gather(table, kCalLines, kCalLines, 1, sink, s));
                else if (id == 3) OK_SPMV(spmv_calib_gather(table, kCalLines, kCalLines, 2, sink, s));
                else if (id == 4) OK_SPMV(spmv_calib_store(table, kCalStore, 4, s));
                else OK_SPMV(spmv_calib_store(table, kCalStore, 16, s));
            }
        }
        OK_SPMV(spmv_calib_marker(99, s));
        OK_HIP(hipDeviceSynchronize());
        OK_HIP(hipFree(table));
        OK_HIP(hipFree(sink));
    }

    std::ofstream out(argv[2]);
    for (int id : order) {
        const std::vector<Block> &blocks = work[id];
        std::vector<Resident> res(blocks.size());
        int64_t cols = blocks[0].cols;
        float *d_x = nullptr;
        OK_HIP(hipMalloc((void **)&d_x, sizeof(float) * (size_t)(cols > 0 ? cols : 1)));
        OK_SPMV(spmv_synth_x(blocks[0].kind == 'W' ? blocks[0].seed : 20251031ull, 0, cols, d_x, s));
        for (size_t i = 0; i < blocks.size(); ++i) {
            const Block &b = blocks[i];
            Resident &R = res[i];
            R.rows = b.n_local;
            std::vector<int32_t> rp = read_raw<int32_t>(b.rp_file, (size_t)b.n_local + 1);
            const int64_t nnz = rp[(size_t)b.n_local];
            OK_HIP(hipMalloc((void **)&R.d_rp, sizeof(int32_t) * ((size_t)b.n_local + 1)));
            OK_HIP(hipMalloc((void **)&R.d_ci, sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1)));
            OK_HIP(hipMalloc((void **)&R.d_va, sizeof(float) * (size_t)(nnz > 0 ? nnz : 1)));
            OK_HIP(hipMalloc((void **)&R.d_y, sizeof(float) * (size_t)(b.n_local > 0 ? b.n_local : 1)));
            OK_HIP(hipMemcpy(R.d_rp, rp.data(), sizeof(int32_t) * ((size_t)b.n_local + 1), hipMemcpyHostToDevice));
            if (b.kind == 'W') {
                OK_SPMV(spmv_synth_fill(b.seed, b.row0, b.n_local, b.rows, b.cols, b.band, R.d_rp, R.d_ci, R.d_va, s));
            } else {
                std::vector<int32_t> ci = read_raw<int32_t>(b.ci_file, (size_t)nnz);
                std::vector<float> va = read_raw<float>(b.va_file, (size_t)nnz);
                OK_HIP(hipMemcpy(R.d_ci, ci.data(), sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice));
                OK_HIP(hipMemcpy(R.d_va, va.data(), sizeof(float) * (size_t)nnz, hipMemcpyHostToDevice));
            }
            OK_SPMV(spmv_csr_create_device(b.n_local, b.cols, nnz, R.d_rp, R.d_ci, R.d_va, &R.A));
        }
        OK_HIP(hipDeviceSynchronize());
        OK_SPMV(spmv_calib_marker(100 + id, s));                       // plan region
        OK_SPMV(spmv_csr_plan(res[0].A, variant, s));
        int32_t params[8];
        OK_SPMV(spmv_csr_plan_get(res[0].A, variant, params));
        for (size_t i = 1; i < res.size(); ++i) OK_SPMV(spmv_csr_plan_set(res[i].A, variant, params, s));
        OK_SPMV(spmv_calib_marker(300 + id, s));                       // run region
        for (int r = 0; r < runs; ++r)
            for (Resident &R : res) OK_SPMV(spmv_csr_run(R.A, variant, d_x, R.d_y, s));
        OK_SPMV(spmv_calib_marker(500 + id, s));                       // ... ends here
        OK_HIP(hipDeviceSynchronize());
        char plan[256];                                            // (the length the Python binding asks for: the strings are compared)
        OK_SPMV(spmv_csr_plan_describe(res[0].A, variant, plan, sizeof plan));
        out << id << "\t" << plan << "\n";
        for (Resident &R : res) {
            OK_SPMV(spmv_csr_destroy(R.A));
            OK_HIP(hipFree(R.d_rp)); OK_HIP(hipFree(R.d_ci)); OK_HIP(hipFree(R.d_va)); OK_HIP(hipFree(R.d_y));
        }
        OK_HIP(hipFree(d_x));
    }
    out.close();
    return out ? 0 : 1;
}
