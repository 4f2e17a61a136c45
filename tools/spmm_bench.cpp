// Stand-alone micro-benchmark of K1 (glass_spmm_csr_f32) — no torch, links libglass_hip.so only.
// Generates a synthetic symmetric graph on the host, times the kernel with HIP events on its own
// stream, prints algorithmic GB/s against the HBM roofline, and spot-checks rows in fp64.
// Also the program to put after `rocprofv3 ... --` for per-kernel traces and PMC counters.
//
//   spmm_bench <shape> [H] [iters] [--rp K] [--full] [--ld L]
//   --ld L : row stride of X and Y in floats (>= H; default H): the product over a column block of a wider matrix
//   --rp K : override the plan's flat-mode factor (header word 13; K = 0 also clears the flat-share word 15, i.e. the
//            row-mode-only kernel) for A/B runs; --full : check EVERY row in fp64
//   shape: ppi_bp | hpo_neuro | em_user | powerlaw | density-like | N:PAIRS[:zipf] | calib:N
//   calib:N = random permutation matrix (one edge per row, every X row read exactly once): a known
//   byte count in K1's own access pattern, used to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <random>
#include <string>
#include <unordered_set>
#include <vector>

#include "../include/glass_hip.h"

#define HIP_OK(x)                                                                        \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                     \
        }                                                                                \
    } while (0)

struct Graph {
    int64_t n = 0;
    std::vector<int32_t> rowptr, col;
    std::vector<float> val;
};

// distinct undirected pairs, symmetrised, sorted by (row, col); val = 1/deg[row] ("mean")
static Graph make_graph(int64_t n, int64_t pairs, double zipf, uint64_t seed) {
    std::mt19937_64 rng(seed);
    std::vector<double> cdf;
    if (zipf > 0) {
        cdf.resize(n);
        double s = 0;
        for (int64_t i = 0; i < n; ++i) cdf[i] = (s += pow((double)(i + 1), -zipf));
        for (auto& c : cdf) c /= s;
    }
    std::uniform_real_distribution<double> U(0.0, 1.0);
    auto draw = [&]() -> int64_t {
        if (zipf > 0) {
            int64_t k = std::lower_bound(cdf.begin(), cdf.end(), U(rng)) - cdf.begin();
            return std::min<int64_t>(k, n - 1);
        }
        return (int64_t)(rng() % (uint64_t)n);
    };
    std::unordered_set<uint64_t> seen;
    seen.reserve((size_t)pairs * 2);
    std::vector<int32_t> deg(n, 0);
    const int32_t max_deg = 50000;
    std::vector<std::pair<int32_t, int32_t>> und;
    und.reserve(pairs);
    while ((int64_t)und.size() < pairs) {
        int64_t u = draw(), v = draw();
        if (u == v) continue;
        if (u > v) std::swap(u, v);
        if (deg[u] >= max_deg || deg[v] >= max_deg) continue;
        if (!seen.insert((uint64_t)u * (uint64_t)n + (uint64_t)v).second) continue;
        und.emplace_back((int32_t)u, (int32_t)v);
        ++deg[u];
        ++deg[v];
    }
    Graph g;
    g.n = n;
    g.rowptr.assign(n + 1, 0);
    for (int64_t i = 0; i < n; ++i) g.rowptr[i + 1] = g.rowptr[i] + deg[i];
    g.col.resize(2 * pairs);
    std::vector<int32_t> fill(g.rowptr.begin(), g.rowptr.end() - 1);
    for (auto& p : und) {
        g.col[fill[p.first]++] = p.second;
        g.col[fill[p.second]++] = p.first;
    }
    for (int64_t i = 0; i < n; ++i) std::sort(g.col.begin() + g.rowptr[i], g.col.begin() + g.rowptr[i + 1]);
    g.val.resize(2 * pairs);
    for (int64_t i = 0; i < n; ++i)
        for (int32_t e = g.rowptr[i]; e < g.rowptr[i + 1]; ++e) g.val[e] = 1.0f / (float)std::max(deg[i], 1);
    return g;
}

#ifdef GLASS_K1_TRACE
extern "C" int glass_k1_trace_set(unsigned long long* p);
#endif

int main(int argc, char** argv) {
    int rp_override = -1;
    bool full = false;
    int64_t ld = 0;
    std::vector<char*> pos_args;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--rp") && i + 1 < argc) rp_override = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--full")) full = true;
        else if (!strcmp(argv[i], "--ld") && i + 1 < argc) ld = atoll(argv[++i]);
        else pos_args.push_back(argv[i]);
    }
    std::string shape = pos_args.size() > 0 ? pos_args[0] : "ppi_bp";
    int64_t H = pos_args.size() > 1 ? atoll(pos_args[1]) : 64;
    int iters = pos_args.size() > 2 ? atoi(pos_args[2]) : 50;
    if (ld < H) ld = H;
    int64_t n, pairs;
    double zipf = 0;
    if (shape == "ppi_bp") n = 17080, pairs = 316951;
    else if (shape == "hpo_neuro") n = 14587, pairs = 3238174;
    else if (shape == "em_user") n = 50000, pairs = 500000;
    else if (shape == "powerlaw") n = 1000000, pairs = 10000000, zipf = 0.8;
    else if (shape == "density-like") n = 4998, pairs = 29962;
    else if (shape.rfind("calib:", 0) == 0) n = atoll(shape.c_str() + 6), pairs = -1;
    else if (shape.rfind("file:", 0) == 0) n = 0, pairs = -2;  // a CSR dumped by bench.py (the shipped density graph): see below
    else {
        double z = 0;
        long long a = 0, b = 0;
        int k = sscanf(shape.c_str(), "%lld:%lld:%lf", &a, &b, &z);
        if (k < 2) {
            fprintf(stderr, "bad shape %s\n", shape.c_str());
            return 1;
        }
        n = a, pairs = b, zipf = z;
    }
    Graph g;
    if (pairs == -2) {
        // file:<path> — int64 n, int64 nnz, int32 rowptr[n + 1], int32 col[nnz], float val[nnz] (little endian, as numpy wrote them)
        FILE* f = fopen(shape.c_str() + 5, "rb");
        int64_t hdr[2] = {0, 0};
        if (!f || fread(hdr, 8, 2, f) != 2 || hdr[0] <= 0 || hdr[1] <= 0) {
            fprintf(stderr, "cannot read %s\n", shape.c_str() + 5);
            return 1;
        }
        n = hdr[0];
        g.n = n;
        g.rowptr.resize(n + 1);
        g.col.resize(hdr[1]);
        g.val.resize(hdr[1]);
        if (fread(g.rowptr.data(), 4, n + 1, f) != (size_t)(n + 1) || fread(g.col.data(), 4, hdr[1], f) != (size_t)hdr[1] ||
            fread(g.val.data(), 4, hdr[1], f) != (size_t)hdr[1]) {
            fprintf(stderr, "short read on %s\n", shape.c_str() + 5);
            return 1;
        }
        fclose(f);
    } else if (pairs < 0) {  // permutation matrix
        g.n = n;
        g.rowptr.resize(n + 1);
        g.col.resize(n);
        g.val.assign(n, 1.0f);
        for (int64_t i = 0; i <= n; ++i) g.rowptr[i] = (int32_t)i;
        for (int64_t i = 0; i < n; ++i) g.col[i] = (int32_t)i;
        std::mt19937_64 prng(7);
        std::shuffle(g.col.begin(), g.col.end(), prng);
    } else {
        g = make_graph(n, pairs, zipf, 0);
    }
    const int64_t nnz = (int64_t)g.col.size();
    int32_t maxdeg = 0;
    for (int64_t i = 0; i < n; ++i) maxdeg = std::max(maxdeg, g.rowptr[i + 1] - g.rowptr[i]);

    int64_t words = 0;
    if (glass_spmm_plan_build(g.rowptr.data(), n, nullptr, &words)) return 3;
    std::vector<int32_t> plan(words);
    if (glass_spmm_plan_build(g.rowptr.data(), n, plan.data(), &words)) return 3;
    if (rp_override >= 0) {
        plan[13] = rp_override;
        if (rp_override == 0) plan[15] = 0;  // no flat-eligible share: the launch takes the row-mode-only kernel
    }
    const int64_t ws_bytes = glass_spmm_ws_bytes(plan.data(), H);

    std::vector<float> X((size_t)n * ld);
    std::mt19937 r32(1);
    std::normal_distribution<float> N01(0.f, 1.f);
    for (auto& v : X) v = N01(r32);

    int32_t *d_rowptr, *d_col, *d_plan;
    float *d_val, *d_X, *d_Y;
    void* d_ws = nullptr;
    HIP_OK(hipMalloc(&d_rowptr, (n + 1) * 4));
    HIP_OK(hipMalloc(&d_col, nnz * 4));
    HIP_OK(hipMalloc(&d_val, nnz * 4));
    HIP_OK(hipMalloc(&d_plan, words * 4));
    HIP_OK(hipMalloc(&d_X, (size_t)n * ld * 4));
    HIP_OK(hipMalloc(&d_Y, (size_t)n * ld * 4));
    if (ws_bytes > 0) HIP_OK(hipMalloc(&d_ws, ws_bytes));
    HIP_OK(hipMemcpy(d_rowptr, g.rowptr.data(), (n + 1) * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_col, g.col.data(), nnz * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_val, g.val.data(), nnz * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_plan, plan.data(), words * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_X, X.data(), (size_t)n * ld * 4, hipMemcpyHostToDevice));

    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    auto run = [&]() {
        int rc = glass_spmm_csr_f32(d_rowptr, d_col, d_val, d_X, ld, d_Y, ld, n, H, plan.data(), d_plan, d_ws, st);
        if (rc) {
            fprintf(stderr, "spmm rc=%d: %s\n", rc, glass_last_error_string());
            exit(4);
        }
    };
    for (int i = 0; i < 5; ++i) run();
    HIP_OK(hipStreamSynchronize(st));
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    HIP_OK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) run();
    HIP_OK(hipEventRecord(e1, st));
    HIP_OK(hipEventSynchronize(e1));
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    const double t = ms * 1e-3 / iters;
#ifdef GLASS_K1_TRACE
    {   // one traced launch: per-wave wall-clock stamps (100 MHz) of the sweep kernel
        const int nw = plan[4];
        unsigned long long* d_tr;
        HIP_OK(hipMalloc(&d_tr, (size_t)nw * 32));
        HIP_OK(hipMemset(d_tr, 0, (size_t)nw * 32));
        glass_k1_trace_set(d_tr);
        run();
        HIP_OK(hipStreamSynchronize(st));
        glass_k1_trace_set(nullptr);
        std::vector<unsigned long long> tr((size_t)nw * 4);
        HIP_OK(hipMemcpy(tr.data(), d_tr, (size_t)nw * 32, hipMemcpyDeviceToHost));
        unsigned long long t_min = ~0ull, t_max = 0;
        for (int i = 0; i < nw; ++i) {
            t_min = std::min(t_min, tr[i * 4]);
            t_max = std::max(t_max, tr[i * 4 + 3]);
        }
        double ph[3] = {0, 0, 0};
        std::vector<double> starts(nw), lifes(nw);
        for (int i = 0; i < nw; ++i) {
            for (int k = 0; k < 3; ++k) ph[k] += (double)(tr[i * 4 + k + 1] - tr[i * 4 + k]) * 10.0;
            starts[i] = (double)(tr[i * 4] - t_min) * 10.0;
            lifes[i] = (double)(tr[i * 4 + 3] - tr[i * 4]) * 10.0;
        }
        std::sort(starts.begin(), starts.end());
        std::sort(lifes.begin(), lifes.end());
        fprintf(stderr, "trace: %d waves, span %.0f ns; mean phase ns: item %.0f, index %.0f, gather %.0f; life p10/p50/p90/max %.0f/%.0f/%.0f/%.0f; "
                "start p10/p25/p50/p75/p90/max %.0f/%.0f/%.0f/%.0f/%.0f/%.0f\n", nw, (double)(t_max - t_min) * 10.0, ph[0] / nw, ph[1] / nw, ph[2] / nw,
                lifes[nw / 10], lifes[nw / 2], lifes[nw * 9 / 10], lifes[nw - 1], starts[nw / 10], starts[nw / 4], starts[nw / 2],
                starts[nw * 3 / 4], starts[nw * 9 / 10], starts[nw - 1]);
    }
#endif
    // algorithmic bytes per pass: nnz*(4H+8) + N*(4H+4)
    const double bytes = (double)nnz * (4.0 * H + 8) + (double)n * (4.0 * H + 4);

    // spot check 64 rows in fp64
    std::vector<float> Y((size_t)n * ld);
    HIP_OK(hipMemcpy(Y.data(), d_Y, (size_t)n * ld * 4, hipMemcpyDeviceToHost));
    double max_err = 0, max_ref = 0;
    const int64_t n_check = full ? n : 64;
    for (int64_t k = 0; k < n_check; ++k) {
        int64_t r = (full || k < 8) ? k : (int64_t)((uint64_t)(k * 2654435761u) % (uint64_t)n);
        if (!full && k == 8) {  // the longest row
            for (int64_t i = 0; i < n; ++i)
                if (g.rowptr[i + 1] - g.rowptr[i] == maxdeg) r = i;
        }
        for (int64_t c = 0; c < H; ++c) {
            double s = 0;
            for (int32_t e = g.rowptr[r]; e < g.rowptr[r + 1]; ++e) s += (double)g.val[e] * X[(size_t)g.col[e] * ld + c];
            max_err = std::max(max_err, fabs(s - (double)Y[(size_t)r * ld + c]));
            max_ref = std::max(max_ref, fabs(s));
        }
    }
    printf("{\"shape\": \"%s\", \"rp\": %d, \"rows_checked\": %lld, \"N\": %lld, \"nnz\": %lld, \"max_deg\": %d, \"H\": %lld, \"sweep_waves\": %d, "
           "\"long_items\": %d, \"reduce_rows\": %d, \"us_per_pass\": %.2f, \"edges_per_s\": %.4g, \"alg_GBps\": %.1f, "
           "\"frac_of_8TBps\": %.3f, \"spot_rel_err\": %.2e}\n",
           shape.c_str(), plan[13], (long long)n_check, (long long)n, (long long)nnz, maxdeg, (long long)H, plan[4], plan[5], plan[6], t * 1e6,
           nnz / t, bytes / t / 1e9, bytes / t / 8e12, max_err / (max_ref > 0 ? max_ref : 1));
    return max_err / (max_ref > 0 ? max_ref : 1) < 1e-5 ? 0 : 5;
}
