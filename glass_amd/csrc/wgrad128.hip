// Weight gradient of the trans pair at hidden 128 on mid-size graphs (em_user-shape: N = 50 000), rows shared through LDS
// (round 6; reference impl/models.py:158-162 backward: dW[2H, H] = dZ^T X, dZ = mix'(dout) . act'(T)).
//
// The register-pipelined kernel this replaces at that shape (wgrad_partial_split_kernel, linear.hip) cuts the [256 x 128]
// output into four 128 x 64 tiles per slab, one workgroup each, every workgroup reading its rows' gradient, pre-activation
// half and input half straight from global memory: 5 120 B requested per row for 2 048 B of operands.  tools/wgrad_probe.py
// put that launch at 14 us fixed + 0.65-0.75 us per 1 000 rows of REQUEST bandwidth between the CUs and L2 — sharing through
// L1 (a paired form) did not help, fewer vector instructions did not help.  Here a slab has ONE workgroup of
// eight waves that requests every row once: stages of 32 rows, the loaders (two rows x one column quad per thread and
// operand) synthesise dZ, cut row pairs into bf16 pieces (split2) and write the transposed, swizzled images of
// wgrad_common.h (stg2s_*: [piece][column][32 rows], no padding, fragment reads and stores conflict free); wave (ob, ib)
// owns outputs 64 ob .. + 63 x inputs 64 ib .. + 63 = 16 accumulator tiles, 96 MFMAs of v_mfma_f32_16x16x32_bf16 per stage.
// Two 72 KiB stage buffers: one workgroup per CU.  Partials: the four 128 x 64 sub-tiles in plain [o][i] order at the
// places the batched reduce expects them (header[2] = 1), bias partials from the loaders' own sums.
#include "common.h"
#include "wgrad_common.h"
#include "dense_common.h"

namespace glass {

constexpr int kW128Threads = 512;
constexpr int kW128WordsA = 3 * 256 * 16, kW128WordsB = 3 * 128 * 16;   // 32-bit words of the dZ^T / X^T images of one buffer
constexpr int kW128Words = kW128WordsA + kW128WordsB;                   // 18 432 words = 72 KiB
constexpr size_t kW128LdsBytes = (size_t)2 * kW128Words * 4;

__global__ __launch_bounds__(kW128Threads, 1) void wgrad128_trans_kernel(const float* __restrict__ X, int64_t ldx, int64_t N,
                                                                         int rows_per_slab, int n_slabs, float* __restrict__ part_w,
                                                                         float* __restrict__ part_b, float* __restrict__ header,
                                                                         WgradSynth sy) {
    extern __shared__ float lds[];
    constexpr int H = 128;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    // loader role: row pair rp of the stage, columns 4 ga .. 4 ga + 3.  A wave takes 4 row pairs x 16 column quads (waves 0-3
    // the low 64 columns, 4-7 the high ones): the 64 words of one of its image stores then lie in 64 different banks
    // (stg2s_word; with 2 row pairs x 32 quads per wave every store was 2-way conflicted: 0.64 conflict cycles per active one)
    const int rp = (tid >> 4) & 15, ga = (tid & 15) | ((tid >> 8) << 4);
    const int ob = w >> 1, ib = w & 1;       // this wave's 64 outputs / 64 inputs
    const int bx = blockIdx.x;
    if (header && bx == 0 && tid == 0) {
        header[0] = 0.f;
        header[1] = sy.zr;
        header[2] = 1.f;  // tiles in plain [o][i] order
    }
    const int64_t r0 = (int64_t)bx * rows_per_slab;
    const int64_t r1 = r0 + rows_per_slab < N ? r0 + rows_per_slab : N;
    const int n_st = (int)((r1 - r0 + 31) / 32);
    const bool has_t = sy.act != GLASS_ACT_NONE;
    const buf_rsrc r_d = make_rsrc(sy.dsrc, N * sy.ldd * 4), r_x = make_rsrc(X, N * ldx * 4), r_m = make_rsrc(sy.mask, N);
    const buf_rsrc r_t = make_rsrc(has_t ? sy.T : sy.dsrc, has_t ? N * sy.ldt * 4 : 0);  // no activation: reads 0, act'(0) unused
    struct Raw {
        float4 d[2], t1[2], t0[2], x[2];
        unsigned mk[2];
    };
    auto issue = [&](int st, Raw& R) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t r = r0 + 32 * st + 2 * rp + u;
            const bool ok = r < r1;
            const int ri = (int)r;
            R.d[u] = buf_load4(r_d, ok ? (int)((ri * sy.ldd + 4 * ga) * 4) : kBufOOB);
            R.t1[u] = buf_load4(r_t, ok ? (int)((ri * sy.ldt + 4 * ga) * 4) : kBufOOB);
            R.t0[u] = buf_load4(r_t, ok ? (int)((ri * sy.ldt + H + 4 * ga) * 4) : kBufOOB);
            R.x[u] = buf_load4(r_x, ok ? (int)((ri * ldx + 4 * ga) * 4) : kBufOOB);
            R.mk[u] = __builtin_amdgcn_raw_buffer_load_b8(r_m, ok ? ri : kBufOOB, 0, 0);
        }
    };
    float bs1[4] = {0.f, 0.f, 0.f, 0.f}, bs0[4] = {0.f, 0.f, 0.f, 0.f};  // bias partials of outputs 4 ga + k / H + 4 ga + k
    auto commit = [&](int st, const Raw& R) __attribute__((always_inline)) {
        unsigned* dz = reinterpret_cast<unsigned*>(lds) + (st & 1) * kW128Words;
        unsigned* xi = dz + kW128WordsA;
        float z1[2][4], z0[2][4], xv[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float c1 = R.mk[u] ? sy.zr : sy.omz, c0 = R.mk[u] ? sy.omz : sy.zr;
            const float d[4] = {R.d[u].x, R.d[u].y, R.d[u].z, R.d[u].w};
            const float t1[4] = {R.t1[u].x, R.t1[u].y, R.t1[u].z, R.t1[u].w}, t0[4] = {R.t0[u].x, R.t0[u].y, R.t0[u].z, R.t0[u].w};
            xv[u][0] = R.x[u].x, xv[u][1] = R.x[u].y, xv[u][2] = R.x[u].z, xv[u][3] = R.x[u].w;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                z1[u][k] = d[k] * c1, z0[u][k] = d[k] * c0;
                if (has_t) z1[u][k] *= act_grad(sy.act, t1[k]), z0[u][k] *= act_grad(sy.act, t0[k]);
                bs1[k] += z1[u][k];
                bs0[k] += z0[u][k];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned hi, mid, lo;
            const int c = 4 * ga + k;
            split2(z1[0][k], z1[1][k], hi, mid, lo);
            int wd = stg2s_word(c, rp);
            dz[wd] = hi, dz[256 * 16 + wd] = mid, dz[2 * 256 * 16 + wd] = lo;
            split2(z0[0][k], z0[1][k], hi, mid, lo);
            wd = stg2s_word(H + c, rp);
            dz[wd] = hi, dz[256 * 16 + wd] = mid, dz[2 * 256 * 16 + wd] = lo;
            split2(xv[0][k], xv[1][k], hi, mid, lo);
            wd = stg2s_word(c, rp);
            xi[wd] = hi, xi[128 * 16 + wd] = mid, xi[2 * 128 * 16 + wd] = lo;
        }
    };
    Raw rawA, rawB;
    issue(0, rawA);
    issue(1, rawB);
    wg_f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int it = 0; it < 4; ++it) acc[a][it] = (wg_f32x4){0.f, 0.f, 0.f, 0.f};
    commit(0, rawA);
    issue(2, rawA);
    lds_barrier();
    for (int st = 0; st < n_st; ++st) {
        const uint4* dz = reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned*>(lds) + (st & 1) * kW128Words);
        const uint4* xi = dz + kW128WordsA / 4;
        uint4 af[4][3];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) af[a][pc] = dz[pc * 256 * 4 + stg2s_chunk(64 * ob + 16 * a + j, q)];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            uint4 bf[3];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) bf[pc] = xi[pc * 128 * 4 + stg2s_chunk(64 * ib + 16 * it + j, q)];
#pragma unroll
            for (int a = 0; a < 4; ++a) { GLASS_SPLIT6(acc[a][it], af[a], bf); }
        }
        if (st + 1 < n_st) {
            if (st & 1) {
                commit(st + 1, rawA);
                issue(st + 3, rawA);
            } else {
                commit(st + 1, rawB);
                issue(st + 3, rawB);
            }
            lds_barrier();
        }
    }
    // the four 128 x 64 sub-tiles (z = o / 128, y = i / 64) in plain [o][i] order: o = 64 ob + 16 a + 4 q + r, i = 64 ib + 16 it + j
    {
        const int z = ob >> 1, y = ib;
        float* pw = part_w + ((int64_t)(z * 2 + y) * n_slabs + bx) * kTile;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) pw[(64 * (ob & 1) + 16 * a + 4 * q + r) * kIT + 16 * it + j] = acc[a][it][r];
    }
    if (part_b) {  // bias partial: the wave's four row-pair slots of a column by shuffles, the four waves of a column half through LDS
        lds_barrier();  // (every wave is done reading the last stage's image)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bs1[k] += __shfl_xor(bs1[k], 16);
            bs0[k] += __shfl_xor(bs0[k], 16);
            bs1[k] += __shfl_xor(bs1[k], 32);
            bs0[k] += __shfl_xor(bs0[k], 32);
        }
        if (lane < 16) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                lds[(w & 3) * 256 + 4 * ga + k] = bs1[k];
                lds[(w & 3) * 256 + H + 4 * ga + k] = bs0[k];
            }
        }
        lds_barrier();
        if (tid < 256) {
            float s = 0.f;
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) s += lds[ww * 256 + tid];
            part_b[((int64_t)(tid >> 7) * n_slabs + bx) * kOT + (tid & 127)] = s;
        }
    }
}

// ---- comb pair (effective-weight S / L form, reference impl/models.py:169-173 backward): S = sum over ALL rows of dc^T [g || x_]
// ([128 x 256]: one tile per slab, the same eight-wave workgroup with the roles of the images swapped: A = dc^T [128 columns],
// B = [g || x_]^T [256 columns]), L = the same sum over the LABELED rows.  The labeled rows are few (a batch's subgraph
// nodes) and this entry point has the label bytes, not the batch's row list: n_l trailing workgroups each scan the label
// bytes of ONE chunk of N / n_l rows (ordered compaction by wave ballots into LDS, as the register-pipelined kernel does per
// slab), then walk their list through the same 32-row stages with gathered rows.  Partials in the effective-weight layout of
// the batched reduce with the L tiles of a (z = 1, y) block being the first n_l of its n_slabs places (header[0] = 3,
// header[3] = n_l): 4 n_l tiles instead of 4 n_slabs — with one L tile per slab the 224 slabs of config 4 would write and
// re-read 29 MB of nearly empty tiles.
constexpr int kW128ListCap = 3968;  // labeled rows one L workgroup can list (the 15.5 KiB of LDS behind the two stage buffers)

__global__ __launch_bounds__(kW128Threads, 1) void wgrad128_comb_kernel(const float* __restrict__ dc, int64_t ldd,
                                                                        const float* __restrict__ G, int64_t ldg,
                                                                        const float* __restrict__ X, int64_t ldx,
                                                                        const uint8_t* __restrict__ mask, int64_t N, int rows_per_slab,
                                                                        int n_slabs, int n_l, float zr, float* __restrict__ part_w,
                                                                        float* __restrict__ part_b, float* __restrict__ header) {
    extern __shared__ float lds[];
    constexpr int H = 128;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int rp = (tid >> 4) & 15, ga = (tid & 15) | ((tid >> 8) << 4);  // (as in wgrad128_trans_kernel)
    const int ob = w >> 2, ib = w & 3;       // this wave's 64 outputs (of 128) / 64 inputs (of 256)
    const int bx = blockIdx.x;
    if (header && bx == 0 && tid == 0) {
        header[0] = 3.f;   // effective-weight form with n_l list tiles (wgrad_reduce_body)
        header[1] = zr;
        header[2] = 1.f;   // plain [o][i] tiles
        header[3] = (float)n_l;
    }
    const bool lab = bx >= n_slabs;
    int* list = reinterpret_cast<int*>(lds + 2 * kW128Words);
    __shared__ int wave_cnt[kW128Threads / 64];
    int64_t r0 = 0, r1 = 0;
    int n_rows;
    if (!lab) {
        r0 = (int64_t)bx * rows_per_slab;
        r1 = r0 + rows_per_slab < N ? r0 + rows_per_slab : N;
        n_rows = (int)(r1 - r0);
    } else {
        // this chunk's labeled rows in row order
        const int64_t chunk = (N + n_l - 1) / n_l;
        const int64_t c0 = (int64_t)(bx - n_slabs) * chunk, c1 = c0 + chunk < N ? c0 + chunk : N;
        int base = 0;
        for (int64_t p = c0; p < c1; p += kW128Threads) {
            const int64_t r = p + tid;
            const bool flag = r < c1 && mask[r] != 0;
            const unsigned long long bal = __ballot(flag);
            if (lane == 0) wave_cnt[w] = __popcll(bal);
            __syncthreads();
            int off = base, total = 0;
            for (int ww = 0; ww < kW128Threads / 64; ++ww) {
                if (ww < w) off += wave_cnt[ww];
                total += wave_cnt[ww];
            }
            if (flag) list[off + __popcll(bal & ((1ull << lane) - 1ull))] = (int)r;
            base += total;
            __syncthreads();
        }
        n_rows = base;
    }
    const int n_st = (n_rows + 31) / 32;
    const buf_rsrc r_d = make_rsrc(dc, N * ldd * 4), r_g = make_rsrc(G, N * ldg * 4), r_x = make_rsrc(X, N * ldx * 4);
    struct Raw {
        float4 d[2], g[2], x[2];
    };
    auto issue = [&](int st, Raw& R) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int pos = 32 * st + 2 * rp + u;
            int ri = -1;
            if (pos < n_rows) ri = lab ? list[pos] : (int)(r0 + pos);
            R.d[u] = buf_load4(r_d, ri >= 0 ? (int)((ri * ldd + 4 * ga) * 4) : kBufOOB);
            R.g[u] = buf_load4(r_g, ri >= 0 ? (int)((ri * ldg + 4 * ga) * 4) : kBufOOB);
            R.x[u] = buf_load4(r_x, ri >= 0 ? (int)((ri * ldx + 4 * ga) * 4) : kBufOOB);
        }
    };
    float bs[4] = {0.f, 0.f, 0.f, 0.f};  // bias partial of outputs 4 ga + k over this thread's rows
    auto commit = [&](int st, const Raw& R) __attribute__((always_inline)) {
        unsigned* dI = reinterpret_cast<unsigned*>(lds) + (st & 1) * kW128Words;   // dc^T: 128 columns
        unsigned* inI = dI + kW128WordsB;                                             // [g || x_]^T: 256 columns
        const float d0[4] = {R.d[0].x, R.d[0].y, R.d[0].z, R.d[0].w}, d1[4] = {R.d[1].x, R.d[1].y, R.d[1].z, R.d[1].w};
        const float g0[4] = {R.g[0].x, R.g[0].y, R.g[0].z, R.g[0].w}, g1[4] = {R.g[1].x, R.g[1].y, R.g[1].z, R.g[1].w};
        const float x0[4] = {R.x[0].x, R.x[0].y, R.x[0].z, R.x[0].w}, x1[4] = {R.x[1].x, R.x[1].y, R.x[1].z, R.x[1].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned hi, mid, lo;
            const int c = 4 * ga + k;
            bs[k] += d0[k] + d1[k];
            split2(d0[k], d1[k], hi, mid, lo);
            int wd = stg2s_word(c, rp);
            dI[wd] = hi, dI[128 * 16 + wd] = mid, dI[2 * 128 * 16 + wd] = lo;
            split2(g0[k], g1[k], hi, mid, lo);
            inI[wd] = hi, inI[256 * 16 + wd] = mid, inI[2 * 256 * 16 + wd] = lo;
            split2(x0[k], x1[k], hi, mid, lo);
            wd = stg2s_word(H + c, rp);
            inI[wd] = hi, inI[256 * 16 + wd] = mid, inI[2 * 256 * 16 + wd] = lo;
        }
    };
    Raw rawA, rawB;
    issue(0, rawA);
    issue(1, rawB);
    wg_f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int it = 0; it < 4; ++it) acc[a][it] = (wg_f32x4){0.f, 0.f, 0.f, 0.f};
    commit(0, rawA);
    issue(2, rawA);
    lds_barrier();
    for (int st = 0; st < n_st; ++st) {
        const uint4* dI = reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned*>(lds) + (st & 1) * kW128Words);
        const uint4* inI = dI + kW128WordsB / 4;
        uint4 af[4][3];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) af[a][pc] = dI[pc * 128 * 4 + stg2s_chunk(64 * ob + 16 * a + j, q)];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            uint4 bf[3];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) bf[pc] = inI[pc * 256 * 4 + stg2s_chunk(64 * ib + 16 * it + j, q)];
#pragma unroll
            for (int a = 0; a < 4; ++a) { GLASS_SPLIT6(acc[a][it], af[a], bf); }
        }
        if (st + 1 < n_st) {
            if (st & 1) {
                commit(st + 1, rawA);
                issue(st + 3, rawA);
            } else {
                commit(st + 1, rawB);
                issue(st + 3, rawB);
            }
            lds_barrier();
        }
    }
    // sub-tile (z, y = ib): z = 0 the S tile of slab bx, z = 1 the L tile of list workgroup bx - n_slabs (the first n_l places of its block)
    {
        const int z = lab ? 1 : 0, place = lab ? bx - n_slabs : bx;
        float* pw = part_w + ((int64_t)(z * 4 + ib) * n_slabs + place) * kTile;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) pw[(64 * ob + 16 * a + 4 * q + r) * kIT + 16 * it + j] = acc[a][it][r];
    }
    if (part_b) {
        lds_barrier();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bs[k] += __shfl_xor(bs[k], 16);
            bs[k] += __shfl_xor(bs[k], 32);
        }
        if (lane < 16) {
#pragma unroll
            for (int k = 0; k < 4; ++k) lds[(w & 3) * 128 + 4 * ga + k] = bs[k];
        }
        lds_barrier();
        if (tid < 128) {
            float s = 0.f;
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) s += lds[ww * 128 + tid];
            const int z = lab ? 1 : 0, place = lab ? bx - n_slabs : bx;
            part_b[((int64_t)z * n_slabs + place) * kOT + tid] = s;
        }
    }
}

// shapes served: the trans pair of hidden 128 (O = 256, I = 128) on a graph that gives every slab a few 32-row stages
bool wgrad128_shape(int64_t N, int64_t O, int64_t I) { return O == 256 && I == 128 && N >= 8192 && N <= kFusedBwdMaxRows * 4; }

// ... and its comb pair (O = I = 256) in the S / L form above; n_l list workgroups so that a chunk's rows fit one list
bool wgrad128_comb_shape(int64_t N, int64_t O, int64_t I) { return O == 256 && I == 256 && N >= 8192 && N <= kFusedBwdMaxRows * 4; }
int wgrad128_comb_lists(int64_t N) {
    int64_t n_l = ceil_div(N, (int64_t)kW128ListCap);
    if (n_l < 16) n_l = 16;
    return (int)n_l;
}

void launch_wgrad128_comb(const float* dc, int64_t ldd, const float* G, int64_t ldg, const float* X, int64_t ldx, const uint8_t* mask,
                          int64_t N, int rows_per_slab, int n_slabs, float zr, float* part_w, float* part_b, float* header,
                          hipStream_t st) {
    const size_t lds = kW128LdsBytes + (size_t)kW128ListCap * 4;
    (void)hipFuncSetAttribute((const void*)wgrad128_comb_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int n_l = wgrad128_comb_lists(N);
    hipLaunchKernelGGL(wgrad128_comb_kernel, dim3((unsigned)(n_slabs + n_l)), dim3(kW128Threads), lds, st, dc, ldd, G, ldg, X, ldx, mask, N,
                       rows_per_slab, n_slabs, n_l, zr, part_w, part_b, header);
}

void launch_wgrad128_trans(const float* X, int64_t ldx, int64_t N, int rows_per_slab, int n_slabs, float* part_w, float* part_b,
                           float* header, const WgradSynth& sy, hipStream_t st) {
    (void)hipFuncSetAttribute((const void*)wgrad128_trans_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kW128LdsBytes);
    hipLaunchKernelGGL(wgrad128_trans_kernel, dim3((unsigned)n_slabs), dim3(kW128Threads), kW128LdsBytes, st, X, ldx, N, rows_per_slab,
                       n_slabs, part_w, part_b, header, sy);
}

}  // namespace glass
